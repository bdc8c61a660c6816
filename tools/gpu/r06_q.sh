#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_q
mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_wgrad_ordered_gpu.py tests/test_wino_gpu.py -x -q 2>&1 | grep -E "^E|assert|passed|failed" | head -12
i=0
for rep in 1 2 3 4; do
for cfg in "TREE=r05" "X=0" "PICONS_WGRAD_ATOMIC=1" "PICONS_WGRAD_STEM_X6=0" "PICONS_WGRAD_ATOMIC=1 PICONS_WGRAD_STEM_X6=0"; do
  i=$((i+1))
  if [ "$cfg" = "TREE=r05" ]; then cd $R/_ab_r05; else cd $R; fi
  env $cfg timeout 600 python3 bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/b_$i.json 2> $O/b_$i.err
  python3 -c "
import json; j=json.load(open('$O/b_$i.json')); print('%-50s staged %.3f  resident %.3f' % ('$cfg', j['ms_per_step'], (j.get('resident') or {}).get('ms_per_step') or 0))"
done
done

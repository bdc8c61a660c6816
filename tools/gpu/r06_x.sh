#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_x
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_step_gpu.py -x -q -k "small or deterministic or lanes or golden_full_size" 2>&1 | tail -3
i=0
for rep in 1 2 3; do
for cfg in "TREE=r05" "PICONS_CAPS_PREP=0" "PICONS_CAPS_PREP=1"; do
  i=$((i+1))
  if [ "$cfg" = "TREE=r05" ]; then cd $R/_ab_r05; else cd $R; fi
  env $cfg timeout 600 python3 bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/b_$i.json 2> $O/b_$i.err
  python3 -c "
import json; j=json.load(open('$O/b_$i.json')); print('%-30s staged %.3f  resident %.3f  loss %.6f' % ('$cfg', j['ms_per_step'], (j.get('resident') or {}).get('ms_per_step') or 0, j['loss']['total']))"
done
done

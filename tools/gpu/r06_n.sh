#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_n
mkdir -p $O
cd $R
i=0
for rep in 1 2; do
for cfg in "X=0" "PICONS_WGRAD_ATOMIC=1" "PICONS_WGRAD_STEM_X6=0" "PICONS_WINO_STRIPS=0" "PICONS_WGRAD_ATOMIC=1 PICONS_WGRAD_STEM_X6=0 PICONS_WINO_STRIPS=0"; do
  i=$((i+1))
  env $cfg timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/b_$i.json 2> $O/b_$i.err
  python3 -c "
import json; j=json.load(open('$O/b_$i.json')); print('%-75s staged %.3f  resident %.3f' % ('$cfg', j['ms_per_step'], (j.get('resident') or {}).get('ms_per_step') or 0))"
done
done
cd $R/_ab_r05; timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/b_r05.json 2> $O/b_r05.err
python3 -c "
import json; j=json.load(open('$O/b_r05.json')); print('%-75s staged %.3f  resident %.3f' % ('round-5 tree', j['ms_per_step'], (j.get('resident') or {}).get('ms_per_step') or 0))"

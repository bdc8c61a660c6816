#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_w
mkdir -p $O
cd $R
i=0
for rep in 1 2 3; do
for cfg in "X=0" "PICONS_BENCH_EARLY_PREP=1" "PICONS_BENCH_EARLY_PREP=1 PICONS_STAGE_LANE=0"; do
  i=$((i+1))
  env $cfg timeout 600 python3 bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/b_$i.json 2> $O/b_$i.err
  python3 -c "
import json; j=json.load(open('$O/b_$i.json')); print('%-50s staged %.3f  resident %.3f  loss %.6f' % ('$cfg', j['ms_per_step'], (j.get('resident') or {}).get('ms_per_step') or 0, j['loss']['total']))"
done
done

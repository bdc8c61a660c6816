#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_u
mkdir -p $O
cd $R
i=0
for rep in 1 2 3; do
for cfg in "PICONS_HACK_LATE=0" "PICONS_HACK_LATE=1" "PICONS_HACK_LATE=3"; do
  i=$((i+1))
  env $cfg timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-timing --resident-inputs > $O/b_$i.json 2> $O/b_$i.err
  python3 -c "
import json; j=json.load(open('$O/b_$i.json')); print('%-30s resident %.3f' % ('$cfg', j['ms_per_step']))"
done
done

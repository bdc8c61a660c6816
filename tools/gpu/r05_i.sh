#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_i
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in "" MaxPool3d_3a_3x3 Mixed_3c MaxPool3d_4a_3x3 Mixed_4c Mixed_4f; do
    PICONS_SKIP_FWD_AFTER=$v timeout 600 python3 bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing --resident-inputs > $O/b_${v:-default}_$rep.json 2> $O/b_${v:-default}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/b_${v:-default}_$rep.json')); print('skip fwd after %-18s rep $rep: %.3f ms/step' % ('${v:-(behind input)}', j['ms_per_step']))"
  done
done

#!/bin/bash
# round-5 final tree (_ab_r05/, commit 8acd0de) against the current tree on ONE box: staged (headline) and resident legs, interleaved
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_m
mkdir -p $O
for rep in 1 2 3; do
  for tree in _ab_r05 .; do
    cd $R/$tree
    timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/b_${rep}_$(basename $(pwd)).json 2> $O/b_${rep}_$(basename $(pwd)).err
    python3 -c "
import json; j=json.load(open('$O/b_${rep}_$(basename $(pwd)).json')); print('%-8s rep $rep: staged %.3f ms/step  resident %.3f' % ('$tree', j['ms_per_step'], (j.get('resident') or {}).get('ms_per_step') or 0))"
  done
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_g
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in 0 1; do
    PICONS_PRIO=$v timeout 600 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing --resident-inputs > $O/bench_prio${v}_$rep.json 2> $O/bench_prio${v}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/bench_prio${v}_$rep.json')); print('prio=$v rep $rep: %.3f ms/step' % j['ms_per_step'])"
  done
done

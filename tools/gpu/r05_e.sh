#!/bin/bash
# Round 5, call E: BatchNorm finalize folded into the apply kernels (PICONS_BN_FUSED) -- tests and an A/B of the step
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_e
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "bn" > $O/test_bn.log 2>&1; echo "bn tests rc=$?"; tail -4 $O/test_bn.log
timeout 1200 python3 -m pytest tests/test_step_gpu.py -x -q -s -k "early_adam or trajectory" > $O/test_traj.log 2>&1; echo "traj rc=$?"; grep -a "early Adam vs\|passed\|failed" $O/test_traj.log | tail -5
for rep in 1 2 3; do
  for f in 0 1; do
    PICONS_BN_FUSED=$f timeout 600 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/bench_f${f}_$rep.json 2> $O/bench_f${f}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/bench_f${f}_$rep.json')); print('bn_fused=$f rep $rep: staged %.3f ms/step  resident %.3f  loss %.6f' % (j['ms_per_step'], j['resident']['ms_per_step'], j['loss']['total']))"
  done
done
timeout 1500 python3 -m pytest tests/test_step_gpu.py -x -q -k "not trajectory and not early_adam" > $O/test_step.log 2>&1; echo "step tests rc=$?"; tail -3 $O/test_step.log

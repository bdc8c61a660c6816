#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_e
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_wgrad_ordered_gpu.py tests/test_kernels_gpu.py -q -x > $O/t_ordered.log 2>&1; echo "ordered+kernel tests rc=$?"; tail -3 $O/t_ordered.log
timeout 1200 python3 -m pytest tests/test_step_gpu.py -x -q -k "deterministic or golden_full_size or small" > $O/t_step.log 2>&1; echo "step tests rc=$?"; tail -3 $O/t_step.log
for rep in 1 2 3; do
  for lib in libpicons_base.so libpicons.so; do
    PICONS_LIB_NAME=$lib timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --resident-inputs --no-kernel-timing > $O/b_${lib}_$rep.json 2> $O/b_${lib}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/b_${lib}_$rep.json')); print('%-20s rep $rep: %.3f ms/step' % ('$lib', j['ms_per_step']))"
  done
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_b
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_wgrad_ordered_gpu.py -q > $O/t_ordered.log 2>&1; echo "ordered tests rc=$?"; tail -8 $O/t_ordered.log
timeout 2400 python3 -m pytest tests/test_step_gpu.py -x -q > $O/t_step.log 2>&1; echo "step tests rc=$?"; tail -5 $O/t_step.log

#!/bin/bash
# numerics with the 28 x 28 layers in F(4x4, 3x3) too
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4n
mkdir -p $O
cd $R
export PICONS_WINO4_MIN_TILES=49
timeout 2400 python3 -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py tests/test_bench_gpu.py -q > $O/pytest_step.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_step.log | cut -c1-300
timeout 900 python3 tools/probe_grad_margin.py > $O/grad_margin.txt 2>&1; grep -A1 "WINO4=1" $O/grad_margin.txt | cut -c1-330; grep -B1 "WINO4=1" $O/grad_margin.txt | grep "vs fp64" | cut -c1-330
cp gpurun_out/trajectory_default.json $O/ 2>/dev/null

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
mkdir -p gpurun_out/r06_aj
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu -k "deterministic" 2>&1 | tee gpurun_out/r06_aj/pytest.log | tail -5

#!/bin/bash
# the driver's GPU tier: pytest -m gpu, then smoke()
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_tests
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"
tail -15 $O/pytest_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log

#!/bin/bash
# the whole GPU tier as the driver runs it
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_i
mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"
tail -12 $O/pytest_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_wino4_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout 600 python3 tools/bench_wino4.py 10 > $O/bench_wino4.txt 2>&1; cat $O/bench_wino4.txt

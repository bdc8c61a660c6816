#!/bin/bash
# Round 4, call A: the bf16-split conv kernel's own tests.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_a
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_x6_gpu.py -x -q > $O/x6_tests.log 2>&1; echo "rc=$?" >> $O/x6_tests.log
tail -40 $O/x6_tests.log

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_f
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_x6_gpu.py tests/test_wgrad_ordered_gpu.py -x -q -k "stem or ordered" > $O/t.log 2>&1; echo "tests rc=$?"; tail -5 $O/t.log
timeout 300 python3 tools/bench_stem_wgrad.py 2>&1 | tail -6

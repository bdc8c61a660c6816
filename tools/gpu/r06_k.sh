#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
echo "== product"; python3 tools/bench_wgrad_x6.py 2>&1 | grep thw
echo "== A split replaced by one cvt (WRONG results: sizing only)"; PICONS_HACK_NOSPLIT=1 python3 tools/bench_wgrad_x6.py 2>&1 | grep thw
echo "== A and B splits replaced"; PICONS_HACK_NOSPLIT=2 python3 tools/bench_wgrad_x6.py 2>&1 | grep thw

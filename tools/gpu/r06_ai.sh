#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
mkdir -p gpurun_out/r06_ai
timeout 600 python3 tools/host_time_lists.py 2>&1 | tee gpurun_out/r06_ai/host_time_lists.txt | tail -4

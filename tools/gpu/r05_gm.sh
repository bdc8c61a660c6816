#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4s
mkdir -p $O
cd $R
timeout 900 python3 tools/probe_grad_margin.py > $O/grad_margin.txt 2>&1; cat $O/grad_margin.txt

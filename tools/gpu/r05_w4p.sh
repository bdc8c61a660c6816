#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4
mkdir -p $O
cd $R
PROBE_VARS=${PROBE_VARS:-0,32} PICONS_LIB_NAME=libpicons_dg.so PROBE_M=4 timeout 900 python3 tools/probe_wino.py > $O/probe_m4.txt 2>&1; cat $O/probe_m4.txt

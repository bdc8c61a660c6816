#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_p
mkdir -p $O
cd $R
PICONS_STAGE_STREAM_PRIO=0 timeout 900 python3 tools/probe_leg_order.py > $O/leg_order_p0.txt 2>&1; grep "ms/step" $O/leg_order_p0.txt
PICONS_STAGE_STREAM_PRIO=0 timeout 900 python3 tools/probe_leg_order.py resident resident dicts resident staged resident > $O/leg_order_p0b.txt 2>&1; grep "ms/step" $O/leg_order_p0b.txt

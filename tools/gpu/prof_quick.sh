#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
# quick look: kernel stats at 1 and 4 lanes, launch table, lane timeline  ->  gpurun_out/pq/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pq
rm -rf $O; mkdir -p $O
cd /tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- $B > $O/prof.log 2>&1
PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l1 -o l1 -- $B > $O/prof_l1.log 2>&1
cd $R
python3 tools/launch_table.py $O/prof_l1/l1_kernel_trace.csv 70 > $O/launch_table.txt 2>&1
python3 tools/lane_timeline.py $O/prof/r03_kernel_trace.csv > $O/lane_timeline.txt 2>&1
python3 tools/lane_timeline.py $O/prof/r03_kernel_trace.csv --window -2 --by-lane 40 --sequence 0 > $O/lane0.txt 2>&1
tail -5 $O/launch_table.txt

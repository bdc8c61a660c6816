#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
# kernel stats (4 lanes) + single-lane trace for the launch table
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- $B > $O/prof.log 2>&1
PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l1 -o l1 -- $B > $O/prof_l1.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/launch_table.py $O/prof_l1/l1_kernel_trace.csv 60 > $O/launch_table.txt 2>&1
python3 tools/lane_timeline.py $O/prof/r03_kernel_trace.csv > $O/lane_timeline.txt 2>&1
head -30 $O/prof_l1/l1_kernel_stats.csv | cut -c1-150
tail -8 $O/launch_table.txt

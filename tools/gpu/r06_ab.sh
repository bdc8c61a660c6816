#!/bin/bash
# four-lane kernel trace of the current tree + lane 0's idle intervals and every lane's dispatch sequence
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_ab
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- $B > $O/prof.log 2>&1
cd $R
python3 tools/lane_timeline.py $O/prof/t_kernel_trace.csv --window -1 --by-lane 6 --gaps 30 --lane-gaps 0 > $O/lane_timeline.txt 2>&1
for l in 0 1 2 3; do python3 tools/lane_timeline.py $O/prof/t_kernel_trace.csv --window -1 --sequence $l 2>/dev/null | awk '/in order: start ms/{f=1} f' > $O/seq_$l.txt; done
grep -n "intervals of more than 4 us" -A 30 $O/lane_timeline.txt

#!/bin/bash
# kernel traces of the current code: 4 lanes (timeline) and 1 lane (launch table)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_d
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err
MS=$(python3 -c "import json; print(json.load(open('$O/bench.json'))['resident']['ms_per_step'])")
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r06 -- $B > $O/prof.log 2>&1
PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l1 -o l1 -- $B > $O/prof_l1.log 2>&1
cd $R
python3 tools/launch_table.py $O/prof_l1/l1_kernel_trace.csv 80 > $O/launch_table.txt 2>&1
python3 tools/lane_timeline.py $O/prof/r06_kernel_trace.csv --window -1 --expect-ms $MS --by-lane 6 > $O/lane_timeline.txt 2>&1
cp $O/prof/r06_kernel_stats.csv $O/r06_kernel_stats.csv; cp $O/prof_l1/l1_kernel_stats.csv $O/r06_l1_kernel_stats.csv
rm -rf $O/prof/*kernel_trace.csv.bak
tail -8 $O/launch_table.txt; head -30 $O/lane_timeline.txt
python3 -c "
import json; j=json.load(open('$O/bench.json')); print(j['ms_per_step'], j['resident']['ms_per_step'], j['roofline']['kernel'][:40], j['roofline']['kernel_ms_per_step'], j['roofline']['frac'])
for k in ('roofline_conv_x6','roofline_fp32_conv','roofline_winograd','roofline_wgrad_x6','roofline_wgrad_fp32'): print(k, j[k]['kernel_ms_per_step'], j[k]['launches_per_step'], j[k]['frac'])"

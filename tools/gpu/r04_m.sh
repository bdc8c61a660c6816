#!/bin/bash
# Round 4, call M: where the staged leg's extra 0.4 - 0.5 ms over the resident step goes: host time per phase, and a four-lane kernel trace of the
# staged run (lane timeline: which kernels of the input preparation sit on which stream, what runs at the head of a step).
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_m
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_inputpipe.py tests/test_bench_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 600 python3 tools/host_time_staged.py > $O/host_time.txt 2>&1; tail -2 $O/host_time.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/prof_staged -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $O/prof_staged.log 2>&1)
python3 tools/lane_timeline.py $O/prof_staged/s_kernel_trace.csv --window -2 --by-lane 8 > $O/lane_timeline_staged.txt 2>&1; head -30 $O/lane_timeline_staged.txt
for i in 1 2; do
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $O/bench_staged_$i.json 2> $O/bench_staged_$i.err
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/bench_res_$i.json 2> $O/bench_res_$i.err
python3 -c "import json; a=json.load(open('$O/bench_staged_$i.json')); b=json.load(open('$O/bench_res_$i.json')); print('staged %.3f  resident %.3f ms/step' % (a['ms_per_step'], b['ms_per_step']))"
done

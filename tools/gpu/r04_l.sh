#!/bin/bash
# Round 4, call L: tail split of the bf16-split conv kernel (K slices for the tiles of the last, partly filled round), waves without real rows
# skip their MFMAs: kernel tests, step tests, A/B on the same box.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_l
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_x6_gpu.py tests/test_dp_gpu.py -x -q -k "x6 or early_adam" > $O/x6_tests.log 2>&1; echo "rc=$?" >> $O/x6_tests.log; tail -6 $O/x6_tests.log
timeout 1500 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or small or bs8 or trajectory or determin" > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -4 $O/step_tests.log
for rep in 1 2; do
  for ts in 1 0; do
    PICONS_X6_TAIL_SPLIT=$ts timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/bench_ts${ts}_$rep.json 2> $O/bench_ts${ts}_$rep.err
    python3 -c "import json; j=json.load(open('$O/bench_ts${ts}_$rep.json')); print('tail split=$ts rep $rep: %.3f ms/step  %.1f clips/s' % (j['ms_per_step'], j['value']))"
  done
done
for ts in 1 0; do
  (cd /tmp && PICONS_X6_TAIL_SPLIT=$ts PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ts$ts -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_ts$ts.log 2>&1)
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_ts$ts/p_kernel_stats.csv")))
tot = 0
for r in rows:
    if "conv_x6" in r["Name"]:
        ms = float(r["TotalDurationNs"]) / 1e6 / 6; tot += ms
        print("tail split=$ts %8.3f ms/step %4d calls %s" % (ms, int(r["Calls"]) // 6, r["Name"][:90]))
print("tail split=$ts conv_x6 total %.3f ms/step" % tot)
PY
done
python3 tools/compare_conv_launches.py $O/prof_ts1/p_kernel_trace.csv $O/prof_ts0/p_kernel_trace.csv 6 > $O/tail_split_launches.txt 2>&1; tail -3 $O/tail_split_launches.txt

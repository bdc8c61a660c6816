#!/bin/bash
# Round 4, call H: bf16-split row-segment weight gradient: numerics, step tests, single-lane kernel stats, bench.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_h
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_x6_gpu.py -x -q > $O/x6_tests.log 2>&1; echo "rc=$?" >> $O/x6_tests.log; tail -6 $O/x6_tests.log
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "trajectory or golden or small or bs8" > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -4 $O/step_tests.log
for wg in 1 0; do
  (cd /tmp && PICONS_SPLIT_WGRAD=$wg PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wg$wg -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_wg$wg.log 2>&1)
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_wg$wg/p_kernel_stats.csv")))
tot = 0
for r in rows:
    if "wgrad" in r["Name"]:
        ms = float(r["TotalDurationNs"]) / 1e6 / 6; tot += ms
        print("wg=$wg %8.3f ms/step %4d calls %s" % (ms, int(r["Calls"]) // 6, r["Name"][:90]))
print("wg=$wg wgrad total %.3f ms/step" % tot)
PY
done
for wg in 1 0; do
  PICONS_SPLIT_WGRAD=$wg timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/bench_wg$wg.json 2> $O/bench_wg$wg.err
  python3 -c "import json; j=json.load(open('$O/bench_wg$wg.json')); print('split wgrad $wg: %.3f ms/step  %.1f clips/s  loss %.6f' % (j['ms_per_step'], j['value'], j['loss']['total']))"
done

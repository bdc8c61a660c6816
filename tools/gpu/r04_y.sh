#!/bin/bash
# Round 4, call Y: the stem weight gradient's kt slices interleaved in block order (D rows shared through L2) against kt-major order
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_y
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -q -k "wgrad or stem" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for il in 1 0; do
  (cd /tmp && PICONS_WGRAD_STEM_INTERLEAVE=$il PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_il$il -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_il$il.log 2>&1)
  (cd /tmp && PICONS_WGRAD_STEM_INTERLEAVE=$il PICONS_LANES=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_il$il -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/pmc_il$il.log 2>&1)
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/prof_il$il/p_kernel_stats.csv")):
    if "wgrad4" in r["Name"]: print("interleave=$il wgrad4_kernel %.1f us per launch" % (float(r["AverageNs"]) / 1e3))
n = v = 0
for r in csv.DictReader(open("$O/pmc_il$il/f_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE" and "wgrad4" in r["Kernel_Name"]: n += 1; v += float(r["Counter_Value"])
print("interleave=$il wgrad4_kernel fetches %.0f MB per launch (2 x FETCH_SIZE)" % (2 * v * 1024 / max(n, 1) / 1e6))
PY
done
for i in 1 2; do for il in 1 0; do
PICONS_WGRAD_STEM_INTERLEAVE=$il timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/bench_il${il}_$i.json 2> $O/bench_il${il}_$i.err
python3 -c "import json; a=json.load(open('$O/bench_il${il}_$i.json')); print('interleave=$il rep $i: %.3f ms/step' % a['ms_per_step'])"
done; done
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or bs8" > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -2 $O/step_tests.log

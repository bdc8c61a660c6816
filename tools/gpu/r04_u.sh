#!/bin/bash
# Round 4, call U: bf16-split conv only for launches of >= 5 K chunks (PICONS_X6_KMIN=5) against no K rule (=1), same box
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_u
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2 3; do for km in 5 1; do
PICONS_X6_KMIN=$km timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/bench_km${km}_$i.json 2> $O/bench_km${km}_$i.err
python3 -c "import json; a=json.load(open('$O/bench_km${km}_$i.json')); print('kmin=$km rep $i: %.3f ms/step' % a['ms_per_step'])"
done; done
for km in 5 1; do
  (cd /tmp && PICONS_X6_KMIN=$km PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_km$km -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_km$km.log 2>&1)
  python3 - <<PY
import csv
tot = 0
for r in csv.DictReader(open("$O/prof_km$km/p_kernel_stats.csv")):
    if "conv_x6" in r["Name"] or "conv_gemm" in r["Name"]: tot += float(r["TotalDurationNs"]) / 1e6 / 6
print("kmin=$km conv / dgrad (both kernels) %.3f ms/step single-lane" % tot)
PY
done
timeout 1500 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or bs8 or determin or trajectory or small or reference_init" > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -3 $O/step_tests.log

#!/bin/bash
# Round 4, call R: Winograd kernel's output through LDS as row-contiguous 16-byte stores against the four-byte stores (same box)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_r
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -q -k "wino" > $O/wino_tests.log 2>&1; echo "rc=$?" >> $O/wino_tests.log; tail -3 $O/wino_tests.log
timeout 1500 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or bs8 or determin or trajectory or small" > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -3 $O/step_tests.log
for se in 0 1; do
  (cd /tmp && PICONS_WINO_SCALAR_EPI=$se PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_se$se -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_se$se.log 2>&1)
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/prof_se$se/p_kernel_stats.csv")):
    if "wino_conv" in r["Name"]: print("scalar_epi=$se wino_conv_kernel %.3f ms/step over %d launches" % (float(r["TotalDurationNs"]) / 1e6 / 6, int(r["Calls"]) // 6))
PY
done
for i in 1 2; do for se in 0 1; do
PICONS_WINO_SCALAR_EPI=$se timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/bench_se${se}_$i.json 2> $O/bench_se${se}_$i.err
python3 -c "import json; a=json.load(open('$O/bench_se${se}_$i.json')); print('scalar_epi=$se rep $i: %.3f ms/step' % a['ms_per_step'])"
done; done

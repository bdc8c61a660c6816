#!/bin/bash
# Round 4, call K: generic bf16-split weight gradient with / without the register pipeline on the same box (alternating), the new
# early-Adam-under-the-reducer test, the whole GPU suite with the trunk's forward convs back on the fp32 kernel.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_k
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -8 $O/tests.log
for rep in 1 2; do
  for np in 0 1; do
    PICONS_WGRAD_X6_NOPIPE=$np timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/bench_np${np}_$rep.json 2> $O/bench_np${np}_$rep.err
    python3 -c "import json; j=json.load(open('$O/bench_np${np}_$rep.json')); print('nopipe=$np rep $rep: %.3f ms/step  %.1f clips/s' % (j['ms_per_step'], j['value']))"
  done
done
for np in 0 1; do
  (cd /tmp && PICONS_WGRAD_X6_NOPIPE=$np PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_np$np -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_np$np.log 2>&1)
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_np$np/p_kernel_stats.csv")))
tot = 0
for r in rows:
    if "wgrad" in r["Name"]:
        ms = float(r["TotalDurationNs"]) / 1e6 / 6; tot += ms
        print("nopipe=$np %8.3f ms/step %4d calls %s" % (ms, int(r["Calls"]) // 6, r["Name"][:90]))
print("nopipe=$np wgrad total %.3f ms/step" % tot)
PY
done

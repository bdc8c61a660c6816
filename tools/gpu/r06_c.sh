#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== TALL=0 (64x128 / 128x64 tiles, mfast)"; PICONS_X6_TALL=0 python3 tools/bench_spectral_x6.py --save /tmp/ref.pt 2>&1 | tail -3
echo "== TALL=0 MFAST=0"; PICONS_X6_TALL=0 PICONS_X6_MFAST=0 python3 tools/bench_spectral_x6.py 2>&1 | tail -2
echo "== TALL=1"; PICONS_X6_TALL=1 python3 tools/bench_spectral_x6.py --check /tmp/ref.pt 2>&1 | tail -4
cd /tmp
for t in 0 1; do
  PICONS_X6_TALL=$t REPS=3 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f$t -o f -- python3 $R/tools/bench_spectral_x6.py > $O/pmc_f$t.log 2>&1
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/pmc_f$t/f_counter_collection.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == "FETCH_SIZE" and "conv_x6" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:60], r.get("Grid_Size"))].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("TALL=$t", k, "launches", len(v), "FETCH_SIZE avg %.0f (x64 B = %.1f MB; x32 B = %.1f MB)" % (sum(v)/len(v), sum(v)/len(v)*64/1e6, sum(v)/len(v)*32/1e6))
PY
done

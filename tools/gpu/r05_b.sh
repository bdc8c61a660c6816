#!/bin/bash
# Round 5, call B: the Winograd kernel's raw-patch LDS layout (even columns first, lane = (tile, channel half): conflict-free transform reads)
# against the previous build (libpicons_base.so = HEAD's build) on one box: tests, per-layer times, LDS conflict counters.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_b
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_wino_gpu.py -x -q > $O/test_wino.log 2>&1; echo "wino tests rc=$?"; tail -3 $O/test_wino.log
for rep in 1 2; do
  PICONS_LIB_NAME=libpicons_base.so timeout 300 python3 tools/bench_wino.py 20 > $O/wino_base_$rep.txt 2>&1
  timeout 300 python3 tools/bench_wino.py 20 > $O/wino_new_$rep.txt 2>&1
done
paste -d'\n' $O/wino_base_2.txt $O/wino_new_2.txt | sed 's/direct .*winograd/winograd/' | cut -c1-150
export TMPDIR=/tmp
cd /tmp
for v in base new; do
  L=""; [ $v = base ] && L="libpicons_base.so"
  PICONS_LIB_NAME=$L timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$v -o p -- python3 $R/tools/bench_wino.py 3 > $O/pmc_$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for v in ("base", "new"):
    f = glob.glob("gpurun_out/r05_b/pmc_%s/**/p_counter_collection.csv" % v, recursive=True)
    if not f: print(v, "no counter file"); continue
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if "wino_conv_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(v, {k: "%.3g" % x for k, x in acc.items()}, "conflict / active = %.3f" % (acc["SQ_LDS_BANK_CONFLICT"] / max(acc["SQ_LDS_IDX_ACTIVE"], 1)))
PY

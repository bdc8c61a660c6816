#!/bin/bash
# A/B on one box: the tree at the previous commit (_ab_prev) against this tree, interleaved; then this tree's kernel stats
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_z
mkdir -p $O
cd $R; timeout 900 python3 -m pytest tests/test_wgrad_ordered_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "tail" > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -5 $O/pytest_tail.log
for i in 1 2 3; do
  for t in prev cur; do
    if [ $t = prev ]; then cd $R/_ab_prev; else cd $R; fi
    timeout 600 python3 bench.py > $O/bench_${t}_$i.json 2> $O/bench_${t}_$i.err
    python3 - <<PY
import json
d = json.loads(open("$O/bench_${t}_$i.json").read().strip().splitlines()[-1])
print("$t", $i, round(d["ms_per_step"], 3), d["config"].get("resident_ms_per_step"), d["config"].get("staged_minus_resident_ms"))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o cur -- python3 bench.py --steps 10 --warmup 3 > $O/prof_bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/prof/**/cur_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    if any(k in r["Name"] for k in ("tail", "colreduce", "slices_fold", "transpose_multi")):
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY

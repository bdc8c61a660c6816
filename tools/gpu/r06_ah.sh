#!/bin/bash
# is lane 0 waiting for the HOST at the head of a step?  kernel trace + HIP runtime trace of two resident steps: launch call time against kernel start
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_ah
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/prof -o t -- $B > $O/prof.log 2>&1
ls -la $O/prof | head
cd $R
python3 - <<PY
import csv, glob
k = [r for r in csv.DictReader(open("$O/prof/t_kernel_trace.csv")) if r["Kind"] == "KERNEL_DISPATCH"]
api = {r["Correlation_Id"]: r for r in csv.DictReader(open("$O/prof/t_hip_api_trace.csv"))}
for r in k:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
k.sort(key=lambda r: r["s"])
nd = [r for r in k if "to_ndhwc" in r["Kernel_Name"]]
heads = [r for i, r in enumerate(nd) if i == 0 or r["s"] - nd[i - 1]["s"] > 2_000_000]
h = heads[-4]
t0 = h["s"]
print("columns:", list(api[h["Correlation_Id"]].keys()))
n = 0
for r in k:
    if t0 - 400_000 <= r["s"] <= t0 + 1_500_000:
        a = api.get(r["Correlation_Id"])
        call = int(a["Start_Timestamp"]) if a else None
        print("%s stream %s  start %8.1f us  dur %7.1f  launch call at %s us  %s" % ("*" if r is h else " ", r["Stream_Id"], (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3,
              "%9.1f" % ((call - t0) / 1e3) if call else "?", r["Kernel_Name"][:50].replace("(anonymous namespace)::", "")))
        n += 1
        if n > 70: break
PY

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_m
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs"
PICONS_LANES=1 PICONS_LIB_NAME=libpicons_base.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p64 -o l1 -- $B > $O/p64.log 2>&1
PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p192 -o l1 -- $B > $O/p192.log 2>&1
PICONS_LANES=1 PICONS_BN_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/poff -o l1 -- $B > $O/poff.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
for v in ("p64", "p192", "poff"):
    f = glob.glob("gpurun_out/r05_m/%s/**/l1_kernel_stats.csv" % v, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    sel = [r for r in rows if "bn_" in r["Name"] or "colreduce_kernel<0>" in r["Name"]]
    print(v, "BatchNorm family: %d launches / step, %.3f ms / step;" % (sum(int(r["Calls"]) for r in sel) // 3, sum(float(r["TotalDurationNs"]) for r in sel) / 3e6),
          "  ".join("%s %.3f" % (r["Name"].split("(")[0].split("::")[-1][:24], float(r["TotalDurationNs"]) / 3e6) for r in sel[:6]))
PY
for rep in 1 2; do
  for cfg in "libpicons_base.so 1" "libpicons.so 1" "libpicons.so 0"; do
    set -- $cfg
    PICONS_LIB_NAME=$1 PICONS_BN_FUSED=$2 timeout 600 python3 bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing --resident-inputs > $O/b.json 2> $O/b.err
    python3 -c "
import json; j=json.load(open('$O/b.json')); print('$1 fused=$2 rep $rep: %.3f ms/step' % j['ms_per_step'])"
  done
done

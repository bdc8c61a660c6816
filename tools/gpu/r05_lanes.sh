#!/bin/bash
# lane-assignment switches re-measured after the F(4x4,3x3) kernel shortened the skip lane's convs (interleaved, two rounds, one box)
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_lanes
mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-kernel-timing"
for i in 1 2; do
  for v in default PICONS_WGRAD_SEPARATE=0 PICONS_FWD_BRANCH3=0 PICONS_CONV28_ASIDE=0 PICONS_SPREAD_CLASSES=0 PICONS_WGRAD_MULTI=1 PICONS_LATE_PREP=0 PICONS_SKIP_LANE=0 PICONS_DEFER_SIDE=1; do
    if [ "$v" = default ]; then timeout 300 $B > $O/${v}_$i.json 2>> $O/bench.err; else env $v timeout 300 $B > $O/${v}_$i.json 2>> $O/bench.err; fi
  done
done
python3 - <<'PY'
import json, glob, collections
d = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05_lanes/*.json")):
    try:
        d[f.split("/")[-1].rsplit("_", 1)[0]].append(json.load(open(f))["ms_per_step"])
    except Exception as e:
        d[f.split("/")[-1]].append(float("nan"))
for k, v in sorted(d.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print("%-28s %s" % (k, "  ".join("%.3f" % x for x in v)))
PY

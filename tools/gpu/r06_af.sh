#!/bin/bash
# the early Adam op behind the stem BatchNorm backward (PICONS_EARLY_ADAM_BEHIND_BN=1, default: beside the stem weight gradient) against in front of it (=0)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_af
mkdir -p $O
cd $R
for i in 1 2 3; do
  for v in 0 1; do
    PICONS_EARLY_ADAM_BEHIND_BN=$v timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python3 - <<PY
import json
d = json.loads(open("$O/bench_${v}_$i.json").read().strip().splitlines()[-1])
print("adam_behind_bn=$v", $i, round(d["ms_per_step"], 3), round((d.get("resident") or {}).get("ms_per_step", 0), 3))
PY
  done
done
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- $B > $O/prof.log 2>&1
cd $R
python3 tools/lane_timeline.py $O/prof/t_kernel_trace.csv --window -1 --by-lane 6 --gaps 30 --lane-gaps 0 > $O/lane_timeline.txt 2>&1
for l in 0 1 2 3; do python3 tools/lane_timeline.py $O/prof/t_kernel_trace.csv --window -1 --sequence $l 2>/dev/null | awk '/in order: start ms/{f=1} f' > $O/seq_$l.txt; done
grep -n "intervals of more than 4 us" -A 14 $O/lane_timeline.txt

#!/bin/bash
# wspec_master_bwd with nine times shorter blocks: parity, A/B against the side tree (_ab_prev = this tree with the previous kernel), trace
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_ac
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_x6_gpu.py -x -q -m gpu -k "spectral or primary or wspec or x6" > $O/pytest_spec.log 2>&1; echo "spectral tests rc=$?"; tail -3 $O/pytest_spec.log
for i in 1 2 3; do
  for t in prev cur0 cur; do
    if [ $t = prev ]; then cd $R/_ab_prev; else cd $R; fi
    if [ $t = cur0 ]; then export PICONS_X6_BIG_TILE_MIN=0; else unset PICONS_X6_BIG_TILE_MIN; fi
    timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs > $O/bench_${t}_$i.json 2> $O/bench_${t}_$i.err
    python3 - <<PY
import json
d = json.loads(open("$O/bench_${t}_$i.json").read().strip().splitlines()[-1])
print("$t", $i, round(d["ms_per_step"], 3), round((d.get("resident") or {}).get("ms_per_step", 0), 3))
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
grep -n "intervals of more than 4 us" -A 12 $O/lane_timeline.txt
grep "wspec_master_bwd" $O/prof/t_kernel_stats.csv | cut -c1-160

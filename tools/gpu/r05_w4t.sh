#!/bin/bash
# trunk forward (Conv3d_2c) in F(2x2,3x3), everything else as before: gradient margins, the failing golden test, step time against TRUNK_FWD=1
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4t
mkdir -p $O
cd $R
timeout 900 python3 tools/probe_grad_margin.py > $O/grad_margin.txt 2>&1; grep -v amdgpu $O/grad_margin.txt | cut -c1-330
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-kernel-timing"
for i in 1 2; do
  timeout 300 $B > $O/bench_default_$i.json 2>> $O/bench.err
  PICONS_WINO4_TRUNK_FWD=1 timeout 300 $B > $O/bench_trunkfwd4_$i.json 2>> $O/bench.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_w4t/bench_*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], "%.3f ms  %.1f clips/s" % (j["ms_per_step"], j["value"]))
PY

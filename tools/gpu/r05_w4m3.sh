#!/bin/bash
# F(4x4,3x3) on the two-frame 28 x 28 layers (Mixed_3b / 3c) only: numerics tests that failed with all 28 x 28 layers in it, and the step
# (PICONS_WINO4_MIN_T was a one-off switch of Plan.wino_m for this run -- layers with fewer frames stayed in F(2x2,3x3) -- and is not in the tree)
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4m3
mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-kernel-timing"
for i in 1 2; do
  timeout 300 $B > $O/bench_default_$i.json 2>> $O/bench.err
  PICONS_WINO4_MIN_TILES=49 PICONS_WINO4_MIN_T=2 timeout 300 $B > $O/bench_m3_$i.json 2>> $O/bench.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_w4m3/bench_*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], "%.3f ms  %.1f clips/s" % (j["ms_per_step"], j["value"]))
PY
export PICONS_WINO4_MIN_TILES=49 PICONS_WINO4_MIN_T=2
timeout 2400 python3 -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -q > $O/pytest_step.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_step.log | cut -c1-200

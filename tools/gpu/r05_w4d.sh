#!/bin/bash
# the 28 x 28 layers' INPUT GRADIENTS in F(4x4,3x3) (their forwards stay in F(2x2,3x3) with the trunk): step time, gradient margins, parity tests
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4d
mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-kernel-timing"
for i in 1 2 3; do
  timeout 300 $B > $O/bench_default_$i.json 2>> $O/bench.err
  PICONS_WINO4_MIN_TILES=49 timeout 300 $B > $O/bench_dgrad28_$i.json 2>> $O/bench.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_w4d/bench_*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], "%.3f ms  %.1f clips/s" % (j["ms_per_step"], j["value"]))
PY
export PICONS_WINO4_MIN_TILES=49
timeout 900 python3 tools/probe_grad_margin.py 2>&1 | grep -v amdgpu | grep -A1 "WINO4=1" | cut -c1-300
timeout 2400 python3 -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -q > $O/pytest_step.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_step.log | cut -c1-200

#!/bin/bash
# F(4x4, 3x3) in the step: kernel tests, A/B of the bench line (PICONS_WINO4=0 / default / 28x28 layers too), interleaved, then the whole GPU tier
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w4s
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_wino4_gpu.py tests/test_wino_gpu.py -x -q > $O/pytest_wino.log 2>&1; echo "pytest wino rc=$?"; tail -3 $O/pytest_wino.log
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-kernel-timing"
for i in 1 2; do
  PICONS_WINO4=0 timeout 300 $B > $O/bench_w4off_$i.json 2>> $O/bench.err
  timeout 300 $B > $O/bench_w4on_$i.json 2>> $O/bench.err
  PICONS_WINO4_MIN_TILES=49 timeout 300 $B > $O/bench_w4all_$i.json 2>> $O/bench.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_w4s/bench_*.json")):
    try:
        j = json.load(open(f)); print(f.split("/")[-1], "%.3f ms  %.1f clips/s" % (j["ms_per_step"], j["value"]))
    except Exception as e:
        print(f, "failed", e)
PY
timeout 600 python3 tools/bench_wino4.py 10 > $O/bench_wino4.txt 2>&1; cat $O/bench_wino4.txt
bash tools/gpu/r05_tests.sh

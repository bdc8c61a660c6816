#!/bin/bash
# F(2x2,3x3) kernel with its weight fragments straight from global memory: kernel tests, layer bench, step A/B against the previous build
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_w2b
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_wino4_gpu.py tests/test_wino_gpu.py -x -q > $O/pytest_wino.log 2>&1; echo "pytest wino rc=$?"; tail -3 $O/pytest_wino.log
PICONS_LIB_NAME=libpicons_base.so timeout 600 python3 tools/bench_wino.py 20 > $O/bench_wino_base.txt 2>&1
timeout 600 python3 tools/bench_wino.py 20 > $O/bench_wino_new.txt 2>&1
paste -d'\n' $O/bench_wino_base.txt $O/bench_wino_new.txt | grep -v amdgpu | cut -c1-150
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-kernel-timing"
for i in 1 2 3; do
  PICONS_LIB_NAME=libpicons_base.so timeout 300 $B > $O/bench_base_$i.json 2>> $O/bench.err
  timeout 300 $B > $O/bench_new_$i.json 2>> $O/bench.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_w2b/bench_*.json")):
    try:
        j = json.load(open(f)); print(f.split("/")[-1], "%.3f ms  %.1f clips/s" % (j["ms_per_step"], j["value"]))
    except Exception as e:
        print(f, "failed", e)
PY

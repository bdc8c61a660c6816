#!/bin/bash
# Round 4, call B: step parity with the bf16-split conv kernel in the plan + a first bench.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_step_gpu.py tests/test_x6_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
timeout 600 python3 bench.py --steps 60 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/bench_x6.json 2> $O/bench_x6.err; echo "rc=$?"
PICONS_SPLIT=0 timeout 600 python3 bench.py --steps 60 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "rc=$?"
python3 - <<'PY'
import json
for n in ("x6", "fp32"):
    try:
        j = json.load(open("gpurun_out/r04_b/bench_%s.json" % n))
        print(n, "ms/step %.3f" % j["ms_per_step"], "clips/s %.1f" % j["value"], "loss", j["loss"]["total"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -5 $O/bench_x6.err

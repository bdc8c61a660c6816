#!/bin/bash
# tail gradients without atomics: the new tests, the step determinism / data-parallel tests, bench
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_y
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_wgrad_ordered_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "tail" > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -15 $O/pytest_tail.log
timeout 1500 python3 -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -x -q -m gpu > $O/pytest_step.log 2>&1; echo "step tests rc=$?"; tail -8 $O/pytest_step.log
for i in 1 2; do timeout 600 python3 bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench rc=$?"; python3 - <<PY
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], {k: v for k, v in d["config"].items() if "ms" in k})
PY
done

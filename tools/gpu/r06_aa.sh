#!/bin/bash
# Inception forward: the pool branch forked at the module's head (PICONS_POOL_BRANCH_EARLY=1, default) against behind the fused 1x1x1 unit (=0)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_aa
mkdir -p $O
cd $R
for i in 1 2 3; do
  for v in 0 1; do
    PICONS_POOL_BRANCH_EARLY=$v timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python3 - <<PY
import json
d = json.loads(open("$O/bench_${v}_$i.json").read().strip().splitlines()[-1])
print("early=$v", $i, round(d["ms_per_step"], 3), round((d.get("resident") or {}).get("ms_per_step", 0), 3))
PY
  done
done
timeout 1500 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu > $O/pytest_step.log 2>&1; echo "step tests rc=$?"; tail -3 $O/pytest_step.log

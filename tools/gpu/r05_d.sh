#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_d
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_step_gpu.py -x -q -s -k "trajectory or early_adam" > $O/test_traj.log 2>&1; echo "traj rc=$?"; grep -a "early Adam vs\|passed\|failed" $O/test_traj.log | tail -5
python3 - <<'PY'
import json
for m in ("default", "reducer"):
    try:
        v = json.load(open("gpurun_out/trajectory_%s.json" % m))
        for t in ("adam_m_norms", "adam_v_norms"): print(m, t, {k: v[t][k] for k in ("worst_excess", "median_rel", "ref32_median_rel", "largest", "stem")})
    except Exception as e: print(m, e)
PY
timeout 1200 python3 -m pytest tests/test_x6_gpu.py tests/test_bench_gpu.py -x -q > $O/test_x6_bench.log 2>&1; echo "x6+bench rc=$?"; tail -3 $O/test_x6_bench.log

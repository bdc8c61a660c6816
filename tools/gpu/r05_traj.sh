#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r05_traj
mkdir -p $O
cd $R
for w in 0 1; do
  PICONS_WINO4=$w timeout 600 python3 tests/traj_worker.py default $O/traj_w4_$w.json > $O/traj_$w.log 2>&1; echo "traj W4=$w rc=$?"
done
python3 - <<'PY'
import json
for w in (0, 1):
    j = json.load(open("gpurun_out/r05_traj/traj_w4_%d.json" % w))
    print("W4=%d param_norm_excess %.3e" % (w, j["param_norm_excess"]))
    for q in j["param_norm_excess_top"][:5]:
        print("    %-44s excess %.2e  d_vs_f64 %.2e  ref32 %.2e  norm %.3e" % (q["name"], q["excess"], q["d_vs_f64"], q["ref32_d_vs_f64"], q["norm"]))
    print("    step3 loss vs f64", j["steps"][2]["loss_vs_f64"]["total"], "ref32", j["steps"][2]["ref32_vs_f64"]["total"])
PY

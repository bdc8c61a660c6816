#!/bin/bash
# Round 5, call C: the rest of the GPU suite behind the trajectory test, the new A-planes tests, and an A/B of PICONS_X6_APLANES
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_c
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_x6_gpu.py -x -q -k "presplit or stress" > $O/test_ap.log 2>&1; echo "ap tests rc=$?"; tail -5 $O/test_ap.log
timeout 1200 python3 -m pytest tests/test_step_gpu.py -x -q -k "trajectory" > $O/test_traj.log 2>&1; echo "traj rc=$?"; tail -5 $O/test_traj.log
python3 - <<'PY'
import json
for m in ("default", "reducer"):
    try:
        v = json.load(open("gpurun_out/trajectory_%s.json" % m)); print(m, v["adam_m_norms"], v["adam_v_norms"])
    except Exception as e: print(m, e)
PY
for rep in 1 2; do
  for ap in 0 1; do
    PICONS_X6_APLANES=$ap timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --resident-inputs > $O/bench_ap${ap}_$rep.json 2> $O/bench_ap${ap}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/bench_ap${ap}_$rep.json')); r=j['roofline_conv_x6']; print('aplanes=$ap rep $rep: %.3f ms/step  x6 conv %.3f ms (%d launches)  wino %.3f  f32 %.3f' % (j['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], j['roofline_winograd']['kernel_ms_per_step'], j['roofline_fp32_conv']['kernel_ms_per_step']))"
  done
done
timeout 2400 python3 -m pytest tests/ -x -q -m gpu --deselect tests/test_step_gpu.py::test_training_trajectory_vs_reference -k "not test_x6_gpu" > $O/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -5 $O/pytest_rest.log

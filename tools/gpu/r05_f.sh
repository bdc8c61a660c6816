#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_f
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_bench_gpu.py -x -q > $O/test_bench.log 2>&1; echo "bench tests rc=$?"; tail -4 $O/test_bench.log
for rep in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_$rep.json 2> $O/bench_driver_$rep.err
python3 - <<PY
import json
j = json.load(open("gpurun_out/r05_f/bench_driver_$rep.json"))
print("driver cmd rep $rep: %.3f ms/step = %.1f clips/s; resident %.3f; dict %.3f; split_off %.3f" % (j["ms_per_step"], j["value"], j["resident"]["ms_per_step"], j["dict_contract"]["ms_per_step"], j["split_off"]["ms_per_step"]))
for k in ("roofline", "roofline_conv_x6", "roofline_fp32_conv", "roofline_winograd"):
    r = j[k]; print(" ", k, r["kernel"].split(" ")[0], r.get("kernel_ms_per_step"), r.get("launches_per_step"), r.get("frac"), r.get("timed_steps"), r.get("invalid"))
print("  in-region x6:", j["roofline_conv_x6"]["in_region_kernel_ms_per_step"], "busy", j["busy_steps_outside_timed_regions"], "cpu", j["cpu_baseline"]["value"])
PY
done

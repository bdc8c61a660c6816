#!/bin/bash
# Round 5, call A: (1) the 16x16x32 arm of the bf16-split tile probe (VERDICT r4 #1a), (2) the driver's exact bench command with a per-launch dump of
# the event pairs behind the roofline legs (roofline_fp32_conv read 10.2 ms under --steps 20).
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DSTAMPS=1 tools/x6_tile_probe.hip -o $O/x6tile 2> $O/x6tile_build.log || { cat $O/x6tile_build.log; exit 1; }
{
for shp in "131072 256 1152" "12544 256 832"; do
  timeout 300 $O/x6tile $shp
done
} > $O/x6tile.txt 2>&1
grep -v "abl\|fp32 mfma\|4x2\|2x2\"" $O/x6tile.txt | head -60
PICONS_TIMED_DUMP=$O/timed_dump.txt timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/r05_a/bench_driver_cmd.json"))
print("ms/step", j["ms_per_step"], "resident", j["resident"]["ms_per_step"])
for k in ("roofline", "roofline_fp32_conv", "roofline_winograd"):
    r = j[k]; print(k, r["kernel_ms_per_step"], r["launches_per_step"], r["frac"], r["timed_steps"])
PY
awk '/^#/{n++; print; next} {s[n]+=$2; if ($2>0.5) print "  big", n, $0} END{for(i in s) print i, s[i]}' $O/timed_dump.txt | tail -40

#!/bin/bash
# the 64 x 128 bf16-split conv tile on two waves (32 rows x 128 columns each: every activation element split once per block) against 2 x 2 waves
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_al
mkdir -p $O
cd $R
for v in 0 1 0 1; do echo "== PICONS_X6_R64_WN1=$v"; PICONS_X6_R64_WN1=$v timeout 300 python3 tools/bench_spectral_x6.py 2>&1 | grep -v amdgpu.ids | tail -4; done | tee $O/micro.txt
PICONS_X6_R64_WN1=1 timeout 900 python3 -m pytest tests/test_x6_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "x6 or spectral or primary" 2>&1 | tail -2
for i in 1 2 3; do
  for v in 0 1; do
    PICONS_X6_R64_WN1=$v timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python3 - <<PY
import json
d = json.loads(open("$O/bench_${v}_$i.json").read().strip().splitlines()[-1])
print("wn1=$v", $i, round(d["ms_per_step"], 3), round((d.get("resident") or {}).get("ms_per_step", 0), 3))
PY
  done
done

#!/bin/bash
# pc_wspec_master_planes / pc_wspec_master_bwd with LDS-staged strided sides: microbench old library (libpicons_base.so) against new, parity test, step A/B
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_ak
mkdir -p $O
cd $R
echo "== old"; PICONS_LIB_NAME=libpicons_base.so timeout 300 python3 tools/bench_wspec_master.py 2>&1 | grep -v amdgpu.ids | tee $O/micro_old.txt
echo "== new"; timeout 300 python3 tools/bench_wspec_master.py 2>&1 | grep -v amdgpu.ids | tee $O/micro_new.txt
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "planes or spectral or primary" 2>&1 | tail -2
for i in 1 2 3; do
  for t in old new; do
    if [ $t = old ]; then export PICONS_LIB_NAME=libpicons_base.so; else unset PICONS_LIB_NAME; fi
    timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs > $O/bench_${t}_$i.json 2> $O/bench_${t}_$i.err
    python3 - <<PY
import json
d = json.loads(open("$O/bench_${t}_$i.json").read().strip().splitlines()[-1])
print("$t", $i, round(d["ms_per_step"], 3), round((d.get("resident") or {}).get("ms_per_step", 0), 3))
PY
  done
done

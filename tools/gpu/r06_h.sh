#!/bin/bash
# A/B of two library builds: libpicons_base.so against libpicons.so, with the per-family kernel times
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_h
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_x6_gpu.py tests/test_wgrad_ordered_gpu.py -x -q > $O/t.log 2>&1; echo "x6 + ordered tests rc=$?"; tail -2 $O/t.log
timeout 300 python3 tools/bench_stem_wgrad.py 2>&1 | tail -5
for rep in 1 2 3; do
  for lib in libpicons_base.so libpicons.so; do
    PICONS_LIB_NAME=$lib timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --resident-inputs > $O/b_${lib}_$rep.json 2> $O/b_${lib}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/b_${lib}_$rep.json')); g=lambda k:(j.get(k) or {}).get('kernel_ms_per_step') or 0; print('%-20s rep $rep: %.3f ms/step  wgx6 %.3f wgf32 %.3f x6 %.3f wino %.3f f32 %.3f' % ('$lib', j['ms_per_step'], g('roofline_wgrad_x6'), g('roofline_wgrad_fp32'), g('roofline_conv_x6'), g('roofline_winograd'), g('roofline_fp32_conv')))"
  done
done

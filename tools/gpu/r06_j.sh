#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_j
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_wino_gpu.py tests/test_wino4_gpu.py tests/test_wgrad_ordered_gpu.py -x -q > $O/t.log 2>&1; echo "wino tests rc=$?"; tail -6 $O/t.log
for rep in 1 2 3; do
  for st in 0 1; do
    PICONS_WINO_STRIPS=$st timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --resident-inputs > $O/b_st${st}_$rep.json 2> $O/b_st${st}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/b_st${st}_$rep.json')); g=lambda k:(j.get(k) or {}).get('kernel_ms_per_step') or 0; print('strips=$st rep $rep: %.3f ms/step  wino %.3f (frac %.3f issued %.3f) wgx6 %.3f x6 %.3f f32 %.3f' % (j['ms_per_step'], g('roofline_winograd'), j['roofline_winograd']['frac'], j['roofline_winograd']['frac_mfma_issued'], g('roofline_wgrad_x6'), g('roofline_conv_x6'), g('roofline_fp32_conv')))"
  done
done

#!/bin/bash
# round 6, first call: ordered split-K weight gradients -- kernel tests, step tests, A/B against the atomic epilogue
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_a
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_wgrad_ordered_gpu.py -x -q > $O/t_ordered.log 2>&1; echo "ordered tests rc=$?"; tail -5 $O/t_ordered.log
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_x6_gpu.py -x -q > $O/t_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/t_kernels.log
timeout 2400 python3 -m pytest tests/test_step_gpu.py -x -q > $O/t_step.log 2>&1; echo "step tests rc=$?"; tail -5 $O/t_step.log
for rep in 1 2 3; do
  for at in 1 0; do
    PICONS_WGRAD_ATOMIC=$at timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --resident-inputs > $O/b_at${at}_$rep.json 2> $O/b_at${at}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/b_at${at}_$rep.json')); g=lambda k:(j.get(k) or {}).get('kernel_ms_per_step'); print('atomic=$at rep $rep: %.3f ms/step  wgx6 %s wgf32 %s x6 %s wino %s f32 %s' % (j['ms_per_step'], g('roofline_wgrad_x6'), g('roofline_wgrad_fp32'), g('roofline_conv_x6'), g('roofline_winograd'), g('roofline_fp32_conv')))"
  done
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_k
mkdir -p $O
cd $R
python3 tools/probe_tail_bias.py 150 2>&1 | tail -4 | cut -c1-250
timeout 1200 python3 -m pytest tests/test_x6_gpu.py -x -q > $O/test_x6.log 2>&1; echo "x6 tests rc=$?"; tail -2 $O/test_x6.log
for rep in 1 2 3; do
  for lib in libpicons_base.so libpicons.so; do
    PICONS_LIB_NAME=$lib timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --resident-inputs > $O/b_${lib}_$rep.json 2> $O/b_${lib}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/b_${lib}_$rep.json')); r=j['roofline_conv_x6']; print('%-20s rep $rep: %.3f ms/step  x6 conv %.3f ms  wino %.3f  f32 %.3f' % ('$lib', j['ms_per_step'], r['kernel_ms_per_step'], j['roofline_winograd']['kernel_ms_per_step'], j['roofline_fp32_conv']['kernel_ms_per_step']))"
  done
done

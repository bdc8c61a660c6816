#!/bin/bash
# Round 4, call Q: zero-kernel commit of the sample stager (every reader of clip / mask / small inputs re-pointed), short-K launches back on the
# fp32 kernel: tests and the bench legs.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_q
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_inputpipe.py tests/test_bench_gpu.py tests/test_x6_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 1500 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or bs8 or determin or trajectory or small" > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -3 $O/step_tests.log
timeout 900 python3 tools/probe_leg_order.py resident staged resident staged dicts resident > $O/leg_order.txt 2>&1; grep "ms/step" $O/leg_order.txt
timeout 900 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "import json; a=json.load(open('$O/bench.json')); print('staged %.3f resident %.3f dict %.3f split_off %.3f  x6 frac %.3f (%d launches) f32 conv frac %.3f (%d)' % (a['ms_per_step'], a['resident']['ms_per_step'], a['dict_contract']['ms_per_step'], a['split_off']['ms_per_step'], a['roofline']['frac'], a['roofline']['launches_per_step'], a['roofline_fp32_conv']['frac'], a['roofline_fp32_conv']['launches_per_step']))"

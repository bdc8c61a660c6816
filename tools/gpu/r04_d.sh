#!/bin/bash
# Round 4, call D: bf16-split conv incl. the spectral GEMMs: tests, single-lane traces (split on / off / on without one-frame Winograd), bench.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_x6_gpu.py tests/test_kernels_gpu.py -x -q > $O/kernel_tests.log 2>&1; echo "rc=$?" >> $O/kernel_tests.log; tail -3 $O/kernel_tests.log
for cfg in "1 1" "0 1" "1 0"; do
  set -- $cfg
  (cd /tmp && PICONS_SPLIT=$1 PICONS_WINO_T1=$2 PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s$1_t$2 -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/prof_s$1_t$2.log 2>&1)
done
python3 tools/compare_conv_launches.py $O/prof_s1_t1/p_kernel_trace.csv $O/prof_s0_t1/p_kernel_trace.csv 6 > $O/x6_launches.txt 2>&1; tail -2 $O/x6_launches.txt
for cfg in "1 1" "0 1" "1 0"; do
  set -- $cfg
  PICONS_SPLIT=$1 PICONS_WINO_T1=$2 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/bench_s$1_t$2.json 2> $O/bench_s$1_t$2.err
  python3 -c "import json; j=json.load(open('$O/bench_s$1_t$2.json')); print('split $1 wino_t1 $2: %.3f ms/step  %.1f clips/s  loss %.6f' % (j['ms_per_step'], j['value'], j['loss']['total']))"
done
timeout 1500 python3 -m pytest tests/test_step_gpu.py -x -q > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -5 $O/step_tests.log
PICONS_WINO_T1=0 timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "bs8_full_size" > $O/step_tests_t0.log 2>&1; echo "rc=$?" >> $O/step_tests_t0.log; tail -3 $O/step_tests_t0.log

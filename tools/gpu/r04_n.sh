#!/bin/bash
# Round 4, call N: samples written in place in the first conv's layout (no stack / cat / gather / arena copy / layout conversion), row-tile-fastest
# block order for the weight-heavy spectral GEMMs: tests, staged against resident, block order A/B.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_n
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_inputpipe.py tests/test_bench_gpu.py tests/test_x6_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or bs8 or determin" > $O/step_tests.log 2>&1; echo "rc=$?" >> $O/step_tests.log; tail -3 $O/step_tests.log
timeout 600 python3 tools/host_time_staged.py > $O/host_time.txt 2>&1; tail -1 $O/host_time.txt
for i in 1 2; do
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $O/bench_staged_$i.json 2> $O/bench_staged_$i.err
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/bench_res_$i.json 2> $O/bench_res_$i.err
PICONS_X6_MFAST=0 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/bench_res_mf0_$i.json 2> $O/bench_res_mf0_$i.err
python3 -c "import json; a=json.load(open('$O/bench_staged_$i.json')); b=json.load(open('$O/bench_res_$i.json')); c=json.load(open('$O/bench_res_mf0_$i.json')); print('staged %.3f  resident %.3f  resident, column tiles fastest %.3f ms/step' % (a['ms_per_step'], b['ms_per_step'], c['ms_per_step']))"
done
for mf in 1 0; do
  (cd /tmp && PICONS_X6_MFAST=$mf PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mf$mf -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_mf$mf.log 2>&1)
done
python3 tools/compare_conv_launches.py $O/prof_mf1/p_kernel_trace.csv $O/prof_mf0/p_kernel_trace.csv 6 > $O/mfast_launches.txt 2>&1
grep -E "x$" $O/mfast_launches.txt | awk '{r=$NF; sub("x","",r); if (r<0.95 || r>1.05) print}'; tail -1 $O/mfast_launches.txt
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/pmc_f.log 2>&1)
python3 - <<PY
import csv, collections
per = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open("$O/pmc_f/f_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE" and "conv_x6" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-40:]; per[k][0] += 1; per[k][1] += float(r["Counter_Value"])
for k, (n, v) in per.items(): print("%-42s %4d launches  %.1f MB fetched per launch (2 x FETCH_SIZE)" % (k, n, 2 * v * 1024 / n / 1e6))
PY

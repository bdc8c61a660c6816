#!/bin/bash
# Round 4, call I: pipelined generic bf16-split weight gradient (one accumulator vs the hi / lo pair) + which bf16-split launches move
# Mixed_4f.b2a.bn.bias in the bs-8 JHMDB case.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_i
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python3 tools/wgrad_x6_acc_probe.py > $O/acc_probe.txt 2>&1; cat $O/acc_probe.txt | tail -8
timeout 900 python3 -m pytest tests/test_x6_gpu.py -x -q > $O/x6_tests.log 2>&1; echo "rc=$?" >> $O/x6_tests.log; tail -4 $O/x6_tests.log
run() { tag=$1; shift; env "$@" timeout 600 python3 tools/probe_tensor_grad.py 2 conv1.Mixed_4f.b2a.bn.bias > $O/probe_$tag.txt 2>&1; echo "== $tag"; grep -h "Mixed_4f.b2a.bn.bias" $O/probe_$tag.txt || tail -3 $O/probe_$tag.txt; }
run all_wg1
run fwd_only PICONS_SPLIT_LISTS=fwd PICONS_SPLIT_WGRAD=0
run bwd_only PICONS_SPLIT_LISTS=bwd PICONS_SPLIT_WGRAD=0
run wgrad_only PICONS_SPLIT_LISTS=none PICONS_SPLIT_WGRAD=1
run bwd_rows_le_12544 PICONS_SPLIT_LISTS=bwd PICONS_SPLIT_ROWS_MAX=12544 PICONS_SPLIT_WGRAD=0
run bwd_rows_gt_12544 PICONS_SPLIT_LISTS=bwd PICONS_SPLIT_ROWS_MIN=12545 PICONS_SPLIT_WGRAD=0
PICONS_LANES=1 timeout 600 python3 bench.py --steps 40 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/bench_1lane.json 2> $O/bench_1lane.err
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/bench.json 2> $O/bench.err
python3 -c "import json; j=json.load(open('$O/bench.json')); print('%.3f ms/step  %.1f clips/s  loss %.6f' % (j['ms_per_step'], j['value'], j['loss']['total']))"

#!/bin/bash
# Round 4, call E: after the double-accumulating column reduction: whole GPU suite + bench.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -6 $O/tests.log
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --leg-steps 30 > $O/bench.json 2> $O/bench.err
python3 -c "import json; j=json.load(open('$O/bench.json')); print(json.dumps({k: (j[k] if not isinstance(j[k], dict) else {q: j[k][q] for q in j[k] if q in ('ms_per_step','value','frac','achieved','kernel_ms_per_step','host_wait_ms_per_step','launches_per_step')}) for k in ('ms_per_step','value','resident','dict_contract','split_off','roofline','roofline_fp32_conv','roofline_winograd')}, indent=1))"
head -3 gpurun_out/step_grad_err_jhmdb_bv_bs8.txt

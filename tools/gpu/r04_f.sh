#!/bin/bash
# Round 4, call F: which bf16-split launches move the bs-8 JHMDB case's Mixed_4f.b2a.bn.bias over its bar; trajectory test; bench legs.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_f
mkdir -p $O
cd $R
PICONS_SPLIT_SPECTRAL=0 timeout 600 python3 tools/probe_tensor_grad.py 2 conv1.Mixed_4f.b2a.bn.bias > $O/probe_nospectral.txt 2>&1
PICONS_SPLIT_ONLY_SPECTRAL=1 timeout 600 python3 tools/probe_tensor_grad.py 2 conv1.Mixed_4f.b2a.bn.bias > $O/probe_onlyspectral.txt 2>&1
grep -h "rel-L2\|^case" $O/probe_nospectral.txt $O/probe_onlyspectral.txt
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "trajectory" > $O/traj.log 2>&1; tail -4 $O/traj.log; cat gpurun_out/trajectory_default.json | head -40
timeout 900 python3 bench.py --steps 60 --no-cpu-baseline --leg-steps 30 > $O/bench.json 2> $O/bench.err
python3 -c "import json; j=json.load(open('$O/bench.json')); print(json.dumps({k: (j[k] if not isinstance(j[k], dict) else {q: j[k][q] for q in j[k] if q in ('ms_per_step','value','frac','achieved','kernel_ms_per_step','host_wait_ms_per_step','launches_per_step','error')}) for k in ('ms_per_step','value','resident','dict_contract','split_off')}, indent=1))"

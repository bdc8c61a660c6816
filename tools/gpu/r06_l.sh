#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_l
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "deterministic or small or lanes or sample_stager" > $O/t.log 2>&1; echo "step tests rc=$?"; tail -3 $O/t.log
for rep in 1 2 3; do
  for hf in 0 1; do
    PICONS_HEAD_FIRST=$hf timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-legs --no-kernel-timing > $O/b_hf${hf}_$rep.json 2> $O/b_hf${hf}_$rep.err
    python3 -c "
import json; j=json.load(open('$O/b_hf${hf}_$rep.json')); print('head_first=$hf rep $rep: staged %.3f ms/step  resident %.3f' % (j['ms_per_step'], (j.get('resident') or {}).get('ms_per_step') or 0))"
  done
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_h
mkdir -p $O
cd $R
for rep in 1 2 3; do
  timeout 600 python3 tools/bench_prev_tmp.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/prev_$rep.json 2> $O/prev_$rep.err
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/new_$rep.json 2> $O/new_$rep.err
  python3 -c "
import json
a=json.load(open('$O/prev_$rep.json')); b=json.load(open('$O/new_$rep.json'))
print('rep $rep  --steps 20: prev %.3f ms (resident %.3f)   new %.3f ms (resident %.3f)' % (a['ms_per_step'], a['resident']['ms_per_step'], b['ms_per_step'], b['resident']['ms_per_step']))"
done
timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs > $O/new_200.json 2> $O/new_200.err
python3 -c "
import json
b=json.load(open('$O/new_200.json')); print('default 200 steps: new %.3f ms (resident %.3f)' % (b['ms_per_step'], b['resident']['ms_per_step']))"
timeout 900 python3 -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -2

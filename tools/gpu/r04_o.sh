#!/bin/bash
# Round 4, call O: the stagers' side stream at high priority (a hardware queue of its own) against normal priority (shares a lane's queue)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_o
mkdir -p $O
cd $R
for i in 1 2; do
for pr in -1 0; do
PICONS_STAGE_STREAM_PRIO=$pr timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $O/bench_staged_p${pr}_$i.json 2> $O/bench_staged_p${pr}_$i.err
python3 -c "import json; a=json.load(open('$O/bench_staged_p${pr}_$i.json')); print('prio $pr rep $i: staged %.3f ms/step' % a['ms_per_step'])"
done
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/bench_res_$i.json 2> $O/bench_res_$i.err
python3 -c "import json; a=json.load(open('$O/bench_res_$i.json')); print('resident rep $i: %.3f ms/step' % a['ms_per_step'])"
done
for pr in -1 0; do
PICONS_STAGE_STREAM_PRIO=$pr timeout 900 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing > $O/bench_legs_p$pr.json 2> $O/bench_legs_p$pr.err
python3 -c "import json; a=json.load(open('$O/bench_legs_p$pr.json')); print('prio $pr: staged %.3f resident %.3f dict %.3f (host wait %.2f)' % (a['ms_per_step'], a['resident']['ms_per_step'], a['dict_contract']['ms_per_step'], a['dict_contract']['host_wait_ms_per_step']))"
done

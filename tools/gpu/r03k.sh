#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "bn_finalize or unit3d" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or small or deterministic" >> $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
for i in 1 2; do timeout 300 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/bench$i.json 2> $O/bench$i.err; done
grep -E "passed|failed|rc=" $O/tests.log
python3 -c "
import json
for i in (1,2):
    j=json.load(open('gpurun_out/r03k/bench%d.json'%i)); print(j['ms_per_step'])
"

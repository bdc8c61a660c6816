#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03l
mkdir -p $O
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_dropin_gpu.py -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
for i in 1 2; do timeout 300 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/bench$i.json 2> $O/bench$i.err; done
grep -E "passed|failed|rc=" $O/tests.log
python3 -c "
import json
for i in (1,2):
    j=json.load(open('gpurun_out/r03l/bench%d.json'%i)); print(j['ms_per_step'])
"

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
timeout 900 python3 -m pytest tests/test_wino_gpu.py -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 600 python3 tools/bench_wino.py 10 > $O/bench_wino.txt 2>&1
tail -5 $O/tests.log; grep -v amdgpu.ids $O/bench_wino.txt

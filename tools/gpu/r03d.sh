#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout 600 python3 -m pytest tests/test_wino_gpu.py -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 600 python3 tools/bench_wino.py 10 > $O/bench_wino.txt 2>&1
timeout 900 python3 tools/probe_wino.py > $O/probe_wino.txt 2>&1
tail -8 $O/tests.log; grep -v amdgpu.ids $O/bench_wino.txt; grep -v amdgpu.ids $O/probe_wino.txt

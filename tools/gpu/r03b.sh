#!/bin/bash
# Round 3, GPU call B: Winograd conv kernel -- parity tests, then the go / no-go bench against the gather-GEMM kernel.
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout 600 python3 -m pytest tests/test_wino_gpu.py -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 600 python3 tools/bench_wino.py 10 > $O/bench_wino.txt 2>&1
tail -15 $O/tests.log; cat $O/bench_wino.txt

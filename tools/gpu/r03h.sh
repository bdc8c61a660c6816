#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout 900 python3 -m pytest tests/test_wino_gpu.py -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -25 $O/tests.log

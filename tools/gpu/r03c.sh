#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 900 python3 tools/probe_wino.py > $O/probe_wino.txt 2>&1
cat $O/probe_wino.txt | grep -v amdgpu.ids

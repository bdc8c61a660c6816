#!/bin/bash
# step-level parity (reference fixtures at 8x224x224, oracle cases at 112^2, reference-init case) under non-default switch combinations
export TMPDIR=/tmp
mkdir -p gpurun_out/switches
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -m gpu -k "golden or small or reference_init or early_adam or trajectory" > gpurun_out/switches/$i.log 2>&1
  echo "$cfg :: $(tail -1 gpurun_out/switches/$i.log)"
done

#!/bin/bash
# usage: ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> one 100-step bench per setting, ms per step
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 300 python3 bench.py --steps ${AB_STEPS:-100} --no-cpu-baseline --no-kernel-timing --resident-inputs > gpurun_out/ab/$i.json 2> gpurun_out/ab/$i.err
  python3 -c "
import json,sys
try:
    j=json.load(open('gpurun_out/ab/$i.json')); print('%-50s %.3f ms' % ('$cfg', j['ms_per_step']))
except Exception as e: print('$cfg', 'FAILED', e)
"
done

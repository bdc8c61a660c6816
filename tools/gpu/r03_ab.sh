#!/bin/bash
# lane / switch A-B on the final code (each: 100 steps, no kernel-timing leg, inputs resident)
export TMPDIR=/tmp
O=gpurun_out/r03ab
mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/$name.json 2> $O/$name.err; python3 -c "
import json,sys
try:
    j=json.load(open('$O/$name.json')); print('%-34s %.3f ms' % ('$name', j['ms_per_step']))
except Exception as e: print('$name failed')
"; }
run base PICONS_LANES=4
run base2 PICONS_LANES=4
run lanes3 PICONS_LANES=3
run lanes5 PICONS_LANES=5
run skip_lane_off PICONS_SKIP_LANE=0
run branch3_off PICONS_FWD_BRANCH3=0
run wgrad_multi PICONS_WGRAD_MULTI=1
run late_prep_off PICONS_LATE_PREP=0
run prio PICONS_PRIO=1
run wino_off PICONS_WINO=0

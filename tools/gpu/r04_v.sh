#!/bin/bash
# Round 4, call V: the lane-assignment switches re-measured with this round's kernel durations (resident step, one box, two passes)
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_v
mkdir -p $O
cd $R
i=0
for rep in 1 2; do
for cfg in "PICONS_LANES=4" "PICONS_WGRAD_SEPARATE=0" "PICONS_SPREAD_CLASSES=0" "PICONS_CONV28_ASIDE=0" "PICONS_FWD_BRANCH3=0" "PICONS_LATE_PREP=0" "PICONS_WGRAD_MULTI=1" "PICONS_DEFER_SIDE=1" "PICONS_EARLY_ADAM=0" "PICONS_WGRAD_ROUNDS=1.5" "PICONS_CONV_STAGES=2" "PICONS_LANES=3"; do
  i=$((i+1))
  env $cfg timeout 300 python3 bench.py --steps 100 --no-cpu-baseline --no-kernel-timing --no-extra-legs --resident-inputs > $O/b_$i.json 2> $O/b_$i.err
  python3 -c "import json; a=json.load(open('$O/b_$i.json')); print('%-28s rep $rep: %.3f ms/step' % ('$cfg', a['ms_per_step']))" 2>/dev/null || echo "$cfg failed"
done; done

#!/bin/bash
# Round 4, call C: (1) where the bs-8 JHMDB case's one gradient over the bar comes from; (2) single-lane kernel stats with and without the
# bf16-split conv kernel.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
for sp in 1 0; do
  (cd /tmp && PICONS_SPLIT=$sp PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_split$sp -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --resident-inputs > $O/prof_split$sp.log 2>&1)
done
for sp in 1 0; do
  PICONS_SPLIT=$sp timeout 900 python3 tools/probe_tensor_grad.py 2 conv1.Mixed_4f.b2a.bn.bias conv1.Mixed_4f.b2a.bn.weight > $O/probe_split$sp.txt 2>&1
done
cat $O/probe_split1.txt $O/probe_split0.txt | grep -v Warn | tail -30
find $O -name "*kernel_stats.csv" | head

#!/bin/bash
# Round 4, call G: which bf16-split launches move Mixed_4f.b2a.bn.bias (bs-8 JHMDB case).
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_g
mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 600 python3 tools/probe_tensor_grad.py 2 conv1.Mixed_4f.b2a.bn.bias > $O/probe_$tag.txt 2>&1; echo "== $tag"; grep -h "rel-L2" $O/probe_$tag.txt || tail -3 $O/probe_$tag.txt; }
run fwd_only PICONS_SPLIT_LISTS=fwd PICONS_SPLIT_WGRAD=0
run bwd_only PICONS_SPLIT_LISTS=bwd PICONS_SPLIT_WGRAD=0
run bwd_rows_le_12544 PICONS_SPLIT_LISTS=bwd PICONS_SPLIT_ROWS_MAX=12544 PICONS_SPLIT_WGRAD=0
run bwd_rows_gt_12544 PICONS_SPLIT_LISTS=bwd PICONS_SPLIT_ROWS_MIN=12545 PICONS_SPLIT_WGRAD=0

#!/bin/bash
# Round 4, call J: is Mixed_4f.b2a.bn.bias' distance from fp64 (bs-8 JHMDB case) a draw?  The same step under settings that are all
# fp32-accurate but round differently, with and without the bf16 split.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_j
mkdir -p $O
cd $R
timeout 2400 python3 tools/probe_rerolls.py 2 conv1.Mixed_4f.b2a.bn.bias,conv1.Mixed_4f.b2a.bn.weight \
  "split:" "fp32:PICONS_SPLIT=0" \
  "fp32 lanes1:PICONS_SPLIT=0,PICONS_LANES=1" "fp32 no-winograd:PICONS_SPLIT=0,PICONS_WINO=0" "fp32 winograd T>1 only:PICONS_SPLIT=0,PICONS_WINO_T1=0" \
  "fp32 unstacked 1x1:PICONS_SPLIT=0,PICONS_FUSE1X1=0" "fp32 direct 9x9:PICONS_SPLIT=0,PICONS_SPECTRAL=0" "fp32 27-ch tail:PICONS_SPLIT=0,PICONS_TAIL6=0" \
  "fp32 generic wgrad:PICONS_SPLIT=0,PICONS_WGRAD_ROW=0" \
  "split lanes1:PICONS_LANES=1" "split no-winograd:PICONS_WINO=0" "split winograd T>1 only:PICONS_WINO_T1=0" "split unstacked 1x1:PICONS_FUSE1X1=0" \
  "split direct 9x9:PICONS_SPECTRAL=0" "split 27-ch tail:PICONS_TAIL6=0" "split fwd only:PICONS_SPLIT_LISTS=fwd,PICONS_SPLIT_WGRAD=0" \
  "split bwd only:PICONS_SPLIT_LISTS=bwd" > $O/rerolls.txt 2>&1
cat $O/rerolls.txt

#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
# Round 3, GPU call A: diagnostics (matrix-pipe ceiling in situ, conv shapes with counters) + the full GPU suite + the default bench.
# Run from the repo root on the GPU box:  bash tools/gpu/r03a.sh
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_probe.hip -o $O/mfma_peak_probe 2>/dev/null && timeout 300 $O/mfma_peak_probe > $O/mfma_peak_probe.txt 2>&1
ABLATE_REPS=10 timeout 600 python3 tools/ablate_conv.py > $O/ablate.txt 2>&1
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ablate -o a -- python3 $GRAFT_REPO_ROOT/tools/ablate_conv.py > $GRAFT_REPO_ROOT/$O/pmc_ablate.log 2>&1)
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
tail -3 $O/tests.log; cat $O/mfma_peak_probe.txt; cat $O/ablate.txt; head -c 1500 $O/bench.json

#!/bin/bash
O=gpurun_out/r03g
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_probe.hip -o $O/mfma_peak_probe 2>/dev/null && PROBE_FILL=1 timeout 300 $O/mfma_peak_probe > $O/fill.txt 2>&1
cat $O/fill.txt

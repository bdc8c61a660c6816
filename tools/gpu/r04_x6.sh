#!/bin/bash
# Round 4: fp32-on-bf16-matrix-cores tile probe (tools/x6_tile_probe.hip) on GEMM shapes of the step.
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04_x6
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DSPLIT_TRUNC=1 -DSTAMPS=1 $R/tools/x6_tile_probe.hip -o $O/x6tile_trunc 2>/dev/null || exit 1
{
for shp in "131072 256 1152" "12544 256 832"; do
  timeout 300 $O/x6tile_trunc $shp
done
} > $O/x6tile.txt 2>&1
cat $O/x6tile.txt

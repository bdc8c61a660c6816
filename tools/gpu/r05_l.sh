#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r05_l
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DSTAMPS=1 tools/x6_tile_probe.hip -o $O/x6tile 2> $O/build.log || { cat $O/build.log; exit 1; }
{ for shp in "131072 256 1152" "13120 512 7488" "12544 256 832"; do timeout 300 $O/x6tile $shp; done; } > $O/x6tile.txt 2>&1
grep -E "^M=|128x64 pipe|early B|wait stamps|256x128 pipe, B planes, 8x1" $O/x6tile.txt | cut -c1-330

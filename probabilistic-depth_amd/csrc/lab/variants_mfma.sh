#!/bin/bash
# Build variants of the matrix-pipe sweep kernel (experiment / diagnostic knobs) as separate libraries under
# gpurun_variants/, selected through PDEPTH_LIB; the product library is never overwritten.
#   usage: tools/variants_mfma.sh "name1:-DMFMA_STAMPS" "name2:-DMFMA_MAXB=12" ...     (build container)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=probabilistic-depth_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-inline-asm"
mkdir -p gpurun_variants
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c $C/sweep_mfma.hip -o /tmp/sweep_mfma_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/libm_$name.so $C/capi.o $C/sweep_direct.o $C/sweep_tiled.o \
      $C/sweep_tiled_n2.o /tmp/sweep_mfma_$name.o $C/sweep_cells.o $C/sweep_cells_fast.o $C/dpv.o $C/warp.o $C/extras.o $C/correlation_general.o $C/ufield.o
  echo built gpurun_variants/libm_$name.so "($defs)"
done

#!/bin/bash
# Build variants of the cell-list kernel with other knobs and time bench.py with each (experiments only).
#   usage: tools/variants_cells.sh build "name1:-DCELLS_OCC=2" "name2:-DCELLS_NBUF=2" ...   (build container)
#          tools/variants_cells.sh run [bench args]                                          (GPU box)
# Variants are separate libraries under gpurun_variants/ selected through PDEPTH_LIB; the product library is never touched.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=probabilistic-depth_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-inline-asm"
if [ "$1" = build ]; then
  shift; mkdir -p gpurun_variants
  for spec in "$@"; do
    name=${spec%%:*}; defs=${spec#*:}
    /opt/rocm/bin/hipcc $FLAGS $defs -c $C/sweep_cells_fast.hip -o /tmp/sweep_cells_fast_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/lib_$name.so $C/capi.o $C/sweep_direct.o $C/sweep_tiled.o \
        $C/sweep_tiled_n2.o $C/sweep_cells.o /tmp/sweep_cells_fast_$name.o $C/dpv.o $C/warp.o $C/extras.o
    echo built gpurun_variants/lib_$name.so "($defs)"
  done
else
  shift || true
  for round in 1 2; do
    for f in probabilistic-depth_amd/libpdepth_hip.so gpurun_variants/lib_*.so; do
      PDEPTH_LIB=$PWD/$f python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'fallback', d['gather_fallback_tiles'])"
    done
  done
fi

#!/bin/bash
# A/B builds of the distance-form sweep kernel: tools/variants_dist.sh name "flags" [name "flags" ...]
# -> gpurun_variants/libpdepth_<name>.so (select with PDEPTH_LIB=...); the other objects come from the product build.
# SRC=pack_dist tools/variants_dist.sh ... : variants of another source of the library
set -e
cd "$(dirname "$0")/../probabilistic-depth_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../gpurun_variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -Wall -Wno-unused-function -Wno-inline-asm"
SRC=${SRC:-sweep_dist}
OTHERS=$(echo "capi.o sweep_direct.o sweep_pack.o pack_dist.o sweep_dist.o sweep_tiled.o dpv.o warp.o extras.o correlation_general.o ufield.o sweep_tiled_n2.o" | sed "s/\b$SRC\.o//")
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc $FLAGS $flags -c $SRC.hip -o /tmp/${SRC}_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/libpdepth_$name.so /tmp/${SRC}_$name.o $OTHERS
    echo "built gpurun_variants/libpdepth_$name.so ($flags)"
done

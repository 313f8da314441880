#!/bin/bash
# time bench.py with each probabilistic-depth_amd/libvariant_*.so swapped in (experiments only)
# (link a variant from csrc/: capi.o sweep_direct.o <sweep_tiled variant>.o sweep_tiled_n2.o dpv.o warp.o extras.o)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp probabilistic-depth_amd/libpdepth_hip.so /tmp/full.so
for f in probabilistic-depth_amd/libvariant_*.so; do
  cp $f probabilistic-depth_amd/libpdepth_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['roofline']['kernel_ms'], d['gather_fallback_tiles'])"
done
cp /tmp/full.so probabilistic-depth_amd/libpdepth_hip.so

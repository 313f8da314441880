#!/bin/bash
# usage: tools/dbg/pmc_variants.sh "<counters>" [bench args] -- the PMC pass of pmc_pass.sh for the product library and every gpurun_variants/lib_*.so
ctrs=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for f in probabilistic-depth_amd/libpdepth_hip.so gpurun_variants/lib_*.so; do
  n=$(basename $f .so); echo "== $n"
  PDEPTH_LIB=$PWD/$f tools/dbg/pmc_pass.sh v_$n "$ctrs" "$@" | grep sweep_tiled
  PDEPTH_LIB=$PWD/$f python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   kernel_ms', round(d['roofline']['kernel_ms'],4))"
done

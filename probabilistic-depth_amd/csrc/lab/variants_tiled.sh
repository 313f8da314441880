#!/bin/bash
# Build variants of the two-tile tiled kernel (counter experiments, e.g. the PDEPTH_CONF_* hooks) and collect the LDS
# counters of bench.py with each.  Variants are separate libraries under gpurun_variants/ selected through PDEPTH_LIB.
#   usage: tools/variants_tiled.sh build "name1:-DPDEPTH_CONF_TAPS" ...     (build container)
#          tools/variants_tiled.sh pmc                                       (GPU box)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=probabilistic-depth_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-inline-asm"
if [ "$1" = build ]; then
  shift; mkdir -p gpurun_variants; rm -f gpurun_variants/lib_*.so
  for spec in "$@"; do
    name=${spec%%:*}; defs=${spec#*:}
    /opt/rocm/bin/hipcc $FLAGS -DPDEPTH_NSUB=2 $defs -c $C/sweep_tiled.hip -o /tmp/sweep_tiled_n2_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/lib_$name.so $C/capi.o $C/sweep_direct.o $C/sweep_tiled.o $C/sweep_mfma.o \
        /tmp/sweep_tiled_n2_$name.o $C/sweep_cells.o $C/sweep_cells_fast.o $C/dpv.o $C/warp.o $C/extras.o $C/correlation_general.o $C/ufield.o
    echo built gpurun_variants/lib_$name.so "($defs)"
  done
else
  export TMPDIR=/tmp
  for f in probabilistic-depth_amd/libpdepth_hip.so gpurun_variants/lib_*.so; do
    n=$(basename $f .so); out=gpurun_out/conf_$n; rm -rf $out; mkdir -p $out
    export PDEPTH_LIB=$PWD/$f
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $out -- \
      python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1 || true
    python3 - $out $n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep_tiled_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
    python3 -c "import json,sys; d=[l for l in open('$out/log.txt') if l.startswith('{')]; print('   kernel_ms', json.loads(d[-1])['roofline']['kernel_ms'] if d else 'n/a')"
  done
fi

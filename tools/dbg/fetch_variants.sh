#!/bin/bash
# usage: tools/dbg/fetch_variants.sh name [name ...]: time (packed entry, headline shapes) and FETCH_SIZE per call of the
# distance-form kernel per gpurun_variants/libpdepth_<name>.so ("base" = the product library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for v in "$@"; do
  lib=gpurun_variants/libpdepth_$v.so; [ "$v" = base ] && lib=""
  export PDEPTH_LIB=$lib
  t=$(timeout -k 10 200 python tools/dbg/dist_time.py 2>&1 | tail -1)
  out=gpurun_out/fetch_$v; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 tools/dbg/one_sweep.py 4 256 512 packed 20 > $out/log.txt 2>&1 || true
  f=$(python3 - $out <<'PY'
import csv, glob, os, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[-1])) if "sweep_dist" in r["Kernel_Name"]] if fs else []
print("FETCH_SIZE x2 = %.0f MB per call (n=%d)" % (2 * 1.024e-3 * sum(v) / max(len(v), 1), len(v)))
PY
)
  echo "$v: $t   $f"
done

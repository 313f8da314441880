#!/bin/bash
# usage: tools/prof_rows.sh <tag>  -- rocprofv3 kernel-trace summaries of the rows that had none (GPU box): one file per case under
# gpurun_out/rows_<tag>/, the per-kernel table of each in gpurun_out/rows_<tag>.txt (copied to profiles/ by hand).
set -e
tag=$1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/rows_$tag
mkdir -p $out
export TMPDIR=/tmp
for case in cfg5 cfg5_packed cfg3 cfg2_tiled model_real model_real_packed reduce_ex dpv_fuse ufield correlation correlation_general pack_views pack_views_small; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$case -- python3 tools/prof_rows.py $case 20 > $out/$case.log 2>&1 || true
  echo "== $case (python3 tools/prof_rows.py $case 20; 23 calls incl. warm-up)"
  python3 - $out/$case <<'PY'
import csv, glob, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**/*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 20 and "pdepth" in r["Name"]:
            print("   %-86s calls %4d  avg %10.2f us  min %10.2f  max %10.2f" % (r["Name"][:86], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done

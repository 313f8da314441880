#!/bin/bash
# usage: tools/prof_small.sh <tag>   -- GPU time of the small (model-real) sweeps from a rocprofv3 kernel trace: per call
# the kernels launched and their average durations, next to the wall time of back-to-back calls (host + launch overhead).
set -e
tag=$1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/small_$tag
mkdir -p $out
export TMPDIR=/tmp
for case in "1 64 128 auto" "1 64 96 auto" "1 64 128 packed" "1 64 128 tiled1" "4 64 128 auto" "4 64 128 packed"; do
  set -- $case
  name=B$1_$2x$3_$4
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 tools/dbg/one_sweep.py $1 $2 $3 $4 200 > $out/$name.log 2>&1 || true
  python3 - $out/$name $name <<'PY'
import csv, glob, os, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**/*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]
wall = [l for l in open(sys.argv[1] + ".log") if l.startswith("wall")]
tot, rows = 0.0, []
for f in fs:
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) < 150: continue   # setup kernels (copies, fills) are not part of a step
        per_call = float(r["TotalDurationNs"]) / 200.0 / 1e3
        tot += per_call
        rows.append("      %-70s calls/step %.2f  avg %.2f us" % (r["Name"][:70], int(r["Calls"]) / 210.0, float(r["AverageNs"]) / 1e3))
print("%s: GPU time per call %.1f us in %d kernels; %s" % (sys.argv[2], tot, len(rows), wall[-1].strip() if wall else ""))
print("\n".join(rows))
PY
done

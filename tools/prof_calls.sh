#!/bin/bash
# usage: tools/prof_calls.sh <tag> "B H W algo" ...   -- rocprofv3 kernel trace of 200 back-to-back fused sweep calls per case:
# the kernels of a call with their average durations, next to the wall time per call.
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/calls_$tag
mkdir -p $out
export TMPDIR=/tmp
for case in "$@"; do
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
        per_call = float(r["TotalDurationNs"]) / 210.0 / 1e3
        tot += per_call
        rows.append("      %-70s calls/step %.2f  avg %.2f us" % (r["Name"][:70], int(r["Calls"]) / 210.0, float(r["AverageNs"]) / 1e3))
print("%s: GPU time per call %.1f us in %d kernels; %s" % (sys.argv[2], tot, len(rows), wall[-1].strip() if wall else ""))
print("\n".join(rows))
PY
done

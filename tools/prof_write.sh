#!/bin/bash
# usage: tools/prof_write.sh <tag> <lib or ""> -- WRITE_SIZE / FETCH_SIZE of 30 packed sweeps (headline shape) with a variant library
tag=$1; lib=$2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/wr_$tag
mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export PDEPTH_LIB=$lib
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/p1 -- python3 tools/dbg/one_sweep.py 4 256 512 packed 30 > $out/p1.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p2 -- python3 tools/dbg/one_sweep.py 4 256 512 packed 30 > $out/p2.log 2>&1 || true
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(os.path.join(root, sub + "/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs: print("no counters", sub); continue
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(fs[-1])):
        if "sweep_dist" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, v in acc.items(): print("   %-12s %.5g KB per launch (n=%d)" % (c, sum(v) / len(v), len(v)))
PY

#!/bin/bash
# usage: tools/dbg/pmc_pass.sh <tag> "<counters>" [bench args]  -- one rocprofv3 --pmc pass of bench.py, prints per-kernel means
tag=$1; ctrs=$2; shift 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/pmcx_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/log.txt 2>&1 || tail -5 $out/log.txt
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pdepth" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(cs.items())})
PY

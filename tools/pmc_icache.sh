#!/bin/bash
# usage: tools/pmc_icache.sh <tag> [bench args...] -- instruction-cache counters of one sweep implementation (GPU box)
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_DCACHE_REQ SQC_DCACHE_MISSES GRBM_GUI_ACTIVE --output-format csv -d $out/p5 -- $B "$@" > $out/p5.log 2>&1 || true
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("p5",):
    fs = sorted(glob.glob(os.path.join(root, sub, "**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    for f in fs:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "sweep" not in k: continue
            print("kernel", k, sub)
            for c, v in sorted(cs.items()): print("     %-28s %.4g" % (c, sum(v) / len(v)))
PY

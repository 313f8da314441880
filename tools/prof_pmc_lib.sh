#!/bin/bash
# usage: tools/prof_pmc_lib.sh <tag> <lib> : instruction-mix PMC pass of 30 back-to-back dist sweeps with a variant library
tag=$1; lib=$2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmcl_$tag
mkdir -p $out
export TMPDIR=/tmp
export PDEPTH_LIB=$lib
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $out/p1 -- python3 tools/dbg/one_sweep.py 4 256 512 dist 30 > $out/p1.log 2>&1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $out/p2 -- python3 tools/dbg/one_sweep.py 4 256 512 dist 30 > $out/p2.log 2>&1 || true
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(os.path.join(root, sub + "/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs: print("no counters", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[-1])):
        acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "sweep_dist" not in k: continue
        for c, v in sorted(cs.items()): print("   %-26s %.5g" % (c, sum(v) / len(v)))
PY

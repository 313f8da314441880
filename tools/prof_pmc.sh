#!/bin/bash
# usage: tools/prof_pmc.sh <tag> "B H W algo"   -- kernel trace + PMC passes (each in its own run) of 30 back-to-back fused sweep calls
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
set -- $1
run() { d=$1; shift; rocprofv3 "$@" --output-format csv -d $out/$d -- python3 tools/dbg/one_sweep.py $B $H $W $A 30 > $out/$d.log 2>&1 || true; }
B=$1; H=$2; W=$3; A=$4
run trace --kernel-trace --stats
run pmc1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run pmc2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
run pmc5 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
def find(pattern):
    fs = sorted(glob.glob(os.path.join(root, pattern), recursive=True), key=os.path.getmtime)
    return fs[-1:]
for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats")
    for i, row in enumerate(csv.reader(open(f))):
        if i < 6: print("  ", ", ".join(row[:8]))
for sub in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5"):
    for f in find(sub + "/**/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters:", sub)
        for k, cs in acc.items():
            if "sweep" not in k and "pack" not in k: continue
            print("  kernel", k)
            for c, v in sorted(cs.items()):
                print("     %-28s mean/dispatch %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY

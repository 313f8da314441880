#!/bin/bash
# usage: tools/pmc_mfma.sh <tag> [bench args...]  -- PMC passes of bench.py for one sweep implementation (GPU box)
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d $out/p1 -- $B "$@" > $out/p1.log 2>&1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH --output-format csv -d $out/p2 -- $B "$@" > $out/p2.log 2>&1 || true
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC --output-format csv -d $out/p3 -- $B "$@" > $out/p3.log 2>&1 || true
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/p4 -- $B "$@" > $out/p4.log 2>&1 || true
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("p1", "p2", "p3", "p4"):
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

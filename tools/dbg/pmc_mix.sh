#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmc_mix
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --algo mfma"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH --output-format csv -d $out/p1 -- $B "$@" > $out/p1.log 2>&1 || true
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/p2 -- $B "$@" > $out/p2.log 2>&1 || true
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(os.path.join(root, sub, "**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    for f in fs:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "sweep_mfma" not in k: continue
            for c, v in sorted(cs.items()): print("     %-28s %.4g" % (c, sum(v) / len(v)))
PY

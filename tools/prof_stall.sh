#!/bin/bash
# usage: tools/prof_stall.sh <tag> <lib> : instruction-fetch / LDS / TA stall counters of 30 back-to-back dist sweeps
tag=$1; lib=$2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/stall_$tag
mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export PDEPTH_LIB=$lib
run() { d=$1; shift; rocprofv3 "$@" --output-format csv -d $out/$d -- python3 tools/dbg/one_sweep.py 4 256 512 dist 30 > $out/$d.log 2>&1 || true; }
run q1 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run q2 --pmc SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
run q3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run q4 --pmc TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("q1", "q2", "q3", "q4"):
    fs = sorted(glob.glob(os.path.join(root, sub + "/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs: print("no counters", sub); os.system("tail -2 %s/%s.log" % (root, sub)); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[-1])):
        acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "sweep_dist" not in k: continue
        for c, v in sorted(cs.items()): print("   %-40s %.5g" % (c, sum(v) / len(v)))
PY

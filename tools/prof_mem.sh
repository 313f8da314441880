#!/bin/bash
# usage: tools/prof_mem.sh <tag> "B H W algo"   -- memory-side PMC passes of 30 back-to-back fused sweep calls
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/mem_$tag
mkdir -p $out
export TMPDIR=/tmp
set -- $1
B=$1; H=$2; W=$3; A=$4
run() { d=$1; shift; rocprofv3 "$@" --output-format csv -d $out/$d -- python3 tools/dbg/one_sweep.py $B $H $W $A 30 > $out/$d.log 2>&1 || true; }
run m1 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run m2 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run m3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run m4 --pmc TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("m1", "m2", "m3", "m4"):
    fs = sorted(glob.glob(os.path.join(root, sub + "/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        print("== no counters:", sub); os.system("tail -3 %s/%s.log" % (root, sub)); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[-1])):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters:", sub)
    for k, cs in acc.items():
        if "sweep" not in k: continue
        print("  kernel", k)
        for c, v in sorted(cs.items()):
            print("     %-40s mean/dispatch %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY

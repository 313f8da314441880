#!/bin/bash
# usage: tools/prof_l2.sh <tag> <lib or ""> ["B H W algo"] -- L2 hit / miss and fetched bytes of 30 back-to-back packed sweeps with a variant library
tag=$1; lib=$2; shape=${3:-"4 256 512 dist"}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/l2_$tag
mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export PDEPTH_LIB=$lib
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/p1 -- python3 tools/dbg/one_sweep.py $shape 30 > $out/p1.log 2>&1 || true
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
fs = sorted(glob.glob(os.path.join(root, "p1/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
if not fs: print("no counters"); os.system("tail -5 %s/p1.log" % root); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(fs[-1])):
    acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "sweep_dist" not in k: continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()): print("   %-26s %.5g" % (c, v))
    if "TCC_HIT_sum" in m: print("   L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
    if "TCC_EA0_RDREQ_sum" in m: print("   read bytes beyond L2: %.1f MB" % ((m["TCC_EA0_RDREQ_32B_sum"] * 32 + (m["TCC_EA0_RDREQ_sum"] - m["TCC_EA0_RDREQ_32B_sum"]) * 64) / 1e6))
PY

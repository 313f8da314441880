"""Summarise rocprofv3 CSV output of tools/prof.sh: per-kernel time stats + PMC counters per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def find(pattern):
    # gpurun merges every call's output into the same directory: keep the newest run only
    fs = sorted(glob.glob(os.path.join(root, pattern), recursive=True), key=os.path.getmtime)
    return fs[-1:]
for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8: print("  ", ", ".join(row[:8]))
for sub in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5", "pmc6"):
    for f in find(sub + "/**/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters:", sub)
        for k, cs in acc.items():
            if not any(t in k for t in ("sweep", "dpv", "pack_c4", "pack_dist", "stats")): continue
            print("  kernel", k)
            for c, v in sorted(cs.items()):
                print("     %-24s mean/dispatch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))

export TMPDIR=/tmp
for v in nosplitpack base; do
  lib=gpurun_variants/libpdepth_$v.so; [ "$v" = base ] && lib=""
  export PDEPTH_LIB=$lib
  for c in pack_views_small model_real; do
    out=gpurun_out/pv_${v}_$c; rm -rf $out
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/prof_rows.py $c 20 > /dev/null 2>&1
    echo "== $v $c"
    python3 - $out <<'PY'
import csv, glob, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**/*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 20 and "pdepth" in r["Name"]:
            print("   %-60s avg %8.2f us" % (r["Name"][:60], float(r["AverageNs"]) / 1e3))
PY
  done
done

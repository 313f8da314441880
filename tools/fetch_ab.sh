#!/bin/bash
# A/B of the sweep kernel variants (probabilistic-depth_amd/libvariant_*.so): call time and L2-miss read bytes (experiments only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
cp probabilistic-depth_amd/libpdepth_hip.so /tmp/full.so
for f in /tmp/full.so probabilistic-depth_amd/libvariant_*.so; do
  cp $f probabilistic-depth_amd/libpdepth_hip.so 2>/dev/null
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['roofline']['kernel_ms'])"
  rm -rf gpurun_out/fx; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fx -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('gpurun_out/fx/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'sweep_tiled' in r['Kernel_Name']]
print('   FETCH_SIZE KB/dispatch', sum(v)/len(v), '-> reads MB', 2*sum(v)/len(v)*1024/1e6)
PY
done
cp /tmp/full.so probabilistic-depth_amd/libpdepth_hip.so

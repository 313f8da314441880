for v in COMPUTE STAGING; do
  cp probabilistic-depth_amd/libpdepth_hip.so /tmp/full.so
  cp probabilistic-depth_amd/libpdepth_ablate_$v.so probabilistic-depth_amd/libpdepth_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $v', d['roofline']['kernel_ms'])"
  cp /tmp/full.so probabilistic-depth_amd/libpdepth_hip.so
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', d['roofline']['kernel_ms'])"

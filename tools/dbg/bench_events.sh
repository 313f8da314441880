for t in "" 1; do
  PDEPTH_BENCH_TRACE=$t timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PDEPTH_BENCH_TRACE=%s' % '$t', 'ms_per_step %.4f kernel_ms %.4f packed_entry %.4f cold %.4f peaked %.4f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['packed_entry']['kernel_ms'], l['cold_start']['ms_per_step'], l['peaked']['ms_per_step']))"
done

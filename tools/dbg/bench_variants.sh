#!/bin/bash
# usage: tools/dbg/bench_variants.sh name [name ...] -- bench.py (headline + packed entry) per gpurun_variants/libpdepth_<name>.so ("base" = the product library)
for v in "$@"; do
  lib=gpurun_variants/libpdepth_$v.so; [ "$v" = base ] && lib=""
  PDEPTH_LIB=$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-secondary 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s ms_per_step %.4f packed_entry %.4f' % ('$v', l['ms_per_step'], l['packed_entry']['kernel_ms']))" || exit 1
done

#!/bin/bash
# A/B of the current library against every probabilistic-depth_amd/libvariant_*.so, both poses, two rounds (experiments only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pose in mono stereo; do echo $pose; for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pose $pose 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cur', d['roofline']['kernel_ms'], d['gather_fallback_tiles'])"; bash tools/variants.sh --pose $pose; done; done

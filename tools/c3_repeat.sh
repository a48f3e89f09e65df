#!/bin/bash
# tools/c3_repeat.sh [n] — bench_more.c3 n times (processes of one box)
cd "$(dirname "$0")/.."
for r in $(seq 1 ${1:-4}); do
  LSDR_BENCH_MORE_ONLY=${C3_ONLY:-c3} LSDR_BENCH_FULL=gpurun_out/c3ab.json timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-verify > /dev/null 2> gpurun_out/c3ab.err
  python -c "
import json
m=json.load(open('gpurun_out/c3ab.json'))['more']
print('run $r', m[list(m)[0]].get('value'), m[list(m)[0]].get('pass'), m[list(m)[0]]['roofline'].get('avg_launch_ms', m[list(m)[0]]['roofline'].get('tile_kernel_avg_launch_ms')), m[list(m)[0]].get('error','')[:200])"
done

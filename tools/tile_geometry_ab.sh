#!/bin/bash
# tools/tile_geometry_ab.sh — the C2 headline at several receiver tile geometries (tile_len / warm-up, samples of the decimated stream),
# three rounds interleaved (placement and box state move a single run by ±2 %): GPU box
cd "$(dirname "$0")/.."
for r in 1 2; do
for g in "256 256" "256 128" "384 128" "512 128" "512 256" "384 256"; do
  set -- $g
  timeout 300 python bench.py --no-more --no-cpu --tile-len $1 --tile-warmup $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=j['verified']
print('round $r tile $1 warm-up $2:', j['value'], 'frac', j['roofline']['frac'], 'launch ms', j['roofline']['avg_launch_ms'], 'unplaced', j.get('unplaced',{}).get('value'), 'pass', v['pass'], 'eq', v.get('equal_decisions'), 'mean/p99/max dcost', v.get('mean_abs_dcost'), v.get('p99_abs_dcost'), v.get('max_abs_dcost'), 'bad seams', v.get('bad_seams'))"
done
done

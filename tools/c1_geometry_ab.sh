#!/bin/bash
# tools/c1_geometry_ab.sh — bench.py --workload c1 at several tile geometries (every run checks all 32 TS against the reference binary): GPU box
cd "$(dirname "$0")/.."
for g in "4096 512" "4096 256" "4096 384" "8192 256" "6144 256" "4096 256" "4096 512"; do
  set -- $g
  timeout 600 python bench.py --workload c1 --no-cpu --no-single --c1-tile $1 --c1-warmup $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tile $1 warm-up $2:', j['value'], 'ms/step', j['ms_per_step'], 'tile kernel ms', j['roofline']['avg_launch_ms'], 'valu', j['roofline']['valu_issue']['frac'], 'verified', j['verified'])"
  python -c "
import json
d=json.load(open('bench_full.json'))
v=d['verified']
print('   whole TS identical:', sum(1 for r in v['per_capture'] if r.get('whole_ts_identical')), 'of', len(v['per_capture']), 'seams', d['config'].get('receiver_seams'))"
done

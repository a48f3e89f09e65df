#!/bin/bash
# tools/c1_sweep.sh — bench.py --workload c1 over capture counts / groups / tile lengths (GPU box)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"
for cfg in "$@"; do
  set -- $(echo $cfg | tr ',' ' ')
  r=$(python bench.py --workload c1 --steps 8 --warmup 2 --no-cpu --no-single --no-verify --c1-captures $1 --c1-groups $2 --c1-tile $3 --c1-warmup ${4:-512} --c1-aux-cus ${5:-0} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])")
  echo "captures $1 groups $2 tile $3 warm ${4:-512} aux ${5:-0}: $r"
done

#!/bin/bash
# tools/rx_ablation_r06.sh — what the receiver's memory side costs the filter launch in the C2 pipeline (measure build: LSDR_RX_DBG 1 = no symbol
# stores, 2 = no window loads, 3 = neither; results are garbage, nothing is verified): GPU box
cd "$(dirname "$0")/.."
M=$PWD/tools/variants/liblsdr_hip_measure.so
for r in 1 2; do
for d in 0 1 2 3; do
  LSDR_HIP_LIB=$M LSDR_RX_DBG=$d timeout 300 python bench.py --no-more --no-cpu --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r LSDR_RX_DBG=$d:', j['value'], 'frac', j['roofline']['frac'], 'launch ms', j['roofline']['avg_launch_ms'], 'unplaced', j.get('unplaced',{}).get('value'))"
done
done

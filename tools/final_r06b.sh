#!/bin/bash
# tools/final_r06b.sh — the default bench line once more at the last commit (the carried-ring tiles changed c2_offset), and the list-cut test
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06_final; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_fir.py -q -m gpu -k "carried or run_multi" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_full.json $OUT/bench_full.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_final/bench.json').read())
print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['unplaced']['value'])
print({k:v[0] for k,v in d['summary'].items() if isinstance(v,list)})
PY

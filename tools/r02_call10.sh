#!/bin/bash
mkdir -p gpurun_out/c10
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c10/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/c10/rc.txt
tail -25 gpurun_out/c10/pytest.log
timeout 900 python bench.py --steps 6 --warmup 2 --captures 3 --cpu-seconds 5 > gpurun_out/c10/bench.json 2> gpurun_out/c10/bench.err; echo "bench rc=$?" | tee -a gpurun_out/c10/rc.txt
tail -5 gpurun_out/c10/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/c10/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j.get("verified",{}).get("pass"))
for k,v in j.get("more",{}).items():
    if k in ("anf1","exact_batch"): print(k, json.dumps(v)[:1200])
    else: print(k, v.get("value"), v.get("error"), v.get("bench_seconds"))
PY

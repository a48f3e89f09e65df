#!/bin/bash
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 2 --captures 3 --cpu-seconds 8 > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err; echo "bench rc=$?" | tee gpurun_out/c6/rc.txt
tail -5 gpurun_out/c6/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/c6/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j.get("verified",{}).get("pass"))
for k,v in j.get("more",{}).items():
    print(k, json.dumps(v)[:900])
print(json.dumps(j.get("cpu_baseline")))
PY

#!/bin/bash
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_notch.py tests/test_gpu_host_app.py -q -x > gpurun_out/c7/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/c7/rc.txt
tail -15 gpurun_out/c7/pytest.log
timeout 900 python bench.py --steps 6 --warmup 2 --captures 3 --cpu-seconds 5 > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err; echo "bench rc=$?" | tee -a gpurun_out/c7/rc.txt
tail -5 gpurun_out/c7/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/c7/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j.get("verified",{}).get("pass"))
for k,v in j.get("more",{}).items():
    if k in ("anf1","c3","c5_rescoped"): print(k, json.dumps(v)[:1500])
    else: print(k, v.get("value"), v.get("error"))
print(json.dumps(j.get("cpu_baseline")))
PY

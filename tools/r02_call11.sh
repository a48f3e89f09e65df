#!/bin/bash
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
for i in 1 2 3; do for cap in 3 6 2; do
timeout 300 python bench.py --steps 4 --warmup 1 --batches-per-step 24 --captures $cap --no-cpu --no-more > gpurun_out/c11/b_${cap}_$i.json 2> gpurun_out/c11/b_${cap}_$i.err
python - <<PY
import json
j=json.loads(open("gpurun_out/c11/b_${cap}_$i.json").read().strip().splitlines()[-1])
v=j["verified"]; print("cap $cap run $i:", j["value"], v["pass"], v["fir_bit_exact"], v.get("fir_diff"))
PY
done; done
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c11/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/c11/rc.txt
tail -25 gpurun_out/c11/pytest.log | cut -c1-300

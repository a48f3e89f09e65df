#!/bin/bash
mkdir -p gpurun_out/c17
run() { name=$1; shift; timeout 200 python bench.py --steps 6 --warmup 2 --batches-per-step 24 --no-cpu --no-more "$@" > gpurun_out/c17/$name.json 2> gpurun_out/c17/$name.err; echo "$name rc=$? $(python tools/bench_brief.py < gpurun_out/c17/$name.json 2>/dev/null) $(python -c "import json;j=json.loads(open('gpurun_out/c17/$name.json').read().strip().splitlines()[-1]);v=j['verified'];print(v['pass'],v.get('equal_decisions'),v.get('mean_abs_dcost'))")" | tee -a gpurun_out/c17/rc.txt; }
run t128w256 --captures 3
run t256w256 --captures 3 --tile-len 256
run t256w384 --captures 3 --tile-len 256 --tile-warmup 384
run t512w256 --captures 3 --tile-len 512
run t384w384 --captures 3 --tile-len 384 --tile-warmup 384
run t256w256c4 --captures 4 --tile-len 256
run t256w256c1 --captures 1 --tile-len 256
bash tools/timeline.sh --no-more --no-verify --captures 3 --batches-per-step 8 > gpurun_out/c17/tl3.log 2>&1
python tools/overlap.py gpurun_out/timeline.csv | tee gpurun_out/c17/overlap_cap3.txt

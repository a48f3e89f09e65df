#!/bin/bash
mkdir -p gpurun_out/c18
run() { name=$1; shift; timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-more "$@" > gpurun_out/c18/$name.json 2> gpurun_out/c18/$name.err; echo "$name rc=$? $(python tools/bench_brief.py < gpurun_out/c18/$name.json 2>/dev/null)" | tee -a gpurun_out/c18/rc.txt; }
run b64 --batches-per-step 24
run b128 --batch-msamples 128 --batches-per-step 12
run b64c3 --batches-per-step 24 --captures 3
run b128c3 --batch-msamples 128 --batches-per-step 12 --captures 3
run b128c2 --batch-msamples 128 --batches-per-step 12 --captures 2
bash tools/profile_bench.sh c18/prof 2>&1 | tail -25

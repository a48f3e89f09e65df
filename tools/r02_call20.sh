#!/bin/bash
mkdir -p gpurun_out/c20
run() { name=$1; shift; timeout 300 python bench.py --steps 6 --warmup 2 --batches-per-step 24 --no-cpu --no-more "$@" > gpurun_out/c20/$name.json 2> gpurun_out/c20/$name.err; echo "$name rc=$? $(python tools/bench_brief.py < gpurun_out/c20/$name.json 2>/dev/null)" | tee -a gpurun_out/c20/rc.txt; }
for rep in 1 2; do
run l32_$rep
LSDR_RX_LANES=64 run l64_$rep
LSDR_RX_LANES=16 run l16_$rep
done
LSDR_RX_LANES=64 run l64_c1 --captures 1
run l32_c1 --captures 1
timeout 900 python -m pytest tests/test_gpu_bench_pipeline.py tests/test_gpu_rx_tiled.py -q 2>&1 | tail -3

#!/bin/bash
mkdir -p gpurun_out/c27
python tools/c3_profile.py exact_batch 2>&1 | tail -1 | cut -c1-500
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c27/bench.json 2> gpurun_out/c27/bench.err; echo "rc=$?"
python tools/bench_brief.py < gpurun_out/c27/bench.json

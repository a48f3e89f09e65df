#!/bin/bash
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/c4/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/c4/rc.txt
tail -8 gpurun_out/c4/pytest.log
run() { name=$1; shift; timeout 200 python bench.py --steps 6 --warmup 2 --batches-per-step 24 --no-cpu --no-more "$@" > gpurun_out/c4/$name.json 2> gpurun_out/c4/$name.err; echo "$name rc=$? $(python tools/bench_brief.py < gpurun_out/c4/$name.json 2>/dev/null)" | tee -a gpurun_out/c4/rc.txt; }
run c1_nomask --captures 1
run c1_x16 --captures 1 --rx-cus 16 --cu-pattern xcd_major
run c1_i16 --captures 1 --rx-cus 16 --cu-pattern interleaved
run c1_x32 --captures 1 --rx-cus 32 --cu-pattern xcd_major
run c1_i32 --captures 1 --rx-cus 32 --cu-pattern interleaved
run c1_x8 --captures 1 --rx-cus 8 --cu-pattern xcd_major
run c2_nomask --captures 2
run c2_x16 --captures 2 --rx-cus 16 --cu-pattern xcd_major
run c2_x32 --captures 2 --rx-cus 32 --cu-pattern xcd_major
run c2_i32 --captures 2 --rx-cus 32 --cu-pattern interleaved
run c3_x32 --captures 3 --rx-cus 32 --cu-pattern xcd_major
run c3_nomask --captures 3
run c6_x32 --captures 6 --rx-cus 32 --cu-pattern xcd_major

#!/bin/bash
mkdir -p gpurun_out/c22
V=$PWD/tools/variants
python tools/fir_alone.py
for w in 6 8 10; do LSDR_HIP_LIB=$V/liblsdr_hip_t64.so LSDR_FIR_PERSIST=$w python tools/fir_alone.py; done
for w in 3 4 5; do LSDR_HIP_LIB=$V/liblsdr_hip_t128.so LSDR_FIR_PERSIST=$w python tools/fir_alone.py; done
run() { name=$1; shift; timeout 300 python bench.py --steps 6 --warmup 2 --batches-per-step 24 --no-cpu --no-more "$@" > gpurun_out/c22/$name.json 2> gpurun_out/c22/$name.err; echo "$name rc=$? $(python tools/bench_brief.py < gpurun_out/c22/$name.json 2>/dev/null)" | tee -a gpurun_out/c22/rc.txt; }
run base
LSDR_HIP_LIB=$V/liblsdr_hip_t64.so LSDR_FIR_PERSIST=8 run t64w8
LSDR_HIP_LIB=$V/liblsdr_hip_t128.so LSDR_FIR_PERSIST=4 run t128w4
LSDR_HIP_LIB=$V/liblsdr_hip_t64.so LSDR_FIR_PERSIST=8 run t64w8_c1 --captures 1
run base_c1 --captures 1

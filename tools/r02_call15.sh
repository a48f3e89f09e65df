#!/bin/bash
mkdir -p gpurun_out/c16
run() { name=$1; shift; timeout 200 python bench.py --steps 6 --warmup 2 --batches-per-step 24 --no-cpu --no-more --no-verify "$@" > gpurun_out/c16/$name.json 2> gpurun_out/c16/$name.err; echo "$name rc=$? $(python tools/bench_brief.py < gpurun_out/c16/$name.json 2>/dev/null)" | tee -a gpurun_out/c16/rc.txt; }
for rep in 1 2; do
run base_c1_$rep --captures 1
LSDR_HIP_LIB=$PWD/tools/variants/liblsdr_hip_fprio1.so run fprio1_c1_$rep --captures 1
LSDR_HIP_LIB=$PWD/tools/variants/liblsdr_hip_fprio3.so run fprio3_c1_$rep --captures 1
run base_c3_$rep --captures 3
LSDR_HIP_LIB=$PWD/tools/variants/liblsdr_hip_fprio1.so run fprio1_c3_$rep --captures 3
LSDR_HIP_LIB=$PWD/tools/variants/liblsdr_hip_fprio3.so run fprio3_c3_$rep --captures 3
done

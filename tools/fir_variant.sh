#!/bin/bash
# tools/fir_variant.sh NAME "-DFLAGS…" — liblsdr_hip_NAME.so = the shipped objects with fir_filter.hip rebuilt under the flags (experiments on k_fir_mfma_stream)
set -e
cd "$(dirname "$0")/../leansdr_amd/csrc"
mkdir -p ../../tools/variants/bound
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form $2 -c fir_filter.hip -o ../../tools/variants/bound/fir_$1.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/variants/liblsdr_hip_$1.so ../../tools/variants/bound/fir_$1.o $(ls *.o | grep -v '^fir_filter.o$')

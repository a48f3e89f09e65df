#!/bin/bash
# tools/fir_bound_occ.sh VARIANT… — the complex-tap (and real-tap) stream launch alone over 256 Mi samples from variant builds (tools/fir_variant.sh)
out=gpurun_out/fir_variants.txt; : > $out
run() { echo "## $*" >> $out; env "$@" FIR_ARITH=blk FIR_ALONE_MI=256 FIR_ALONE_REPS=100 timeout 300 python tools/fir_alone.py 2>&1 | tail -2 | sed -e 's/FIR_ARITH.*bit-exact/bit-exact/' >> $out; }
for v in "$@"; do
  L=LSDR_HIP_LIB=tools/variants/liblsdr_hip_$v.so
  run $L FIR_FREQ=0.001
  run $L
done
cat $out

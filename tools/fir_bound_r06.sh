#!/bin/bash
# tools/fir_bound_r06.sh — what bounds k_fir_mfma_stream with complex taps (c2_offset, the IV pass of anf1)?  The launch alone over
# 256 Mi samples with real and with complex taps, from the shipped library and from two builds of fir_filter.hip that leave one side out
# (-DLSDR_STREAM_NOLOAD: no sample loads — the matrix side alone; -DLSDR_STREAM_NOMFMA: one VALU op in place of each MFMA — the memory
# side alone; their outputs are garbage, "bit-exact False" is expected there).  Build: see profiles/r06_bench/README.md (fir_bound.txt).
out=gpurun_out/fir_bound.txt; : > $out
for lib in "" tools/variants/liblsdr_hip_noload.so tools/variants/liblsdr_hip_nomfma.so; do
  for freq in "" 0.001; do
    echo "## lib=${lib:-shipped} freq=${freq:-0}" >> $out
    env ${lib:+LSDR_HIP_LIB=$lib} ${freq:+FIR_FREQ=$freq} FIR_ARITH=blk FIR_ALONE_MI=256 FIR_ALONE_REPS=20,200 timeout 300 python tools/fir_alone.py 2>&1 | tail -3 >> $out
  done
done
cat $out

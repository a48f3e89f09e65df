#!/bin/bash
# debug: decision traces of the GPU viterbi_sync and the oracle for one mode
LSDR_VIT_DEBUG=1 LO_VIT_DEBUG=1 python tools/vit_modes.py $1 2> gpurun_out/vit_trace.txt
grep -c VIT gpurun_out/vit_trace.txt

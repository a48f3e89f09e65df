#!/bin/bash
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
python tools/notch_debug.py 2>&1 | tee gpurun_out/c8/notch_debug.txt
cd /tmp && rm -rf /tmp/np && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -- python /root/repo/tools/notch_debug.py > /tmp/np.log 2>&1
f=$(find /tmp/np -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200 | tee /root/repo/gpurun_out/c8/notch_kernel_stats.txt

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/c24
cd /tmp && rm -rf /tmp/c3p && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3p -- python /root/repo/tools/c3_profile.py c3 > /root/repo/gpurun_out/c24/c3.log 2>&1
f=$(find /tmp/c3p -name "*kernel_stats.csv" | head -1); cut -c1-150 "$f" | head -24 | tee /root/repo/gpurun_out/c24/c3_kernel_stats.txt
tail -2 /root/repo/gpurun_out/c24/c3.log | cut -c1-600

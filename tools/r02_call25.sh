#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/c25
timeout 900 python -m pytest tests/test_gpu_fec.py tests/test_gpu_host_app.py tests/test_gpu_tx.py -q 2>&1 | tail -4
python tools/c3_profile.py c3 2>&1 | tail -1 | cut -c1-400
LSDR_VIT_GENERIC=1 python tools/c3_profile.py c3 2>&1 | tail -1 | cut -c1-200
cd /tmp && rm -rf /tmp/c3p && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3p -- python /root/repo/tools/c3_profile.py c3 > /root/repo/gpurun_out/c25/c3.log 2>&1
f=$(find /tmp/c3p -name "*kernel_stats.csv" | head -1); cut -c1-130 "$f" | head -8

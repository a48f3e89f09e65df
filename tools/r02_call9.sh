#!/bin/bash
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
python tools/notch_debug.py 2>&1 | tee gpurun_out/c9/notch_debug.txt
timeout 600 python -m pytest tests/test_gpu_notch.py -q > gpurun_out/c9/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/c9/rc.txt
tail -5 gpurun_out/c9/pytest.log

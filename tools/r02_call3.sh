#!/bin/bash
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c3/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/c3/rc.txt
tail -15 gpurun_out/c3/pytest.log
for cap in 1 2; do
  bash tools/timeline.sh --no-more --no-verify --captures $cap --batches-per-step 8 > gpurun_out/c3/tl$cap.log 2>&1
  cp gpurun_out/timeline.csv gpurun_out/c3/timeline_cap$cap.csv
  python tools/overlap.py gpurun_out/c3/timeline_cap$cap.csv > gpurun_out/c3/overlap_cap$cap.txt 2>&1
  cat gpurun_out/c3/overlap_cap$cap.txt
done
cat gpurun_out/c3/rc.txt

#!/bin/bash
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rx_tiled.py tests/test_gpu_rx.py tests/test_gpu_host_app.py tests/test_gpu_ref_graph.py -q > gpurun_out/c5/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/c5/rc.txt
tail -8 gpurun_out/c5/pytest.log
timeout 300 python tools/rx_tol_report.py > gpurun_out/c5/tol.log 2>&1; echo "tol rc=$?" >> gpurun_out/c5/rc.txt
run() { name=$1; shift; timeout 200 python bench.py --steps 6 --warmup 2 --batches-per-step 24 --no-cpu --no-more "$@" > gpurun_out/c5/$name.json 2> gpurun_out/c5/$name.err; echo "$name rc=$? $(python tools/bench_brief.py < gpurun_out/c5/$name.json 2>/dev/null)" | tee -a gpurun_out/c5/rc.txt; }
run c1 --captures 1
LSDR_RX_NO_ARITH=1 run c1_lut --captures 1
run c2 --captures 2
run c3 --captures 3
run c4 --captures 4
run c6 --captures 6
run c1_w384 --captures 1 --tile-warmup 384
run c2_w384 --captures 2 --tile-warmup 384
run c2_w512 --captures 2 --tile-warmup 512
bash tools/timeline.sh --no-more --no-verify --captures 1 --batches-per-step 8 > gpurun_out/c5/tl1.log 2>&1
python tools/overlap.py gpurun_out/timeline.csv | tee gpurun_out/c5/overlap_cap1.txt

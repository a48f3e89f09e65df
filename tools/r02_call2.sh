#!/bin/bash
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
timeout 300 python tools/rx_tol_report.py > gpurun_out/c2/tol.log 2>&1; echo "tol rc=$?" > gpurun_out/c2/rc.txt
timeout 400 python bench.py --steps 10 --warmup 2 --cpu-seconds 6 > gpurun_out/c2/bench6.json 2> gpurun_out/c2/bench6.err; echo "bench6 rc=$?" >> gpurun_out/c2/rc.txt
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-more --captures 3 > gpurun_out/c2/bench3.json 2> gpurun_out/c2/bench3.err; echo "bench3 rc=$?" >> gpurun_out/c2/rc.txt
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-more --captures 2 > gpurun_out/c2/bench2.json 2> gpurun_out/c2/bench2.err; echo "bench2 rc=$?" >> gpurun_out/c2/rc.txt
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-more --captures 4 --tile-warmup 384 > gpurun_out/c2/bench4_w384.json 2> gpurun_out/c2/bench4_w384.err; echo "bench4w rc=$?" >> gpurun_out/c2/rc.txt
cat gpurun_out/c2/rc.txt

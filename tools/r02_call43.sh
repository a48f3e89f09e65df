mkdir -p gpurun_out/c43
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c43/bench.json 2> gpurun_out/c43/bench.err; echo "bench rc=$?"
python tools/bench_brief.py < gpurun_out/c43/bench.json
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/c43/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c43/pytest.log

mkdir -p gpurun_out/c56
timeout 200 python tools/chain_bench.py 8000 --tiled --viterbi --repeat 10 > gpurun_out/c56/chain.log 2>&1
timeout 100 python tools/chain_bench.py 8000 --tiled --viterbi --repeat 10 >> gpurun_out/c56/chain.log 2>&1
timeout 100 python tools/chain_bench.py 8000 --tiled --repeat 10 >> gpurun_out/c56/chain.log 2>&1
timeout 100 python tools/chain_bench.py 8000 --tiled --hs --repeat 10 >> gpurun_out/c56/chain.log 2>&1
timeout 100 python tools/chain_bench.py 8000 --viterbi --repeat 2 >> gpurun_out/c56/chain.log 2>&1
cut -c1-330 gpurun_out/c56/chain.log
timeout 300 python -m pytest tests/test_gpu_host_app.py tests/test_gpu_ref_graph.py -m gpu -x -q 2>&1 | tail -2

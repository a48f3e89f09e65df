mkdir -p gpurun_out/c40
timeout 200 python tools/chain_bench.py 8000 --tiled --viterbi --repeat 10 > gpurun_out/c40/chain.log 2>&1
timeout 100 python tools/chain_bench.py 8000 --tiled --viterbi --repeat 10 >> gpurun_out/c40/chain.log 2>&1
timeout 100 python tools/chain_bench.py 8000 --tiled --repeat 10 >> gpurun_out/c40/chain.log 2>&1
timeout 100 python tools/chain_bench.py 8000 --tiled --hs --repeat 10 >> gpurun_out/c40/chain.log 2>&1
cat gpurun_out/c40/chain.log | cut -c1-400

for rep in 1 2; do for w in 3 24 48 96; do
  echo -n "headline SWPC=$w rep $rep: "
  LSDR_MFMA_SWPC=$w timeout 200 python bench.py --steps 20 --warmup 5 --no-more --no-cpu 2>/dev/null | python tools/bench_brief.py | head -1
done; done
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -3

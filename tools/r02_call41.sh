timeout 200 python -m pytest tests/test_gpu_fec.py tests/test_gpu_host_app.py -m gpu -x -q 2>&1 | tail -3
timeout 100 python tools/more_one.py c3 2>&1 | cut -c1-80

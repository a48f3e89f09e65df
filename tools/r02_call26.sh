#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/c26
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/c26/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/c26/rc.txt
tail -4 gpurun_out/c26/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_bench.sh c26/prof 2>&1 | tail -12

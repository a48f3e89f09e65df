mkdir -p gpurun_out/c62
python tools/chain_bench.py 8000 --tiled --repeat 10 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --hip-trace --kernel-trace --stats -d /tmp/c62 -o app --output-format csv -- $GRAFT_REPO_ROOT/leansdr_amd/host/apps/leandvb_amd --u8 -f 2400e3 --sr 2000e3 --cr 1/2 --tiled < /tmp/cap_8000x10.u8 > /tmp/out.ts 2> $GRAFT_REPO_ROOT/gpurun_out/c62/app.err
cd $GRAFT_REPO_ROOT
f=$(find /tmp/c62 -name '*hip_api_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150
f=$(find /tmp/c62 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -10 "$f" | cut -c1-150
tail -3 gpurun_out/c62/app.err | cut -c1-200

cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/r06_prof2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd /tmp; rm -rf /tmp/prof_stats
LSDR_BENCH_UNPLACED=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$REPO/bench.py" --steps 60 --warmup 5 --no-cpu --no-more --no-verify > /tmp/prof_stats.log 2>&1
grep '^{"metric"' /tmp/prof_stats.log | tail -1 > "$OUT/bench_under_rocprof.json"
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -5 "$OUT/kernel_stats.csv" | cut -c1-200
cat "$OUT/bench_under_rocprof.json" | cut -c1-1500

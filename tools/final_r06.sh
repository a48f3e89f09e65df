#!/bin/bash
# tools/final_r06.sh — the round's closing measurements on one box: the default bench line, then the same command under rocprofv3 --kernel-trace --stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/r06_final; mkdir -p $OUT
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_full.json $OUT/bench_full.json 2>/dev/null
cut -c1-1800 $OUT/bench.json
cd /tmp; rm -rf /tmp/prof_stats
LSDR_BENCH_UNPLACED=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$REPO/bench.py" --steps 60 --warmup 5 --no-cpu --no-more --no-verify > /tmp/prof_stats.log 2>&1
grep '^{"metric"' /tmp/prof_stats.log | tail -1 > "$OUT/bench_under_rocprof.json"
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -4 "$OUT/kernel_stats.csv" | cut -c1-220
cut -c1-600 "$OUT/bench_under_rocprof.json"

#!/bin/bash
# tools/profile_bench.sh — the measurement set behind bench.py's `roofline` object, run on the GPU box:
#   gpurun_out/prof/bench.json, bench_one_capture.json, bench_no_overlap.json   the bench line (default: 3 captures per GPU /
#                                                              one capture / one capture, fir_filter and receiver back to back)
#   gpurun_out/prof/kernel_stats.csv                           rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/prof/pmc_fetch_size.csv, pmc_write_size.csv     one PMC counter per pass (rows of the fir kernel only)
# Copy what is to be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu --captures 1 > "$OUT/bench_one_capture.json" 2>> "$OUT/bench.err"
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-overlap > "$OUT/bench_no_overlap.json" 2>> "$OUT/bench.err"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu > /tmp/prof_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu > /tmp/prof_$c.log 2>&1
  f=$(find /tmp/prof_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "k_fir" "$f") > "$OUT/pmc_$(echo $c | tr A-Z a-z).csv"; fi
done
ls -la "$OUT"
tail -c 600 "$OUT/bench.json"

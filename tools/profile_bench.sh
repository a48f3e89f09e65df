#!/bin/bash
# tools/profile_bench.sh [outdir-name] — the measurement set behind bench.py's `roofline` object, run on the GPU box:
#   bench.json                      the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5), full line incl. `more`
#   kernel_stats.csv                rocprofv3 --kernel-trace --stats of the headline region (--no-more)
#   pmc_fetch_size.csv, pmc_write_size.csv   one PMC counter per pass (rows of the fir kernel only) -> pmc_traffic.json
#   timeline.csv, overlap.txt       kernel trace of a short run: fir stream occupancy, gaps, receiver kernels inside fir launches
#   viterbi_q4.txt                  viterbi_sync alone: lane = state kernel against k_viterbi_q4, SQ counters of the latter
# Copy what is to be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${1:-prof}
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu --no-more --no-verify > /tmp/prof_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -- python "$REPO/bench.py" --steps 1 --warmup 1 --batches-per-step 6 --no-cpu --no-more --no-verify > /tmp/prof_$c.log 2>&1
  f=$(find /tmp/prof_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "k_fir" "$f") > "$OUT/pmc_$(echo $c | tr A-Z a-z).csv"; fi
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, json, sys, os
out = sys.argv[1]
def avg(name):
    rows = [r for r in csv.DictReader(open(os.path.join(out, name))) if float(r["Counter_Value"]) > 0]
    vals = sorted(float(r["Counter_Value"]) for r in rows)
    big = [v for v in vals if v > 0.5 * vals[-1]]          # the full-size launches (not the acquisition ones)
    return sum(big) / len(big), len(big)
try:
    f, nf = avg("pmc_fetch_size.csv"); w, nw = avg("pmc_write_size.csv")
    j = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    caps = j["config"]["captures_per_gpu"]
    d = {"_comment": "HBM traffic of the dominant kernel from rocprofv3 PMC passes (one counter per pass: --pmc FETCH_SIZE, --pmc WRITE_SIZE) of "
                     "`bench.py --steps 1 --warmup 1 --batches-per-step 6 --no-cpu --no-more --no-verify`; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md "
                     "(gfx950 tallies 128-B requests at 64 B; calibrated in round 1 on tools/membench), WRITE_SIZE (KiB) uncorrected. The receiver kernels run "
                     "concurrently on other streams: their counters fall into whichever fir dispatch window they overlap. Averages over the full-size launches.",
         "kernel": "k_fir_persist<0,1,1,30> (one launch over the batches of all captures, lsdr_fir_filter_run_multi)",
         "batch_samples": j["config"]["batch_samples_per_capture"] * caps,
         "fetch_size_kib_per_launch": f, "fetch_correction": 2.0, "write_size_kib_per_launch": w, "launches_averaged": [nf, nw],
         "traffic_bytes_per_launch": int(f * 1024 * 2 + w * 1024)}
    json.dump(d, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("traffic", d["traffic_bytes_per_launch"], "algorithmic", j["roofline"]["algorithmic_bytes_per_launch"])
except Exception as e:
    print("pmc summary failed:", e)
PY
# ---- BASELINE config 1 / 4 (bench.py --workload c1): its own line, kernel statistics, HBM-side traffic of the whole chain
timeout 900 python bench.py --workload c1 --steps 20 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err"
cd /tmp; rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python "$REPO/bench.py" --workload c1 --steps 5 --warmup 1 --no-cpu --no-verify > /tmp/prof_c1.log 2>&1
f=$(find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/c1_kernel_stats.csv"
cd "$REPO"
timeout 900 python tools/pmc_traffic.py "$OUT/c1_pmc_traffic.json" -- python "$REPO/bench.py" --workload c1 --steps 1 --warmup 1 --no-cpu --no-verify --c1-captures 2 --c1-workers 1 > "$OUT/c1_pmc_traffic.txt" 2>&1
# ---- effective shader clock of fir_filter launches (GRBM_GUI_ACTIVE / duration: MI355X_MICROARCH.md "DVFS give-back"): a lone launch,
#      a burst of 20 after idle, 400 and 4000 back to back
cd /tmp; rm -rf /tmp/prof_clk
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_clk -- env FIR_ALONE_REPS=1,20,400,20 python "$REPO/tools/fir_alone.py" > "$OUT/fir_alone_pmc.log" 2>&1
f=$(find /tmp/prof_clk -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > "$OUT/fir_clock.txt" <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fir" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
d = [(float(r["Counter_Value"]) / 8 / ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
print("# fir_filter (k_fir_persist, 64 Mi samples per launch) under rocprofv3 --pmc GRBM_GUI_ACTIVE: effective clock = counter / 8 XCDs / duration")
print("# launch  clock_MHz  duration_us")
for i, (c, t) in enumerate(d):
    if i < 30 or i % 20 == 0:
        print(f"{i:6d} {c:9.0f} {t:10.1f}")
print(f"# median clock {statistics.median(c for c, _ in d):.0f} MHz, median duration {statistics.median(t for _, t in d):.1f} us over {len(d)} launches")
PY
cd "$REPO"
env FIR_ALONE_REPS=1,20,400,4000,20 python tools/fir_alone.py > "$OUT/fir_alone.txt" 2>&1
timeout 900 python tools/rx_tol_report.py > "$OUT/rx_tol_report.jsonl" 2>&1
timeout 600 bash tools/vit_q4_report.sh "$OUT/viterbi_q4.txt" > /dev/null 2>&1   # viterbi_sync alone, both kernels + SQ counters
bash tools/timeline.sh --no-more --no-verify --batches-per-step 8 > "$OUT/timeline.log" 2>&1
cp gpurun_out/timeline.csv "$OUT/timeline.csv"; python tools/overlap.py "$OUT/timeline.csv" > "$OUT/overlap.txt" 2>&1; cat "$OUT/overlap.txt"
ls -la "$OUT"
python tools/bench_brief.py < "$OUT/bench.json"

#!/bin/bash
# tools/profile_r06.sh [outdir-name] — round 6's measurement set (profile_r05.sh with the buffers placed by lsdr_arena and c1 on lsdr_capture_batch) behind bench.py's `roofline` object (run on the GPU box; copy what is
# to be judged into profiles/r06_bench/ afterwards):
#   bench.json            the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5), full line incl. `more`
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the headline region (--no-more --no-verify)
#   pmc_*.csv, pmc_traffic.json   FETCH_SIZE / WRITE_SIZE (one counter per pass, rows of the fir kernel) -> HBM bytes per launch
#   pmc_sq.txt            SQ counters of the headline's filter kernel (MFMA busy, VALU / LDS / MFMA instructions, wait classes)
#   overlap.txt           kernel trace of a short run: fir stream occupancy, receiver kernels inside fir launches, per-queue busy
#   rx_alone.txt          the receiver kernels of the same pipeline with the filter launches skipped
#   fir_alone.txt         the filter kernels alone (exact / fma / mfma / blk, real and complex taps)
#   c1.json, c1_kernel_stats.csv   bench.py --workload c1
#   anf1_timeline.txt     kernel trace of the default-graph pipeline (bench_more.anf1: lsdr_notch_fir + receiver)
#   membench.txt          tools/membench: the streaming-read ceiling of this chip (roofline.ceiling)
# The work-skipping hooks (LSDR_FIR_SKIP, LSDR_RX_SKIP, LSDR_RX_DBG) exist only in the measure build: the runs that use them load
# tools/variants/liblsdr_hip_measure.so through LSDR_HIP_LIB.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${1:-r06_prof}
mkdir -p "$OUT"
cd "$REPO"; export TMPDIR=/tmp
MEASURE_LIB=$REPO/tools/variants/liblsdr_hip_measure.so
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cp bench_full.json "$OUT/bench_full.json" 2>/dev/null
python tools/bench_brief.py < "$OUT/bench.json" 2>/dev/null | tail -25
cd /tmp; rm -rf /tmp/prof_stats
# (the stats run: 60 steps = 5 760 timed filter launches, no unplaced pass — what is left outside the timed region is the warm-up's 480 launches and
#  the placement probes' ≈ 900; its own JSON line is kept next to the statistics: the average there and `roofline.avg_launch_ms` here are one process's)
LSDR_BENCH_UNPLACED=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$REPO/bench.py" --steps 60 --warmup 5 --no-cpu --no-more --no-verify > /tmp/prof_stats.log 2>&1
grep '^{"metric"' /tmp/prof_stats.log | tail -1 > "$OUT/bench_under_rocprof.json"
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -8 "$OUT/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -- python "$REPO/bench.py" --steps 1 --warmup 1 --batches-per-step 6 --no-cpu --no-more --no-verify > /tmp/prof_$c.log 2>&1
  f=$(find /tmp/prof_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "k_fir" "$f") > "$OUT/pmc_$(echo $c | tr A-Z a-z).csv"; fi
done
rm -rf /tmp/prof_sq1 /tmp/prof_sq2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/prof_sq1 -- python "$REPO/bench.py" --steps 1 --warmup 1 --batches-per-step 6 --no-cpu --no-more --no-verify > /tmp/prof_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq2 -- python "$REPO/bench.py" --steps 1 --warmup 1 --batches-per-step 6 --no-cpu --no-more --no-verify > /tmp/prof_sq2.log 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import csv, json, sys, os, glob, collections
out = sys.argv[1]
def avg(name):
    rows = [r for r in csv.DictReader(open(os.path.join(out, name))) if float(r["Counter_Value"]) > 0]
    vals = sorted(float(r["Counter_Value"]) for r in rows)
    big = [v for v in vals if v > 0.5 * vals[-1]]          # the full-size launches (not the acquisition ones)
    return sum(big) / len(big), len(big), rows[-1]["Kernel_Name"][:80]
try:
    f, nf, kn = avg("pmc_fetch_size.csv"); w, nw, _ = avg("pmc_write_size.csv")
    j = json.load(open(os.path.join(out, "bench_full.json")))
    caps = j["config"]["captures_per_gpu"]
    d = {"_comment": "HBM traffic of the dominant kernel from rocprofv3 PMC passes (one counter per pass: --pmc FETCH_SIZE, --pmc WRITE_SIZE) of "
                     "`bench.py --steps 1 --warmup 1 --batches-per-step 6 --no-cpu --no-more --no-verify`; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md "
                     "(gfx950 tallies 128-B requests at 64 B), WRITE_SIZE (KiB) uncorrected. The receiver kernels run concurrently on another stream: "
                     "their counters fall into whichever fir dispatch window they overlap. Averages over the full-size launches.",
         "kernel": kn, "batch_samples": j["config"]["batch_samples_per_capture"] * caps,
         "fetch_size_kib_per_launch": f, "fetch_correction": 2.0, "write_size_kib_per_launch": w, "launches_averaged": [nf, nw],
         "traffic_bytes_per_launch": int(f * 1024 * 2 + w * 1024)}
    json.dump(d, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("traffic", d["traffic_bytes_per_launch"], "algorithmic", j["roofline"]["algorithmic_bytes_per_launch"], "ratio", d["traffic_bytes_per_launch"] / j["roofline"]["algorithmic_bytes_per_launch"])
except Exception as e:
    print("pmc summary failed:", e)
with open(os.path.join(out, "pmc_sq.txt"), "w") as o:
    o.write("# SQ counters per launch of the headline's kernels (rocprofv3 --pmc, two passes; `bench.py --steps 1 --warmup 1 --batches-per-step 6 --no-cpu --no-more --no-verify`);\n"
            "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per v_mfma_f32_16x16x4_f32), summed over the chip\n")
    for d in ("/tmp/prof_sq1", "/tmp/prof_sq2"):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                import re
                m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]); kn = (m.group(0) if m else r["Kernel_Name"])[:60]
                acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for kn, cs in acc.items():
                if not any(k in kn for k in ("k_fir", "k_rx")): continue
                for cn, v in sorted(cs.items()):
                    v2 = sorted(v); big = [x for x in v2 if x >= 0.5 * v2[-1]] or v2
                    o.write(f"{kn:62s} {cn:28s} mean {sum(big)/len(big):16.1f} over {len(big)} launches\n")
print(open(os.path.join(out, "pmc_sq.txt")).read()[:3000])
PY
bash tools/timeline.sh --no-more --no-verify --batches-per-step 8 > "$OUT/timeline.log" 2>&1
python tools/overlap.py gpurun_out/timeline.csv > "$OUT/overlap.txt" 2>&1; cat "$OUT/overlap.txt"
# the receiver chain alone: same pipeline, filter launches skipped after the buffers are filled (LSDR_FIR_SKIP)
LSDR_HIP_LIB=$MEASURE_LIB LSDR_FIR_SKIP=1 bash tools/timeline.sh --no-more --no-verify --batches-per-step 8 > /dev/null 2>&1
python - > "$OUT/rx_alone.txt" <<'PY'
import re, collections
d = collections.defaultdict(list)
for ln in open("gpurun_out/timeline.csv"):
    p = ln.split(",")
    m = re.search(r"(k_rx\w+)", p[0])
    if m: d[m.group(1)].append(int(p[2]) - int(p[1]))
print("receiver kernels of the C2 pipeline with the filter launches skipped (measure build, LSDR_FIR_SKIP=1): per batch")
for k, v in d.items():
    v = v[len(v) // 2:]
    print(f"  {k:28s} n={len(v):4d} mean {sum(v) / len(v) / 1e3:8.1f} us")
PY
cat "$OUT/rx_alone.txt"
{
for a in exact fma mfma blk; do FIR_ARITH=$a FIR_ALONE_REPS=20,400 timeout 120 python tools/fir_alone.py 2>&1 | tail -3; done
for a in exact fma mfma blk; do FIR_FREQ=0.0123 FIR_ARITH=$a FIR_ALONE_REPS=400 timeout 120 python tools/fir_alone.py 2>&1 | tail -1; done
LSDR_MFMA_STREAM=0 FIR_ARITH=blk FIR_ALONE_REPS=400 timeout 120 python tools/fir_alone.py 2>&1 | tail -1
LSDR_MFMA_SWPC=3 FIR_ARITH=blk FIR_ALONE_REPS=400 timeout 120 python tools/fir_alone.py 2>&1 | tail -1
} > "$OUT/fir_alone.txt" 2>&1; cat "$OUT/fir_alone.txt"
timeout 280 bash tools/nf_timeline.sh "gpurun_out/${1:-r06_prof}/anf1_timeline.txt" X=1 > /dev/null 2>&1; head -30 "$OUT/anf1_timeline.txt"
[ -x tools/membench ] && timeout 100 tools/membench > "$OUT/membench.txt" 2>&1; tail -4 "$OUT/membench.txt"
timeout 900 python bench.py --workload c1 --steps 20 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err"; echo "c1 rc=$?"
cp bench_full.json "$OUT/c1_full.json" 2>/dev/null
cd /tmp; rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python "$REPO/bench.py" --workload c1 --steps 5 --warmup 1 --no-cpu --no-verify > /tmp/prof_c1.log 2>&1
f=$(find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/c1_kernel_stats.csv"
ls -la "$OUT"

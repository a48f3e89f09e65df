#!/bin/bash
# tools/prof_c1.sh [outdir] [extra bench args…] — kernel statistics + a timeline of `bench.py --workload c1` (run on the GPU box)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${1:-c1_prof}; shift
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_c1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python "$REPO/bench.py" --workload c1 --steps 4 --warmup 1 --no-cpu --no-verify --no-single "$@" > "$OUT/bench.json" 2> "$OUT/prof.err"
f=$(find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
t=$(find /tmp/prof_c1 -name "*kernel_trace.csv" | head -1); cp "$t" "$OUT/kernel_trace.csv"
python - "$OUT" "$t" <<'PY'
import csv, sys
out, t = sys.argv[1], sys.argv[2]
for r in list(csv.DictReader(open(out + "/kernel_stats.csv")))[:22]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} avg_us {float(r["AverageNs"])/1e3:10.1f} {r["Percentage"]:>6s}%')
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last k_rxb_detect_fft pair backwards
t0 = int(rows[0]["Start_Timestamp"])
idx = [i for i, r in enumerate(rows) if "k_rxb_tiles" in r["Kernel_Name"]]
start = idx[-4] - 12 if len(idx) >= 4 else 0
with open(out + "/timeline.txt", "w") as f:
    for r in rows[max(start, 0):]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        line = f'{s/1e3:12.1f} us  +{(e-s)/1e3:9.1f} us  q{r.get("Queue_Id","?"):>3s}  {r["Kernel_Name"][:70]}'
        f.write(line + "\n")
print(open(out + "/timeline.txt").read()[-6000:])
PY

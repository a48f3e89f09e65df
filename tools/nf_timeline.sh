#!/bin/bash
# kernel timeline of the fused default-graph pipeline (bench_more.anf1) under rocprofv3 --kernel-trace: per-kernel means, the filter pass's
# gaps, what runs inside them.  Usage (GPU box): bash tools/nf_timeline.sh OUT.txt [ENV=VAL ...]
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tlnf
env "$@" LSDR_BENCH_MIN_SECONDS=0.15 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlnf -- python "$REPO/tools/more_one.py" anf1 > /tmp/tlnf.log 2>&1
f=$(find /tmp/tlnf -name "*kernel_trace.csv" | head -1)
python - "$f" "$REPO/$OUT" "$*" <<'PY'
import csv, sys, re, collections
import numpy as np
rows = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
def key(n):
    m = re.search(r"(k_\w+|__amd\w+)", n)
    return m.group(1) if m else n[:28]
pas = [(s, e, q) for n, s, e, q in rows if "k_fir_mfma_stream" in n and "ELi12ELi1" in n.replace(" ", "") or ("k_fir_mfma_stream<30, 1, 12, 1" in n)]
pas = [p for p in pas if p[1] - p[0] > 200000]
pas = pas[len(pas) // 2:]
o = open(sys.argv[2], "w")
def P(*a):
    print(*a); print(*a, file=o)
P("# env:", sys.argv[3])
t0, t1 = pas[0][0], pas[-1][1]
dur = np.array([e - s for s, e, q in pas]) / 1e3
gaps = np.array([pas[i + 1][0] - pas[i][1] for i in range(len(pas) - 1)]) / 1e3
P(f"filter pass: {len(pas)} launches, mean {dur.mean():.1f} us, start-to-start {np.diff([p[0] for p in pas]).mean() / 1e3:.1f} us, gap mean {gaps.mean():.1f} p50 {np.median(gaps):.1f} max {gaps.max():.1f} us; queues {sorted(set(p[2] for p in pas))}")
agg = collections.OrderedDict()
for n, s, e, q in rows:
    if s < t0 or e > t1:
        continue
    d = agg.setdefault((key(n), q), [0, 0])
    d[0] += 1; d[1] += e - s
for (k, q), (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    P(f"  {k:28s} queue {q:>3s} n={c:5d} mean {tot / c / 1e3:8.1f} us  per pass {tot / len(pas) / 1e3:8.1f} us")
# one typical gap: what ran between the end of pass i and the start of pass i+1
i = len(pas) // 2
P("between two passes:")
for n, s, e, q in rows:
    if e > pas[i][1] - 30000 and s < pas[i + 1][0] + 30000 and "k_rx" not in n:
        P(f"   {key(n):28s} q{q:>3s} start {(s - pas[i][1]) / 1e3:8.1f} us  end {(e - pas[i][1]) / 1e3:8.1f} us (relative to the end of the pass)")
PY

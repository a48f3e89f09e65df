#!/usr/bin/env python3
"""tools/overlap.py timeline.csv — from a rocprofv3 kernel trace of bench.py (tools/timeline.sh): how busy the fir stream is,
how long the gaps between consecutive fir launches are, and how much of the receiver kernels' time overlaps a fir launch."""
import re
import sys
import numpy as np
rows = []
for ln in open(sys.argv[1]):
    p = ln.rstrip("\n").split(",")
    rows.append((p[0], int(p[1]), int(p[2]), p[3] if len(p) > 3 else ""))
rows.sort(key=lambda r: r[1])
fir = [(s, e) for n, s, e, q in rows if "k_fir" in n and e - s > 50000]
if len(fir) > 40:
    fir = fir[len(fir) // 2:]          # the timed half
t0, t1 = fir[0][0], fir[-1][1]
busy = sum(e - s for s, e in fir)
gaps = np.array([fir[i + 1][0] - fir[i][1] for i in range(len(fir) - 1)]) / 1e3
print(f"fir launches {len(fir)}: mean {busy / len(fir) / 1e3:.1f} us, stream busy {busy / (t1 - t0):.3f}, gap mean {gaps.mean():.1f} us  p50 {np.median(gaps):.1f}  max {gaps.max():.1f}")
names = {}
for n, s, e, q in rows:
    if s < t0 or e > t1 or "k_fir" in n:
        continue
    m = re.search(r"(k_\w+|__amd\w+)", n)
    key = m.group(1) if m else n[:28]
    d = names.setdefault(key, [0, 0, 0])
    d[0] += 1; d[1] += e - s
    # overlap with any fir launch
    for fs, fe in fir:
        if fe <= s:
            continue
        if fs >= e:
            break
        d[2] += min(e, fe) - max(s, fs)
for k, (c, tot, ov) in sorted(names.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:30s} n={c:5d} mean {tot / c / 1e3:8.1f} us  total {tot / 1e6:8.2f} ms  inside a fir launch {ov / max(1, tot):.2f}")

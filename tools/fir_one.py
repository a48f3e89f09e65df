#!/usr/bin/env python3
"""tools/fir_one.py — run only the C2 fir_filter kernel a few times (profiling target)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 64) << 20
freq = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = capi.Ctx(0)
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_in = ctx.alloc(n * 8)
d_blk = ctx.upload(blk)
for r in range(n // len(blk)):
    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
d_out = ctx.alloc(n // 30 * 8 + 64)
c = capi.lowpass(312, np.float32((2e6 / 2) * (1 + 0.35 / 2) / 240e6))
f = capi.FirFilter(ctx, c, 30, in_scale=75.0)
if freq:
    f.set_freq(freq)
e0, e1 = ctx.event(), ctx.event()
f.run_dev(d_in.ptr, n, d_out.ptr, n // 30)
ctx.sync()
ctx.event_record(e0)
for _ in range(reps):
    cons, prod = f.run_dev(d_in.ptr, n, d_out.ptr, n // 30)
ctx.event_record(e1)
ms = ctx.event_elapsed_ms(e0, e1) / reps
print(f"fir C2 n={n} freq={freq}: {ms:.4f} ms/launch  {(cons*8+prod*8)/ms/1e6:.1f} GB/s")

#!/usr/bin/env python3
"""tools/placement_probe4.py — is a "fast" buffer fast because part of it survives in the 256 MB Infinity Cache from one launch to the next?
The headline's filter launch over the same buffer back to back vs alternating between two buffers (every launch then reads data last touched 4 GiB ago)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench
ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
n = 256 << 20
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_blk = ctx.upload(blk)
f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
bufs = []
for b in range(8):
    d = ctx.alloc(n * 8 + 4096)
    for r in range(n // len(blk)):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    bufs.append(d)
out = ctx.alloc(n // decim * 8 + 1024)
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
def t(ptrs, reps=12):
    for p in ptrs * 2:
        f.run_dev(p, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for i in range(reps):
        f.run_dev(ptrs[i % len(ptrs)], n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / reps) / 1e9
single = [t([d.ptr]) for d in bufs]
print("same buffer back to back, TB/s:", " ".join(f"{v:.2f}" for v in single))
order = np.argsort(single)[::-1]
fast = [bufs[i] for i in order[:2]]; slow = [bufs[i] for i in order[-2:]]
print(f"two fastest alternating: {t([d.ptr for d in fast]):.2f}   two slowest alternating: {t([d.ptr for d in slow]):.2f}   all eight in turn: {t([d.ptr for d in bufs], 16):.2f}")

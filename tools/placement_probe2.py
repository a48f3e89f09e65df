#!/usr/bin/env python3
"""tools/placement_probe2.py — is a buffer fast or slow as a whole?  The headline's filter launch over the quarters of six 2 GiB buffers."""
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
for b in range(6):
    d = ctx.alloc(n * 8 + 4096)
    for r in range(n // len(blk)):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    bufs.append(d)
out = ctx.alloc(n // decim * 8 + 1024)
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
def t(ptr, cnt):
    for _ in range(3):
        f.run_dev(ptr, cnt, out.ptr, cnt // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(8):
        f.run_dev(ptr, cnt, out.ptr, cnt // decim)
    ctx.event_record(e1); ctx.sync()
    return cnt * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / 8) / 1e9
for rnd in range(2):
    for k, d in enumerate(bufs):
        q = n // 4
        print(f"round {rnd} buffer {k} ({d.ptr:x}): whole {t(d.ptr, n):.2f} TB/s; quarters " + " ".join(f"{t(d.at(i * q * 8), q):.2f}" for i in range(4)), flush=True)

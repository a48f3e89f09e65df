#!/usr/bin/env python3
"""tools/placement_map2.py — separate 2 GiB allocations vs 2 GiB windows of ONE large allocation, same process: is a large arena always of the fast kind?"""
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
out = ctx.alloc(n // decim * 8 + 1024)
e0, e1 = ctx.event(), ctx.event()
def fill(buf, nbytes):
    for r in range(nbytes // blk.nbytes):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, buf.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    ctx.sync()
def t(ptr):
    for _ in range(3):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(6):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / 6) / 1e9
for phase in range(2):
    small = []
    for k in range(6):
        d = ctx.alloc(n * 8 + 4096); fill(d, n * 8); small.append(d)
    t(small[0].ptr)
    print(f"phase {phase}: six separate 2 GiB allocations:", " ".join(f"{t(d.ptr):.2f}" for d in small), flush=True)
    for G in (4, 16, 48):
        big = ctx.alloc((G << 30) + 4096); fill(big, G << 30)
        offs = sorted(set([0, 1, G // 2, G - 2]))
        print(f"phase {phase}: one {G} GiB allocation, 2 GiB windows at GiB offsets {offs}:", " ".join(f"{t(big.at(o << 30)):.2f}" for o in offs), flush=True)
        big.free()
    for d in small:
        d.free()

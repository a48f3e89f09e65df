#!/usr/bin/env python3
"""tools/placement_map.py — ONE 96 GiB allocation, the headline's filter launch over 2 GiB windows at 1 GiB steps: where are the fast places?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench
ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
n = 256 << 20
G = int(os.environ.get("MAP_GIB", 96))
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_blk = ctx.upload(blk)
big = ctx.alloc((G << 30) + 4096)
for r in range((G << 30) // blk.nbytes):
    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, big.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
ctx.sync()
f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
out = ctx.alloc(n // decim * 8 + 1024)
e0, e1 = ctx.event(), ctx.event()
def t(ptr):
    for _ in range(2):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(6):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / 6) / 1e9
t(big.ptr)
print(f"base {big.ptr:x}; TB/s of a 2 GiB window by offset (GiB):")
row = []
for off in range(0, G - 2 + 1):
    row.append(t(big.at(off << 30)))
    if len(row) == 16:
        print(f"  {off - 15:3d}..{off:3d}: " + " ".join(f"{v:.2f}" for v in row), flush=True); row = []
if row:
    print("  tail: " + " ".join(f"{v:.2f}" for v in row))
# finer: 64 MiB steps across the first transition found

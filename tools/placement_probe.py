#!/usr/bin/env python3
"""tools/placement_probe.py — does WHERE a 2 GiB input buffer lands decide how fast fir_filter streams it?  One process, several
buffers allocated one after the other (all kept), the headline's filter launch timed over each, twice round."""
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
bufs, outs = [], []
for b in range(int(os.environ.get("PROBE_BUFS", 6))):
    d = ctx.alloc(n * 8 + 4096)
    for r in range(n // len(blk)):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    bufs.append(d); outs.append(ctx.alloc(n // decim * 8 + 1024))
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
for rnd in range(2):
    row = []
    for d, o in zip(bufs, outs):
        for _ in range(2):
            f.run_dev(d.ptr, n, o.ptr, n // decim)
        ctx.sync(); ctx.event_record(e0)
        for _ in range(10):
            f.run_dev(d.ptr, n, o.ptr, n // decim)
        ctx.event_record(e1); ctx.sync()
        row.append(ctx.event_elapsed_ms(e0, e1) / 10)
    print(f"round {rnd}: ms per 256 Mi launch by buffer:", " ".join(f"{v:.4f}" for v in row), " | input ptrs", " ".join("%x" % d.ptr for d in bufs) if rnd == 0 else "", flush=True)
# same buffer pair, output buffer swapped: is it the input or the output that matters?
row = []
for k in range(len(bufs)):
    o = outs[(k + 1) % len(outs)]
    ctx.sync(); ctx.event_record(e0)
    for _ in range(10):
        f.run_dev(bufs[k].ptr, n, o.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    row.append(ctx.event_elapsed_ms(e0, e1) / 10)
print("outputs rotated by one:             ", " ".join(f"{v:.4f}" for v in row))

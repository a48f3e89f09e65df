#!/usr/bin/env python3
"""tools/size_probe.py — the headline's filter launch alone over buffers of several sizes: is the slow mode an aliasing of the eight XCDs'
contiguous input ranges (their distance is 1/8 of the batch: 256 Mi samples -> 2^28 - 2^16 bytes)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench
ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_blk = ctx.upload(blk)
f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
e0, e1 = ctx.event(), ctx.event()
for msamp in (256, 255, 250, 237, 200, 173, 128):
    n = msamp << 20
    row = []
    for b in range(3):
        d = ctx.alloc(n * 8 + 4096); o = ctx.alloc(n // decim * 8 + 1024)
        for r in range(n // len(blk)):
            capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
        for _ in range(3):
            f.run_dev(d.ptr, n, o.ptr, n // decim)
        ctx.sync(); ctx.event_record(e0)
        for _ in range(8):
            f.run_dev(d.ptr, n, o.ptr, n // decim)
        ctx.event_record(e1); ctx.sync()
        ms = ctx.event_elapsed_ms(e0, e1) / 8
        row.append(n * 8.0333 / ms / 1e9)
        d.free(); o.free()
    print(f"{msamp:4d} Mi samples (XCD ranges {n // 8 * 8 / 2**20:8.2f} MiB apart): TB/s on 8.03 B/sample by buffer:", " ".join(f"{v:.2f}" for v in row), flush=True)

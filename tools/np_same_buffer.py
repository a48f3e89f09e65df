#!/usr/bin/env python3
"""tools/np_same_buffer.py — one filter launch over 256 Mi samples with different rows per wave tile (LSDR_MFMA_NP / _NP_CP, read per create) and
workgroups queued per CU (LSDR_MFMA_SWPC), in ONE process on the SAME buffers, alternating: TB/s of algorithmic bytes.  NP_FREQ=f: complex taps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench
ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
n = 256 << 20
freq = float(os.environ.get("NP_FREQ", 0))
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_blk = ctx.upload(blk)
out = ctx.alloc(n // decim * 8 + 1024)
filt = {}
for cfg in os.environ.get("NP_CFGS", "8:96,4:96,4:192").split(","):
    np_, wpc = (int(v) for v in cfg.split(":"))
    os.environ["LSDR_MFMA_NP"] = os.environ["LSDR_MFMA_NP_CP"] = str(np_); os.environ["LSDR_MFMA_SWPC"] = str(wpc)
    f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
    if freq:
        f.set_freq(freq)
    filt[(np_, wpc)] = f
e0, e1 = ctx.event(), ctx.event()
def t(f, ptr):
    for _ in range(2):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(8):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / 8) / 1e9
bufs = []
for k in range(int(os.environ.get("NBUF", 8))):
    d = ctx.alloc(n * 8 + 4096)
    for r in range(n * 8 // blk.nbytes):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    ctx.sync(); bufs.append(d)
    row = []
    for rep in range(2):
        for key, f in filt.items():
            row.append(f"{key[0]}:{key[1]} {t(f, d.ptr):.2f}")
    print(f"buffer {k}:", "  ".join(row), flush=True)

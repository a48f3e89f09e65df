#!/usr/bin/env python3
"""tools/mode_by_buffer.py — the headline's filter launch over N separately allocated 2 GiB buffers with consecutive (chunked) and strided tile lists and
48 / 96 workgroups per CU queued (read per create), one process: does a buffer of the slow kind prefer another launch shape?  TB/s of algorithmic bytes."""
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
out = ctx.alloc(n // decim * 8 + 1024)
filt = {}
for chunk, wpc, np_ in ((1, 96, 8), (0, 96, 8), (0, 48, 8), (1, 48, 8), (1, 192, 4), (0, 24, 8)):
    os.environ["LSDR_MFMA_CHUNK"] = str(chunk); os.environ["LSDR_MFMA_SWPC"] = str(wpc); os.environ["LSDR_MFMA_NP"] = str(np_)
    filt[f"{'chunk' if chunk else 'strid'}{wpc}np{np_}"] = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
e0, e1 = ctx.event(), ctx.event()
def t(f, ptr):
    for _ in range(2):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(6):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / 6) / 1e9
bufs = []
for k in range(int(os.environ.get("NBUF", 14))):
    d = ctx.alloc(n * 8 + 4096)
    for r in range(n * 8 // blk.nbytes):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    ctx.sync(); bufs.append(d)
    print(f"buffer {k:2d}:", "  ".join(f"{key} {t(f, d.ptr):.2f}" for key, f in filt.items()), flush=True)

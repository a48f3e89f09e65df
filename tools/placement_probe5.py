#!/usr/bin/env python3
"""tools/placement_probe5.py — the headline's filter launch over 2 GiB inputs allocated with hipExtMallocWithFlags: default, fine-grained,
uncached, physically contiguous.  (Is the slow kind a cache policy or a physical layout?)"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
n = 256 << 20
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_blk = ctx.upload(blk)
f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
out = ctx.alloc(n // decim * 8 + 1024)
e0, e1 = ctx.event(), ctx.event()
def t(ptr):
    for _ in range(3):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(8):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / 8) / 1e9
for name, flags in (("default", 0), ("fine-grained", 1), ("uncached", 3), ("contiguous", 4), ("default", 0), ("contiguous", 4), ("uncached", 3)):
    row = []
    for k in range(4):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), n * 8 + 4096, flags)
        if rc:
            row.append(f"alloc rc={rc}"); break
        for r in range(n // len(blk)):
            capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, C.c_void_p(p.value + r * blk.nbytes), d_blk.ptr, blk.nbytes))
        ctx.sync()
        row.append(f"{t(p.value):.2f}")
    print(f"{name:14s} flags {flags}: TB/s by buffer (all kept):", " ".join(row), flush=True)

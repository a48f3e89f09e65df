#!/usr/bin/env python3
"""tools/va_align_probe.py — is a resident buffer's kind (fast / slow) a matter of its VIRTUAL address's alignment?  2 GiB hipMalloc buffers: pointer,
its alignment, TB/s of the headline's filter launch; then 5 GiB allocations and the launch over the 1 GiB / 2 GiB-aligned and a deliberately odd
(+ 2 MiB + 4 KiB) 2 GiB window inside each."""
import ctypes as C
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
f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
e0, e1 = ctx.event(), ctx.event()
def t(ptr, reps=6):
    p = C.c_void_p(ptr)
    for _ in range(2):
        f.run_dev(p, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(reps):
        f.run_dev(p, n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / reps) / 1e9
def fill(ptr, nbytes):
    for r in range(nbytes // blk.nbytes):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, C.c_void_p(ptr + r * blk.nbytes), d_blk.ptr, blk.nbytes))
    ctx.sync()
def tz(p):
    return (p & -p).bit_length() - 1
keep = []
for k in range(int(os.environ.get("NBUF", 10))):
    extra = (4096, 0, 2 << 20)[k % 3]
    d = ctx.alloc(n * 8 + extra); keep.append(d); fill(d.ptr, n * 8)
    print(f"hipMalloc(2 GiB + {extra:>7d}): ptr {d.ptr:#x} aligned to 2^{tz(d.ptr)}  {t(d.ptr):.2f} TB/s", flush=True)
for k in range(3):
    d = ctx.alloc(5 << 30); keep.append(d); fill(d.ptr, 5 << 30)
    a1 = (d.ptr + (1 << 30) - 1) & ~((1 << 30) - 1)
    a2 = (d.ptr + (2 << 30) - 1) & ~((2 << 30) - 1)
    odd = a1 + (2 << 20) + 4096
    print(f"hipMalloc(5 GiB): ptr {d.ptr:#x} (2^{tz(d.ptr)});  window at base {t(d.ptr):.2f}  1 GiB-aligned {t(a1):.2f}  2 GiB-aligned {t(a2):.2f}  odd {t(odd):.2f}", flush=True)

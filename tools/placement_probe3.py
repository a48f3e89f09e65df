#!/usr/bin/env python3
"""tools/placement_probe3.py — does de-phasing the eight XCDs' walks (LSDR_MFMA_XROT) make the slow buffers fast?"""
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
rots = ["strided", "chunked", "chunked wpc24", "chunked wpc96", "strided wpc3"]
firs = []
for r in rots:
    os.environ["LSDR_MFMA_CHUNK"] = "1" if "chunked" in r else "0"
    os.environ["LSDR_MFMA_SWPC"] = r.split("wpc")[1] if "wpc" in r else "48"
    firs.append(capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK))
bufs = []
for b in range(6):
    d = ctx.alloc(n * 8 + 4096)
    for r in range(n // len(blk)):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    bufs.append(d)
out = ctx.alloc(n // decim * 8 + 1024)
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
def t(f, ptr, cnt):
    for _ in range(3):
        f.run_dev(ptr, cnt, out.ptr, cnt // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(8):
        f.run_dev(ptr, cnt, out.ptr, cnt // decim)
    ctx.event_record(e1); ctx.sync()
    return cnt * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / 8) / 1e9
# results must not depend on the rotation: compare outputs
f0 = firs[0]; f0.run_dev(bufs[0].ptr, n, out.ptr, n // decim); ref = ctx.download(out, np.complex64, 1 << 20)
for f, r in zip(firs[1:], rots[1:]):
    f.run_dev(bufs[0].ptr, n, out.ptr, n // decim)
    assert np.array_equal(ctx.download(out, np.complex64, 1 << 20).view(np.uint64), ref.view(np.uint64)), r
print("walk", rots, "(TB/s on 8.03 B/sample; outputs identical)")
for rnd in range(2):
    for k, d in enumerate(bufs):
        print(f"round {rnd} buffer {k}: " + " ".join(f"{t(f, d.ptr, n):.2f}" for f in firs), flush=True)

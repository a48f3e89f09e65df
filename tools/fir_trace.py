#!/usr/bin/env python3
"""tools/fir_trace.py — per-phase cycle breakdown of the persistent fir kernel
(needs the -DLSDR_FIR_TRACE build: LSDR_HIP_LIB=tools/trace/liblsdr_hip_trace.so)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 64) << 20
freq = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ctx = capi.Ctx(0)
d_in = ctx.alloc(n * 8)
blk = (np.random.default_rng(0).standard_normal(1 << 23).astype(np.float32)).view(np.complex64)
d_blk = ctx.upload(blk)
for r in range(n // len(blk)):
    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
d_out = ctx.alloc(n // 30 * 8 + 64)
c = capi.lowpass(312, np.float32((2e6 / 2) * (1 + 0.35 / 2) / 240e6))
arith = {"exact": capi.FIR_EXACT, "fma": capi.FIR_FMA, "mfma": capi.FIR_MFMA, "blk": capi.FIR_MFMA_BLK}[os.environ.get("FIR_ARITH", "exact")]
f = capi.FirFilter(ctx, c, 30, in_scale=75.0, arith=arith)
if freq:
    f.set_freq(freq)
for _ in range(3):
    f.run_dev(d_in.ptr, n, d_out.ptr, n // 30)
ctx.sync()
mf = arith in (capi.FIR_MFMA, capi.FIR_MFMA_BLK)
is_blk = arith == capi.FIR_MFMA_BLK
W = int(os.environ.get("LSDR_MFMA_W", "2"))
nwg = (int(os.environ.get("LSDR_MFMA_WPC", "2" if W == 4 else "3")) if mf else int(os.environ.get("LSDR_FIR_PERSIST", "2"))) * 256
tr = np.zeros(nwg * 4 * 8, np.uint64)
capi.lib.lsdr_fir_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert capi.lib.lsdr_fir_trace_read(tr.ctypes.data, len(tr)) == 0
tr = tr.reshape(nwg, 4, 8).astype(np.float64)
if mf:
    tr = tr[:, :W, :]
tiles = (n // 30 // ((118 if is_blk else 128) * W if mf else 256)) / nwg
names = ["prologue issue", "wait loads + LDS write", "barrier A", "issue next loads (first part)" if mf else "issue next loads",
         "MFMA phase (+ later parts)" if mf else "taps", "store", "barrier B", "-"]
print(f"FIR_ARITH={os.environ.get('FIR_ARITH','exact')} W={W} workgroups={nwg}")
tot = tr.sum(axis=2).mean()
print(f"tiles/WG {tiles:.1f}; total cycles/wave {tot:.0f} ({tot/tiles:.0f}/tile)  [s_memtime ticks = 100 MHz? see ratio]")
for i, nm in enumerate(names[:7]):
    print(f"  {nm:26s} {tr[:, :, i].mean()/tiles:9.1f} per tile   ({100*tr[:, :, i].mean()/tot:5.1f} %)")

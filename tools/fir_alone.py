#!/usr/bin/env python3
"""tools/fir_alone.py — the C2 fir_filter launch alone (no receiver): ms per 64 Mi-sample batch, bit-exactness vs the oracle on a slice."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import leansdr_amd.capi as capi
import bench
import pyoracle as po
ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
n = int(os.environ.get("FIR_ALONE_MI", "64")) << 20      # Mi samples per launch
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_in = ctx.alloc(n * 8); d_blk = ctx.upload(blk)
for r in range(n // len(blk)):
    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
ctx.sync()
d_out = ctx.alloc(n // decim * 8 + 1024)
arith = {"exact": capi.FIR_EXACT, "fma": capi.FIR_FMA, "mfma": capi.FIR_MFMA, "blk": capi.FIR_MFMA_BLK}[os.environ.get("FIR_ARITH", "exact")]
f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=arith)
if os.environ.get("FIR_FREQ"):
    f.set_freq(float(os.environ["FIR_FREQ"]))
for _ in range(3):
    cons, prod = f.run_dev(d_in.ptr, n, d_out.ptr, n // decim)
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
for reps in [int(v) for v in os.environ.get("FIR_ALONE_REPS", "20").split(",")]:      # e.g. 1,20,400,4000: burst vs sustained
    ctx.sync()
    ctx.event_record(e0)
    for _ in range(reps):
        f.run_dev(d_in.ptr, n, d_out.ptr, n // decim)
    ctx.event_record(e1)
    ms = ctx.event_elapsed_ms(e0, e1) / reps
    print(f"  {reps} launches back to back: {ms:.4f} ms each = {(cons*8+prod*8)/ms/1e9:.2f} TB/s", flush=True)
y = ctx.download(d_out, np.complex64, 100000)
O = po.Oracle()
xs = np.tile(blk, 1)[:100000 * decim + len(coeffs)]
if arith == capi.FIR_MFMA_BLK:
    yr = O.fir_filter(coeffs, decim, xs, freq=f.current_freq, fma="blk", scale=75.0)[0][:100000]
else:
    yr = O.fir_filter(coeffs, decim, O.scaler(75.0, xs), freq=f.current_freq, fma=arith != capi.FIR_EXACT)[0][:100000]
print(f"FIR_ARITH={os.environ.get('FIR_ARITH','exact')} W={os.environ.get('LSDR_MFMA_W','-')} WPC={os.environ.get('LSDR_MFMA_WPC','-')} freq={f.current_freq} LSDR_FIR_PERSIST={os.environ.get('LSDR_FIR_PERSIST','2')} lib={os.path.basename(capi.LIB_PATH)}: {ms:.4f} ms per 64 Mi samples = {(cons*8+prod*8)/ms/1e9:.2f} TB/s; bit-exact vs the oracle ({'fmaf chain' if arith != capi.FIR_EXACT else 'reference arithmetic'}) {np.array_equal(y, yr)}")

#!/usr/bin/env python3
"""tools/nf_alone.py — the fused auto_notch + fir_filter block alone (no receiver): ms per run for several batch sizes / detect intervals."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench
ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
rng = np.random.default_rng(0)
period = 4193280
t = np.arange(period)
blk = ((rng.standard_normal(period) + 1j * rng.standard_normal(period)) * 0.7 + 2.0 * np.exp(2j * np.pi * round(0.0137 * period) / period * t)).astype(np.complex64)
for mb, dec_mult in [(64, 1), (256, 1), (256, 4), (256, 16), (128, 1)]:
    reps = (mb << 20) // period
    B = reps * period
    d_x = ctx.alloc((B + 2 * period) * 8); d_blk = ctx.upload(blk)
    for r in range(reps + 2):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_x.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    ctx.sync(); d_blk.free()
    n_out = B // decim
    d_out = ctx.alloc((n_out + 1024) * 8)
    nf = capi.NotchFir(ctx, coeffs, decim, in_scale=75.0, decimation=1024 * 4096 * dec_mult)
    F = 0
    def run():
        global F
        cons, prod = nf.run_dev(d_x.at((F % period) * 8), B + 400, d_out.ptr, n_out)
        F += cons
        return prod
    for _ in range(4):
        run()
    ctx.sync()
    nf.pass_time(True)
    e0, e1 = ctx.event(), ctx.event()
    ctx.event_record(e0)
    n = 12
    for _ in range(n):
        assert run() == n_out
    ctx.event_record(e1)
    ctx.sync()
    ms = ctx.event_elapsed_ms(e0, e1) / n
    pms, pl = nf.pass_time(False)
    print(f"batch {mb} Mi, detect every {dec_mult} x 4 Mi samples ({B // (4194304 * dec_mult)} per run): run {ms:.4f} ms = {B / ms / 1e6:.0f} GS/s; filter pass {pms:.4f} ms "
          f"({B * 8.0333 / pms / 1e9:.2f} TB/s on 8.03 B/sample); rest {ms - pms:.4f} ms; bin {nf.bin()} WPC={os.environ.get('LSDR_NF_WPC', '4')}", flush=True)
    nf.close(); d_x.free(); d_out.free()

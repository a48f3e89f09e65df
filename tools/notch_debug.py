#!/usr/bin/env python3
"""tools/notch_debug.py — where the scan-mode notch differs from the oracle (error profile per block), and its speed alone."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import leansdr_amd.capi as capi
import pyoracle as po
O = po.Oracle()
ctx = capi.Ctx(0)
rng = np.random.default_rng(11)
n = 4096 * 300
t = np.arange(n)
x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 12 + 70 * np.exp(2j * np.pi * 0.0713 * t)
     + 40 * np.exp(-2j * np.pi * 0.27 * t) + 25 * np.exp(2j * np.pi * 0.4 * t)).astype(np.complex64)
for ns in (1, 2, 3):
    want, wbins = O.auto_notch(x, ns, 4096 * 100)
    a = capi.AutoNotch(ctx, ns, 0.0, 4096 * 100, mode=capi.NOTCH_SCAN)
    got = a.run(x)
    print("nslots", ns, "bins", a.bins(), wbins)
    a.close()
    err = np.abs(got - want).reshape(-1, 4096).max(axis=1) / np.abs(want).max()
    top = np.argsort(err)[-6:][::-1]
    print("  max err %.3e; worst blocks:" % err.max(), [(int(b), float("%.2e" % err[b])) for b in top])
    print("  err by block range: 0-99 %.2e  100-199 %.2e  200-299 %.2e" % (err[:100].max(), err[100:200].max(), err[200:].max()))
    b = int(top[0]); d = np.abs(got - want)[b * 4096:(b + 1) * 4096]
    print("  inside worst block: first 64 max %.2e, last 64 max %.2e, argmax %d" % (d[:64].max(), d[-64:].max(), int(d.argmax())))
# speed alone: 64 Mi samples, default decimation
nb = 16384
xin = ctx.alloc(nb * 4096 * 8); xo = ctx.alloc(nb * 4096 * 8)
seg = ctx.upload(x[:4096 * 256])
for r in range(nb // 256):
    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, xin.at(r * 256 * 4096 * 8), seg.ptr, 256 * 4096 * 8))
ctx.sync()
for mode, name in ((capi.NOTCH_SCAN, "scan"),):
    a = capi.AutoNotch(ctx, 1, 0.0, mode=mode)
    a.run_dev(xin.ptr, nb * 4096, xo.ptr, nb * 4096); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        a.run_dev(xin.ptr, nb * 4096, xo.ptr, nb * 4096)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    print(name, "64 Mi samples: %.3f ms = %.1f GS/s = %.2f TB/s" % (dt * 1e3, nb * 4096 / dt / 1e9, nb * 4096 * 16 / dt / 1e12), "bins", a.bins())
    a.close()

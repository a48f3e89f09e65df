#!/usr/bin/env python3
"""tools/vit_alone.py [msymbols] — viterbi_sync alone on a resident stream of framed QPSK 1/2 (or LSDR_VA_8PSK=1: 8PSK 2/3) soft
symbols: ms per call and symbols/s, nothing else on the GPU (LSDR_VIT_TIMING=1 adds the per-call round statistics)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench_more

ctx = capi.Ctx(0)
psk8 = bool(os.environ.get("LSDR_VA_8PSK"))
cstln, rate = (capi.PSK8, capi.FEC23) if psk8 else (capi.QPSK, capi.FEC12)
x, ts8 = bench_more.framed_period(capi, ctx, cstln, rate, 4, 24.0 if psk8 else 20.0, seed=3)
rx = capi.CstlnReceiver(ctx, sampler=capi.SAMP_LINEAR, cstln=cstln, fec=rate, omega=4.0, pll_adjustment=1 / 6.0)
o = rx.run(np.tile(x * np.float32(75.0), 6), meas=False)
sym = o["sym"][len(o["sym"]) // 3:]           # locked part
rx.close()
per = len(x) // 4
sym = sym[: len(sym) // per * per]
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 4) << 20
reps = max(1, n // len(sym))
big = np.tile(sym, reps)
d_in = ctx.upload(big)
d_out = ctx.alloc(len(big))
v = capi.Viterbi(ctx, cstln, rate)
for _ in range(3):
    c, p = v.run_dev(d_in.ptr, len(big), d_out.ptr, len(big))
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 0.5:
    c, p = v.run_dev(d_in.ptr, len(big), d_out.ptr, len(big)); k += 1
dt = (time.perf_counter() - t0) / k
print(f"{'8PSK 2/3' if psk8 else 'QPSK 1/2'}: {len(big)} symbols per call, consumed {c}: {dt*1e3:.3f} ms per call = {c/dt/1e9:.2f} G symbols/s; stats {v.stats()}")

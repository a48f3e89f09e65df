#!/usr/bin/env python3
"""tools/rx_one.py — run only the tiled cstln_receiver on a C2-like decimated stream (profiling target)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
from leansdr_amd import synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
x, _ = synth.qpsk_baseband(4 * 560000, 4, seed=5, rms=50.0, snr_db=20.0)
ctx = capi.Ctx(0)
acq = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0)
acq.run(x[:65536], meas=False)
r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0, mode=capi.RX_TILED, tile_len=L, tile_warmup=W)
r.set_state(acq.state())
d = ctx.upload(x)
o = ctx.alloc(len(x) * 4)
r.run_dev(d.ptr, len(x), o.ptr, len(x), meas=False)
t0 = time.perf_counter()
for _ in range(reps):
    res = r.run_dev(d.ptr, len(x), o.ptr, len(x), meas=False)
dt = (time.perf_counter() - t0) / reps
print(f"rx tiled L={L} W={W} lanes={os.environ.get('LSDR_RX_LANES','4')}: {dt*1e3:.4f} ms/run, {res['produced']} symbols, {r.tiled_stats()}")

#!/usr/bin/env python3
"""tools/rx_trace.py — phase breakdown of the symbol body of k_rx_tiles (trace build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
from leansdr_amd import synth
x, _ = synth.qpsk_baseband(4 * 600000, 4, seed=5, rms=50.0, snr_db=20.0)
ctx = capi.Ctx(0)
acq = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0)
acq.run(x[:65536], meas=False)
r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0, mode=capi.RX_TILED, tile_len=256, tile_warmup=512)
r.set_state(acq.state())
d = ctx.upload(x)
o = ctx.alloc(len(x) * 4)
for _ in range(2):
    res = r.run_dev(d.ptr, len(x), o.ptr, len(x), meas=False)
p = np.zeros(8, np.uint64)
assert capi.lib.lsdr_rx_probe_read(C.c_void_p(p.ctypes.data)) == 0
n = float(p[0])
print("symbols probed", n, "stats", r.tiled_stats())
for i, nm in [(1, "interp (samples + trig + math)"), (2, "constellation LUT gather"), (3, "PLL + timing update"), (4, "between symbols (skip loop, emit, chunk end)"), (5, "probe overhead (per probe)")]:
    print(f"  {nm:46s} {p[i]/n:8.1f} cycles/symbol")
# the exact serial kernel, same probes
capi.lib.lsdr_rx_probe_reset()
ser = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0)
ser.set_state(acq.state())
ser.run(x[:400000], meas=False)
assert capi.lib.lsdr_rx_probe_read(C.c_void_p(p.ctypes.data)) == 0
n = float(p[0])
print("serial kernel: symbols probed", n)
for i, nm in [(1, "interp (samples + trig + math)"), (2, "constellation LUT gather"), (3, "PLL + timing update"), (4, "between symbols (skip loop, emit, chunk end)"), (5, "probe overhead (per probe)")]:
    print(f"  {nm:46s} {p[i]/n:8.1f} cycles/symbol")

#!/usr/bin/env python3
"""tools/rx_quality.py — tiled cstln_receiver: time and agreement with the exact serial receiver over a (tile_len, warm-up) grid."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
from leansdr_amd import synth
snr = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nsym = int(sys.argv[2]) if len(sys.argv) > 2 else 560000
x, _ = synth.qpsk_baseband(4 * nsym, 4, seed=5, rms=50.0, snr_db=snr)
ctx = capi.Ctx(0)
acq = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0)
acq.run(x[:65536], meas=False)
st = acq.state()
ser = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0)
ser.set_state(st)
t0 = time.perf_counter()
ref = ser.run(x, meas=False)["sym"]
print(f"snr {snr} dB: serial {len(ref)} symbols in {(time.perf_counter()-t0)*1e3:.1f} ms")
d = ctx.upload(x)
o = ctx.alloc(len(x) * 4)
for L, W in [(256, 512), (256, 256), (128, 256), (128, 128), (64, 128), (64, 64), (256, 128), (512, 64), (128, 384), (64, 256)]:
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0, mode=capi.RX_TILED, tile_len=L, tile_warmup=W)
    r.set_state(st)
    res = r.run_dev(d.ptr, len(x), o.ptr, len(x), meas=False)
    got = ctx.download(o, ref.dtype, res["produced"])
    stats = r.tiled_stats()
    ts = []
    for _ in range(3):
        r.set_state(st)
        t0 = time.perf_counter(); r.run_dev(d.ptr, len(x), o.ptr, len(x), meas=False); ts.append(time.perf_counter() - t0)
    n = min(len(ref), len(got))
    same_sym = float((ref["symbol"][:n] == got["symbol"][:n]).mean())
    dcost = np.abs(ref["cost"][:n].astype(int) - got["cost"][:n].astype(int))
    print(f"L={L:4d} W={W:4d}: {min(ts)*1e3:7.3f} ms  produced {res['produced']} (serial {len(ref)})  symbol agree {same_sym:.6f}  "
          f"cost: exact {float((dcost==0).mean()):.4f} mean|d| {dcost.mean():.2f} max {dcost.max()}  stats {stats}")

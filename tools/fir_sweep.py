#!/usr/bin/env python3
"""tools/fir_sweep.py — time the fir_filter kernel variants on one GPU (HIP events).
Prints achieved algorithmic GB/s per variant plus a device copy as the practical ceiling."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi  # noqa: E402


def time_fir(ctx, d_in, n_in, d_out, cap, coeffs, decim, reps=10, env=None, freq=0.0, arith=0, in_format=0, scale=75.0):
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    try:
        f = capi.FirFilter(ctx, coeffs, decim, in_format=in_format, in_scale=scale, arith=arith)
    finally:
        for k in env:
            del os.environ[k]
    if freq:
        f.set_freq(freq)
    for _ in range(2):
        cons, prod = f.run_dev(d_in.ptr, n_in, d_out.ptr, cap)
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.event_record(e0)
    for _ in range(reps):
        cons, prod = f.run_dev(d_in.ptr, n_in, d_out.ptr, cap)
    ctx.event_record(e1)
    ms = ctx.event_elapsed_ms(e0, e1) / reps
    f.close()
    bytes_in = cons * (2 if in_format else 8)
    return ms, (bytes_in + prod * 8) / ms / 1e6, cons / ms / 1e3


def main():
    n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 64 << 20
    ctx = capi.Ctx(0)
    rng = np.random.default_rng(0)
    blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
    d_in = ctx.alloc(n * 8)
    d_blk = ctx.upload(blk)
    for r in range(n // len(blk)):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
    ctx.sync()
    d_out = ctx.alloc(n * 8)
    # practical ceiling: device copy of the same buffer
    e0, e1 = ctx.event(), ctx.event()
    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_out.ptr, d_in.ptr, n * 8))
    ctx.sync()
    ctx.event_record(e0)
    for _ in range(5):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_out.ptr, d_in.ptr, n * 8))
    ctx.event_record(e1)
    ms = ctx.event_elapsed_ms(e0, e1) / 5
    print(f"copy d2d {n*8/1e6:.0f} MB: {ms:.3f} ms  read+write {2*n*8/ms/1e6:.0f} GB/s")

    c313 = capi.lowpass(312, np.float32((2e6 / 2) * (1 + 0.35 / 2) / 240e6))
    rows = [
        ("C2 spec real exact", c313, 30, {}, 0.0, 0, 0),
        ("C2 spec complex exact (shifted)", c313, 30, {}, 0.0123, 0, 0),
        ("C2 spec real FMA", c313, 30, {}, 0.0, 1, 0),
        ("C2 spec complex FMA", c313, 30, {}, 0.0123, 1, 0),
        ("C2 generic real exact R=1", c313, 30, {"LSDR_FIR_GENERIC": "1"}, 0.0, 0, 0),
        ("C2 generic complex exact", c313, 30, {"LSDR_FIR_GENERIC": "1"}, 0.0123, 0, 0),
        ("C2 u8 input spec real", c313, 30, {}, 0.0, 0, 1),
        ("N=31 D=1 spec real (R=4)", capi.lowpass(30, np.float32(0.2)), 1, {}, 0.0, 0, 0),
        ("N=64 D=4 spec real (R=4)", capi.lowpass(63, np.float32(0.1)), 4, {}, 0.0, 0, 0),
        ("N=105 D=10 spec real (R=2)", capi.lowpass(104, np.float32(0.04)), 10, {}, 0.0, 0, 0),
    ]
    for name, c, d, env, freq, arith, fmt in rows:
        ms, gbs, msps = time_fir(ctx, d_in, n, d_out, n, c, d, env=env, freq=freq, arith=arith, in_format=fmt)
        print(f"{name:36s} {ms:8.3f} ms  {gbs:8.1f} GB/s algorithmic  {msps/1e3:8.2f} GS/s")
    ctx.close()


if __name__ == "__main__":
    main()

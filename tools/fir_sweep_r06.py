#!/usr/bin/env python3
"""tools/fir_sweep_r06.py — the decimating fir_filter alone over the decimations leandvb --resample can ask for (decim = Fs / (4·Fm),
order = resample_rej·Fs / (22·transition), leandvb.cc:353-378), LSDR_FIR_MFMA_BLK (k_fir_mfma_stream) with real and with complex taps
next to the exact kernel: ms per 64 Mi-sample launch, algorithmic TB/s (8 B read per sample + 8/D written), and bit-exactness against the
oracle's statement of each arithmetic on a slice.  GPU box; writes gpurun_out/r06_fir_sweep.txt (copied to profiles/r06_bench/)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import leansdr_amd.capi as capi
import pyoracle as po

DS = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 20, 24, 25, 30, 32, 33, 34, 36, 40, 48, 50, 60, 64]
ctx = capi.Ctx(0)
O = po.Oracle()
MI = int(os.environ.get("FIR_SWEEP_MI", "64"))      # Mi samples per launch (bench.py's C2 batch: 256)
n = MI << 20
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_in = ctx.alloc(n * 8); d_blk = ctx.upload(blk)
for r in range(n // len(blk)):
    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * blk.nbytes), d_blk.ptr, blk.nbytes))
ctx.sync()
d_out = ctx.alloc(n // 2 * 8 + 4096)
e0, e1 = ctx.event(), ctx.event()
lines = [f"# fir_filter alone, {MI} Mi cf32 samples per launch, 20 launches back to back (HIP events); leandvb's low-pass for Fs/Fm = 4·D\n"
         "# D  ncoeffs  tap_blocks | blk real taps: ms TB/s frac(8 TB/s) exact_bits | blk complex taps (freq 0.0123): ms TB/s frac bits | exact kernel: ms TB/s"]
for D in DS:
    fs_over_fm = 4.0 * D
    order = int(10.0 * fs_over_fm / (22 * 0.5 * 0.35)); order = (order + 1) // 2 * 2
    co = capi.lowpass(order, np.float32(0.5 * (1 + 0.35 / 2) / fs_over_fm))
    N = len(co)
    row = [f"{D:3d} {N:5d} {(N + D - 1) // D:3d}"]
    for arith, freq in ((capi.FIR_MFMA_BLK, 0.0), (capi.FIR_MFMA_BLK, 0.0123)) + (((capi.FIR_EXACT, 0.0),) if not os.environ.get("FIR_SWEEP_NOEXACT") else ()):
        try:
            f = capi.FirFilter(ctx, co, D, in_scale=75.0, arith=arith)
        except Exception as e:
            row.append(f"| refused ({str(e)[-60:]})"); continue
        if freq:
            f.set_freq(freq)
        for _ in range(3):
            cons, prod = f.run_dev(d_in.ptr, n, d_out.ptr, n // D)
        ctx.sync(); ctx.event_record(e0)
        for _ in range(20):
            f.run_dev(d_in.ptr, n, d_out.ptr, n // D)
        ctx.event_record(e1)
        ms = ctx.event_elapsed_ms(e0, e1) / 20
        tbs = (cons * 8 + prod * 8) / ms / 1e9
        m = min(prod, 40000)
        y = ctx.download(d_out, np.complex64, m)
        xs = blk[: m * D + N]
        if arith == capi.FIR_MFMA_BLK:
            yr = O.fir_filter(co, D, xs, freq=freq, fma="blk", scale=75.0)[0][:m]
        else:
            yr = O.fir_filter(co, D, O.scaler(75.0, xs), freq=freq)[0][:m]
        row.append(f"| {ms:7.4f} {tbs:5.2f} {tbs / 8:5.3f} {bool(np.array_equal(y, yr))}")
        f.close()
    lines.append(" ".join(row)); print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", f"r06_fir_sweep_{MI}Mi.txt"), "w").write("\n".join(lines) + "\n")

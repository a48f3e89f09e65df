#!/usr/bin/env python3
"""tools/c1_rx_probe.py — the C1 receiver alone (cu8 at 1.2 samples/symbol, device-resident): ms per run and GS/s of the tiled
cstln_receiver by input format (cu8 fused / cconverter + cf32), tile geometry, run size and number of concurrent captures.
  python tools/c1_rx_probe.py [msamples ...]      env: C1P_GEOS="1024:512,4096:512" C1P_CAPS="1,4" C1P_FMT="u8,f32" """
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench_more

lib = capi.lib
ctx0 = capi.Ctx(0)
x, ts = bench_more.framed_period(capi, ctx0, capi.QPSK, capi.FEC12, 6, 20.0, seed=5, decim=5, groups=5)
P = len(x)
u8 = capi.cconv_f32_u8(ctx0, x * np.float32(75.0)).reshape(-1)
sizes = [int(a) for a in sys.argv[1:]] or [64, 256]
geos = [tuple(int(v) for v in g.split(":")) for g in os.environ.get("C1P_GEOS", "1024:512,2048:512,4096:512").split(",")]
caps = [int(v) for v in os.environ.get("C1P_CAPS", "1,4").split(",")]
fmts = os.environ.get("C1P_FMT", "u8,f32").split(",")
OMEGA = 1.2
rx_kw = dict(sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, fec=capi.FEC12, omega=OMEGA, meas_decimation=1 << 22)

for ms in sizes:
    reps = max(1, (ms << 20) // P)
    B = P * reps
    extra = 8192
    d_in = ctx0.alloc((B + extra) * 2)
    dp = ctx0.upload(u8)
    for r in range(reps):
        capi.check(lib.lsdr_memcpy_d2d(ctx0.h, d_in.at(r * P * 2), dp.ptr, P * 2))
    capi.check(lib.lsdr_memcpy_d2d(ctx0.h, d_in.at(reps * P * 2), dp.ptr, extra * 2))
    ctx0.sync(); dp.free()
    sym_cap = int(B * 0.94) + 65536          # the tiled run reserves ⌈128/(omega−0.1)⌉+3 symbol slots per chunk
    # acquisition state once (serial, cu8)
    acq = capi.CstlnReceiver(ctx0, mode=capi.RX_SERIAL, in_format=capi.IN_CU8, **rx_kw)
    d_tmp = ctx0.alloc((1 << 20) * 4)
    acq.run_dev(d_in.ptr, 1 << 19, d_tmp.ptr, 1 << 20, meas=False)
    st = acq.state(); acq.close(); d_tmp.free()
    for fmt in fmts:
        d_cf = ctx0.alloc((B + extra) * 8) if fmt == "f32" else None
        for (tl, tw) in geos:
            for ncap in caps:
                if ncap * sym_cap * 4 > (150 << 30):
                    continue
                ctxs = [capi.Ctx(0) for _ in range(ncap)]
                rxs, outs = [], []
                for c in ctxs:
                    r = capi.CstlnReceiver(c, mode=capi.RX_TILED, tile_len=tl, tile_warmup=tw, in_format=capi.IN_CF32 if fmt == "f32" else capi.IN_CU8,
                                           out_format=capi.SYM_HARD2 if fmt == "u8h" else capi.SYM_SOFT, **rx_kw)
                    r.set_state(st)
                    rxs.append(r); outs.append(c.alloc(sym_cap * 4 if fmt != "u8h" else sym_cap // 4 + 64))

                def once():
                    for c, r, o in zip(ctxs, rxs, outs):
                        if fmt == "f32":
                            capi.check(lib.lsdr_cconverter_u8_run(c.h, d_in.ptr, B + extra, d_cf.ptr))
                            r.run_async(d_cf.ptr, B + extra, o.ptr, sym_cap)
                        elif fmt == "u8h":
                            r.run_async_hs2(d_in.ptr, B + extra, o.ptr, 0, sym_cap)
                        else:
                            r.run_async(d_in.ptr, B + extra, o.ptr, sym_cap)
                    return [r.wait() for r in rxs]
                once()
                t0 = time.perf_counter()
                n = 3
                for _ in range(n):
                    prod = once()
                dt = (time.perf_counter() - t0) / n
                stt = rxs[0].tiled_stats()
                print(f"B={ms:4d}Mi fmt={fmt} tile={tl}+{tw} caps={ncap}: {dt*1e3:8.3f} ms per round = {ncap*B/dt/1e9:7.2f} GS/s  "
                      f"(symbols {prod[0]}, tiles {stt['tiles']}, dup {stt['dup']} miss {stt['miss']} bad {stt['bad_seams']})", flush=True)
                for r in rxs:
                    r.close()
                for c, o in zip(ctxs, outs):
                    o.free(); c.close()
        if d_cf:
            d_cf.free()
    d_in.free()

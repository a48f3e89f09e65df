#!/usr/bin/env python3
"""tools/gen_bench.py — throughput of the generator side (leandvbtx + leanchansim blocks) on the GPU, device-resident, each
block timed with HIP events over REP launches; optional CPU comparison with the reference binaries in oracle/_ref.
usage: python tools/gen_bench.py [npackets=200000] [--cpu]"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leansdr_amd import capi
lib, vp, c_sz = capi.lib, capi.vp, capi.c_sz
hip = C.CDLL("libamdhip64.so")
NP = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 200000
REP = 5
ctx = capi.Ctx(0)
stream = C.c_void_p(lib.lsdr_ctx_stream(ctx.h))


def ev():
    e = C.c_void_p()
    assert hip.hipEventCreate(C.byref(e)) == 0
    return e


def timed(fn):
    fn()                                     # warm-up
    ctx.sync()
    e0, e1 = ev(), ev()
    hip.hipEventRecord(e0, stream)
    for _ in range(REP):
        fn()
    hip.hipEventRecord(e1, stream)
    hip.hipEventSynchronize(e1)
    ms = C.c_float()
    hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    return ms.value / REP


def report(name, ms, items, unit, bytes_moved):
    print(f"{name:34s} {ms:8.3f} ms  {items / ms / 1e3:10.1f} M{unit}/s  {bytes_moved / ms / 1e6:8.1f} GB/s algorithmic")


rng = np.random.default_rng(1)
ts = rng.integers(0, 256, (NP, 188), dtype=np.uint8)
ts[:, 0] = 0x47
cons, prod = c_sz(), c_sz()
d_ts = ctx.upload(ts)
d_r = ctx.alloc(NP * 188)
h = vp(); capi.check(lib.lsdr_randomizer_create(ctx.h, C.byref(h)))
report("randomizer", timed(lambda: capi.check(lib.lsdr_randomizer_run(h, d_ts.ptr, NP, d_r.ptr, NP, C.byref(cons), C.byref(prod)))), NP, "packet", NP * 376)
d_pk = ctx.alloc(NP * 204)
report("rs_encoder", timed(lambda: capi.check(lib.lsdr_rs_encoder_run(ctx.h, d_r.ptr, NP, d_pk.ptr, NP, C.byref(cons), C.byref(prod)))), NP, "packet", NP * 392)
d_il = ctx.alloc(NP * 204)
report("interleaver", timed(lambda: capi.check(lib.lsdr_interleaver_run(ctx.h, d_pk.ptr, NP, d_il.ptr, NP * 204, C.byref(cons), C.byref(prod)))), NP, "packet", NP * 408)
nby = prod.value
cv = vp(); capi.check(lib.lsdr_convol_create(ctx.h, capi.FEC12, 2, C.byref(cv)))
d_sym = ctx.alloc(nby * 8 + 64)
report("dvb_convol 1/2 QPSK", timed(lambda: capi.check(lib.lsdr_convol_run(cv, d_il.ptr, nby, d_sym.ptr, nby * 8 + 64, C.byref(cons), C.byref(prod)))), nby * 4, "sym", nby * 5)
nsym = prod.value
d_iq = ctx.alloc(nsym * 8)
report("cstln_transmitter", timed(lambda: capi.check(lib.lsdr_cstln_transmitter_run(ctx.h, capi.QPSK, capi.FEC12, d_sym.ptr, nsym, d_iq.ptr))), nsym, "sym", nsym * 9)
co = capi.normalize_power(capi.root_raised_cosine(20, 0.5, 0.35), 1.0 / 75.0)
rs = vp(); capi.check(lib.lsdr_fir_resampler_create(ctx.h, len(co), capi._np(co), 2, C.byref(rs)))
d_y = ctx.alloc(nsym * 2 * 8)
report("fir_resampler x2 (41 taps)", timed(lambda: capi.check(lib.lsdr_fir_resampler_run(rs, d_iq.ptr, nsym, d_y.ptr, nsym * 2, C.byref(cons), C.byref(prod)))), nsym * 2, "S", nsym * 24)
ny = prod.value
ag = vp(); capi.check(lib.lsdr_simple_agc_create(ctx.h, 1.0, 0.0005, C.byref(ag)))
d_z = ctx.alloc(ny * 8)
report("simple_agc", timed(lambda: capi.check(lib.lsdr_simple_agc_run(ag, d_y.ptr, ny, d_z.ptr, ny, C.byref(cons), C.byref(prod)))), ny, "S", ny * 16)
w = vp(); capi.check(lib.lsdr_wgn_create(ctx.h, 0, 0, C.byref(w)))
d_n = ctx.alloc(ny * 8)
report("wgn_c (+adder fused)", timed(lambda: capi.check(lib.lsdr_wgn_run(w, 0.1, d_z.ptr, d_n.ptr, ny))), ny, "S", ny * 16)
dr = vp(); capi.check(lib.lsdr_drifter_create(ctx.h, C.byref(dr)))
d_d = ctx.alloc(ny * 8)
report("drifter (pass-through, amp 0)", timed(lambda: capi.check(lib.lsdr_drifter_run(dr, d_n.ptr, ny, d_d.ptr, 4096))), ny, "S", ny * 16)
capi.check(lib.lsdr_drifter_set_component(dr, 0, 2e-6, 1e-7))
report("drifter (one component, 4096/run)", timed(lambda: capi.check(lib.lsdr_drifter_run(dr, d_n.ptr, ny, d_d.ptr, 4096))), ny, "S", ny * 18)
d_u8 = ctx.alloc(ny * 2)
report("cconverter f32->u8", timed(lambda: capi.check(lib.lsdr_cconverter_f32_u8_run(ctx.h, d_d.ptr, ny, d_u8.ptr))), ny, "S", ny * 10)
print(f"({NP} TS packets -> {ny} baseband samples)")

if "--cpu" in sys.argv:
    ref = os.path.join(ROOT, "oracle", "_ref")
    n = min(NP, 20000)
    open("/tmp/gen_ts.bin", "wb").write(ts[:n].tobytes())
    t0 = time.perf_counter()
    subprocess.run(f"{ref}/leandvbtx -f 2 --agc < /tmp/gen_ts.bin > /tmp/gen_iq.bin", shell=True, check=True)
    t1 = time.perf_counter()
    subprocess.run(f"{ref}/leanchansim --awgn -20 --deterministic < /tmp/gen_iq.bin > /dev/null", shell=True, check=True)
    t2 = time.perf_counter()
    ns = os.path.getsize("/tmp/gen_iq.bin") // 8
    print(f"reference leandvbtx  (1 core): {n} packets -> {ns} samples in {t1 - t0:.2f} s = {ns / (t1 - t0) / 1e6:.1f} MS/s")
    print(f"reference leanchansim (1 core): {ns} samples in {t2 - t1:.2f} s = {ns / (t2 - t1) / 1e6:.1f} MS/s")

"""Empty and tiny inputs through the C ABI of the generator-side blocks: every *_run accepts n = 0 (and inputs shorter than its
minimum) as "no progress", like the reference blocks whose run() finds nothing readable."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_inputs(capi, ctx):
    lib, vp, c_sz = capi.lib, capi.vp, capi.c_sz
    cons, prod = c_sz(7), c_sz(7)
    d = ctx.alloc(4096)
    h = vp(); capi.check(lib.lsdr_randomizer_create(ctx.h, C.byref(h)))
    capi.check(lib.lsdr_randomizer_run(h, d.ptr, 0, d.ptr, 0, C.byref(cons), C.byref(prod)))
    assert (cons.value, prod.value) == (0, 0)
    lib.lsdr_randomizer_destroy(h)
    for fn in (lib.lsdr_rs_encoder_run, lib.lsdr_interleaver_run):
        cons.value = prod.value = 7
        capi.check(fn(ctx.h, d.ptr, 0, d.ptr, 100, C.byref(cons), C.byref(prod)))
        assert (cons.value, prod.value) == (0, 0)
    capi.check(lib.lsdr_interleaver_run(ctx.h, d.ptr, 11, d.ptr, 4096, C.byref(cons), C.byref(prod)))   # needs 12 packets
    assert (cons.value, prod.value) == (0, 0)
    cv = vp(); capi.check(lib.lsdr_convol_create(ctx.h, capi.FEC78, 2, C.byref(cv)))
    capi.check(lib.lsdr_convol_run(cv, d.ptr, 6, d.ptr, 4096, C.byref(cons), C.byref(prod)))             # 7/8 needs 7 bytes
    assert (cons.value, prod.value) == (0, 0)
    lib.lsdr_convol_destroy(cv)
    capi.check(lib.lsdr_cstln_transmitter_run(ctx.h, capi.QPSK, capi.FEC12, d.ptr, 0, d.ptr))
    ag = vp(); capi.check(lib.lsdr_simple_agc_create(ctx.h, 1.0, 0.001, C.byref(ag)))
    capi.check(lib.lsdr_simple_agc_run(ag, d.ptr, 127, d.ptr, 512, C.byref(cons), C.byref(prod)))       # chunks of 128
    assert (cons.value, prod.value) == (0, 0)
    lib.lsdr_simple_agc_destroy(ag)
    w = capi.Wgn(ctx)
    s0 = w.state
    assert len(w.run(0)) == 0 and w.state == s0
    assert len(w.run(1)) == 1
    w.close()
    dr = capi.Drifter(ctx, (0.01, 0, 0), (1e-6, 0, 0))
    assert len(dr.run(np.zeros(0, np.complex64))) == 0 and dr.phases == (0, 0, 0)
    dr.close()
    assert len(capi.adder(ctx, np.zeros(0, np.complex64), np.zeros(0, np.complex64))) == 0
    assert capi.cconv_f32_u8(ctx, np.zeros(0, np.complex64)).shape == (0, 2)
    assert capi.cconv_f32_s16(ctx, np.zeros(0, np.complex64)).shape == (0, 2)
    d.free()


def test_tx_chain_fewer_packets_than_the_interleaver_window(capi, ctx, oracle):
    """11 packets produce nothing (the interleaver looks 11 packets ahead); the 12th releases the first 204 bytes."""
    from leansdr_amd import synth_dvbs
    ts = synth_dvbs.ts_packets(13)
    tx = capi.TxChain(ctx, interp=2)
    assert len(tx.run(ts[:11])) == 0
    y = tx.run(ts[11:])
    tx.close()
    want = oracle.tx_chain(ts, interp=2)
    assert len(y) == len(want) and y.tobytes() == want.tobytes()

"""cstln_receiver with the `leandvb --u8` input stage fused into its loads (lsdr_rx_cfg.in_format = LSDR_IN_CU8;
leandvb.cc:211-217: cconverter<u8,128,f32,0,1,1> → cstln_receiver): the receiver reads cu8 items and converts them on
load exactly as dsp.h:40-50 does, so the cf32 stream never exists in HBM.  Every mode must give the bits of the unfused
pair cconverter → cstln_receiver: serial (vs the reference's golden vector and the oracle), lane-per-capture batch, tiled."""
import numpy as np
import pytest
from conftest import gold, bits_equal
from test_oracle_golden import check_rx_against_golden, state_vec
import pyoracle as po
from leansdr_amd import synth_dvbs

pytestmark = pytest.mark.gpu
OMEGA = float(np.float32(2400e3 / 2000e3))


@pytest.fixture(scope="module")
def capture():
    iq, _ = synth_dvbs.capture_u8(n_packets=400, sps_num=6, sps_den=5, seed=21)
    return iq[: len(iq) // 2 * 2]


def test_serial_cu8_reference_golden(capi, ctx):
    """tests/golden/cstln_receiver.npz lin1p2_u8_*: what the reference's cconverter<u8> → cstln_receiver wrote for `u8`."""
    g = gold("cstln_receiver.npz")
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=OMEGA, meas_decimation=2400, in_format=capi.IN_CU8)
    out = r.run(g["u8"])
    check_rx_against_golden(out, g, "lin1p2_u8")
    r.close()


@pytest.mark.parametrize("sampler", [0, 1, 2])
def test_serial_cu8_equals_oracle(capi, ctx, oracle, capture, sampler):
    g = gold("cstln_receiver.npz")
    kw = dict(sampler=sampler, cstln=1, omega=OMEGA, meas_decimation=4096)
    if sampler == 2:
        kw.update(coeffs=g["rrc_rx"], subsampling=16)
    # a pipebuf<cu8> read pointer is only 2-byte aligned in general: the stream starts at sample 1 of the allocation
    stream = capture[2:]
    ref = oracle.rx(po.rx_params(**kw), oracle.cconverter_u8(stream))
    r = capi.CstlnReceiver(ctx, in_format=capi.IN_CU8, **kw)
    n = len(stream) // 2
    d, o = ctx.upload(capture), ctx.alloc((n + 256) * 4)
    got, pos = [], 0
    for piece in (12929, 1 << 30):          # two calls: the loop state is carried in the handle
        res = r.run_dev(d.at(2 + 2 * pos), min(piece, n - pos), o.ptr, n + 256, meas=False)
        got.append(ctx.download(o, capi.SOFTSYM, res["produced"]).copy())
        pos += res["consumed"]
    got = np.concatenate(got)
    assert pos == ref["consumed"]
    assert bits_equal(got["cost"], ref["sym"]["cost"]) and bits_equal(got["symbol"], ref["sym"]["symbol"])
    sv, _ = state_vec(r.state())
    wv, _ = state_vec(ref["state"])
    assert bits_equal(sv, wv)
    r.close(); d.free(); o.free()


def test_batch_cu8_lane_per_capture(capi, ctx, oracle):
    """lsdr_rx_batch on cu8 captures (BASELINE config 4's unit: a 2.4 MS/s u8 capture per lane), bit-exact per capture."""
    n_streams, n = 70, 128 * 60 + 1
    caps = [synth_dvbs.capture_u8(n_packets=8, sps_num=6, sps_den=5, seed=300 + i, amp=60.0 + i % 9)[0][: 2 * n] for i in range(n_streams)]
    p = po.rx_params(sampler=1, cstln=1, omega=OMEGA, meas_decimation=4096)
    b = capi.RxBatch(ctx, n_streams, sampler=1, cstln=capi.QPSK, omega=OMEGA, meas_decimation=4096, in_format=capi.IN_CU8)
    d_in = [ctx.upload(c) for c in caps]
    d_out = [ctx.alloc(n * 4) for _ in range(n_streams)]
    cons, prod = b.run_dev([d.ptr for d in d_in], n, [d.ptr for d in d_out], n)
    assert cons == (n - 1) // 128 * 128
    for i in range(n_streams):
        ref = oracle.rx(p, oracle.cconverter_u8(caps[i]))
        g = ctx.download(d_out[i], capi.SOFTSYM, prod[i])
        assert ref["consumed"] == cons and bits_equal(g["cost"], ref["sym"]["cost"]) and bits_equal(g["symbol"], ref["sym"]["symbol"]), i
        st = b.state(i)
        for k in ("mu", "phase", "freqw", "agc_gain", "est_insp"):
            assert np.float32(getattr(st, k)).tobytes() == np.float32(getattr(ref["state"], k)).tobytes(), (i, k)
    b.close()
    for d in d_in + d_out:
        d.free()


@pytest.mark.parametrize("sampler", [0, 1, 2])
@pytest.mark.parametrize("tile_len,warm", [(1024, 512), (4096, 512), (256, 256)])
def test_tiled_cu8_is_the_tiled_cf32_receiver(capi, ctx, oracle, capture, sampler, tile_len, warm):
    """The tolerance tiles read the same sample values whichever way they arrive (one unaligned 8-byte window load of four
    cu8 samples instead of three cf32 loads): tiled(cu8) is bit for bit tiled(cconverter(cu8)), state and reports included."""
    if sampler == 0 and tile_len != 1024:
        pytest.skip("nearest sampler: one geometry is enough (it cannot lock at 1.2 samples/symbol)")
    g = gold("cstln_receiver.npz")
    kw = dict(sampler=sampler, cstln=1, omega=OMEGA, meas_decimation=4096, mode=capi.RX_TILED, tile_len=tile_len, tile_warmup=warm)
    if sampler == 2:
        kw.update(coeffs=g["rrc_rx"], subsampling=16)
    x = oracle.cconverter_u8(capture)
    a = capi.CstlnReceiver(ctx, **kw)
    b = capi.CstlnReceiver(ctx, in_format=capi.IN_CU8, **kw)
    oa, ob = a.run(x), b.run(capture)
    assert oa["consumed"] == ob["consumed"] > 100000 and a.tiled_stats() == b.tiled_stats()
    assert bits_equal(oa["sym"]["cost"], ob["sym"]["cost"]) and bits_equal(oa["sym"]["symbol"], ob["sym"]["symbol"])
    assert bits_equal(oa["ss"], ob["ss"]) and bits_equal(oa["mer"], ob["mer"]) and bits_equal(oa["freq"], ob["freq"])
    assert oa["state"].as_dict() == ob["state"].as_dict()
    a.close(); b.close()


def test_tiled_cu8_tracks_the_serial_receiver(capi, ctx, oracle, capture):
    """C1 geometry (1.2 samples/symbol, linear sampler), tiles of 4096 samples after 512 of warm-up, from the serial loop's
    state after acquisition: same symbol count, decisions and costs within the tiled mode's tolerance."""
    from leansdr_amd.tolerance import TOL, check_tiled
    p = po.rx_params(sampler=1, cstln=1, omega=OMEGA, meas_decimation=4096)
    x = oracle.cconverter_u8(capture)
    acq = 128 * 512
    a = oracle.rx(p, x[: acq + 1])
    ref = oracle.rx(p, x[acq:], state_in=a["state"])
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=OMEGA, meas_decimation=4096, mode=capi.RX_TILED, tile_len=4096,
                           tile_warmup=512, in_format=capi.IN_CU8)
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    out = r.run(capture[2 * acq:])
    stats = r.tiled_stats()
    r.close()
    assert out["consumed"] == ref["consumed"]
    rep = check_tiled(out["sym"], ref["sym"], stats, first_exact=512 // 2)
    assert rep["pass"], (rep, TOL)

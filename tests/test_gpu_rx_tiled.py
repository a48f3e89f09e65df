"""LSDR_RX_TILED (throughput mode of cstln_receiver) against the exact serial oracle.

Tolerance (stated): every tile but the first re-acquires
timing and carrier phase during its warm-up, so the loop state differs from the serial
trajectory by loop noise.  On a locked QPSK stream (Es/N0 = 20 dB in the synth's
definition, the bench condition) we require
  * exactly the same number of soft symbols for the same consumed input,
  * >= 99.9 % identical symbol decisions,
  * mean |Δcost| <= 3 % of the constellation's largest |cost| (11236 for QPSK),
and the first tile (which continues from the carried state) must be bit-exact.
"""
import numpy as np
import pytest
from conftest import bits_equal
import pyoracle as po
from leansdr_amd import synth

pytestmark = pytest.mark.gpu

SS_RTOL = 0.02        # signal-strength report and carried AGC state vs the serial receiver
MER_ATOL_DB = 1.0     # MER report vs the serial receiver (the tiles' residual timing/AGC settling shows at ≈ 20 dB MER)


@pytest.fixture(scope="module")
def stream():
    x, _ = synth.qpsk_baseband(4 * 100000, 4, seed=5, rms=50.0, snr_db=20.0)
    return x


@pytest.mark.parametrize("tile_len,warm", [(512, 1024), (1024, 1024), (256, 1024), (128, 512), (256, 256), (128, 256), (0, 0)])
def test_tiled_vs_serial(capi, ctx, oracle, stream, tile_len, warm):
    p = po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=4096)
    acq = 40960
    a = oracle.rx(p, stream[:acq + 1])                      # serial acquisition (exact)
    assert a["consumed"] == acq
    ref = oracle.rx(p, stream[acq:], state_in=a["state"])   # serial continuation = truth

    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0, meas_decimation=4096,
                           mode=capi.RX_TILED, tile_len=tile_len, tile_warmup=warm)
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    out = r.run(stream[acq:])
    stats = r.tiled_stats()
    r.close()
    assert out["consumed"] == ref["consumed"]
    assert len(out["sym"]) == len(ref["sym"]), (len(out["sym"]), len(ref["sym"]), stats)
    same = (out["sym"]["symbol"] == ref["sym"]["symbol"]).mean()
    dcost = np.abs(out["sym"]["cost"].astype(int) - ref["sym"]["cost"].astype(int))
    assert same >= 0.999, (same, stats)
    assert dcost.mean() <= 0.03 * 11236, (dcost.mean(), stats)
    assert stats["bad_seams"] == 0 and stats["tiles"] > 10
    # first tile: exact continuation of the carried state
    n0 = (warm or 256) // 4 - 8      # tile 0 is one warm-up long; (0, 0) = library defaults: 256-sample warm-up at omega 4
    assert bits_equal(out["sym"]["cost"][:n0], ref["sym"]["cost"][:n0])
    # measurement stream has the reference's cadence
    assert len(out["freq"]) == len(ref["freq"]) and len(ref["freq"]) > 50
    # the estimators behind SS and MER (sdr.h:905-913) are EMAs over ALL chunks: the tiles' per-chunk contributions are
    # scanned (k_rx_ema), so the reports follow the serial receiver's, not a per-tile restart
    assert np.allclose(out["ss"], ref["ss"], rtol=SS_RTOL), np.max(np.abs(out["ss"] / ref["ss"] - 1))
    assert np.max(np.abs(out["mer"] - ref["mer"])) <= MER_ATOL_DB, np.max(np.abs(out["mer"] - ref["mer"]))
    s0, s1 = out["state"], ref["state"]
    assert abs(s0.est_insp / s1.est_insp - 1) <= SS_RTOL and abs(s0.agc_gain / s1.agc_gain - 1) <= SS_RTOL
    # --fd-const: one sampled constellation point per chunk (sdr.h:861-864), also in the tiled mode
    assert len(out["cstln"]) == len(ref["cstln"]) > 100
    assert np.mean(np.abs(out["cstln"] - ref["cstln"])) < 0.05 * 75


def test_tiled_short_input_is_exact(capi, ctx, oracle, stream):
    """Fewer chunks than one tile: the tiled mode degenerates to the exact serial loop."""
    p = po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=4096)
    ref = oracle.rx(p, stream[:1000])
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0, meas_decimation=4096, mode=capi.RX_TILED, tile_len=1024, tile_warmup=512)
    out = r.run(stream[:1000])
    r.close()
    assert out["consumed"] == ref["consumed"] == 896
    assert bits_equal(out["sym"]["cost"], ref["sym"]["cost"]) and bits_equal(out["sym"]["symbol"], ref["sym"]["symbol"])


def test_tiled_state_carry_across_runs(capi, ctx, oracle, stream):
    """Two consecutive tiled runs == one stream: counts add up and labels stay in one frame."""
    p = po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=4096)
    ref = oracle.rx(p, stream)
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0, meas_decimation=4096, mode=capi.RX_TILED,
                           tile_len=512, tile_warmup=1024)
    o1 = r.run(stream[:200001], meas=False)
    o2 = r.run(stream[o1["consumed"]:], meas=False)
    r.close()
    got = np.concatenate([o1["sym"], o2["sym"]])
    assert o1["consumed"] + o2["consumed"] == ref["consumed"]
    assert len(got) == len(ref["sym"])
    tail = slice(20000, None)      # after acquisition
    assert (got["symbol"][tail] == ref["sym"]["symbol"][tail]).mean() >= 0.999


def test_queued_runs_equal_synchronous_runs(capi, ctx, oracle, stream):
    """lsdr_rx_run_async / lsdr_rx_wait: three runs queued back to back (loop state carried on the device, quadrant
    fix-up applied by the seam kernel) give the same symbols and final state as three synchronous runs."""
    x = stream
    acq = capi.CstlnReceiver(ctx, sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=4.0)
    acq.run(x[:32768], meas=False)
    st = acq.state()
    parts = [x[32768:32768 + 40000], x[72768:72768 + 40000], x[112768:112768 + 40000]]
    kw = dict(sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=4.0, mode=capi.RX_TILED, tile_len=256, tile_warmup=512)
    a = capi.CstlnReceiver(ctx, **kw); a.set_state(st)
    b = capi.CstlnReceiver(ctx, **kw); b.set_state(st)
    want = []
    pos = 0
    xs = np.concatenate(parts)
    d = ctx.upload(xs)
    o = ctx.alloc(len(xs) * 4)
    # synchronous
    for _ in range(3):
        r = a.run_dev(d.at(pos * 8), min(40001, len(xs) - pos), o.ptr, len(xs), meas=False)
        want.append(ctx.download(o, capi.SOFTSYM, r["produced"]).copy())
        pos += r["consumed"]
    # queued
    outs = [ctx.alloc(len(xs) * 4) for _ in range(3)]
    pos2 = 0
    for k in range(3):
        pos2 += b.run_async(d.at(pos2 * 8), min(40001, len(xs) - pos2), outs[k].ptr, len(xs))
    got = [ctx.download(outs[k], capi.SOFTSYM, b.wait()).copy() for k in range(3)]
    assert pos2 == pos
    for k in range(3):
        assert len(got[k]) > 9000 and bits_equal(got[k], want[k]), k
    sa, sb = a.state(), b.state()
    assert sa.as_dict() == sb.as_dict()
    with pytest.raises(capi.LsdrError):
        b.wait()                      # nothing queued


def test_bench_geometry_on_the_c2_chain(capi, ctx, oracle):
    """The configuration bench.py times: Fs 240 MS/s cf32 at 120 samples/symbol → scaler(×75) + fir_filter(313, /30) on
    the GPU (bit-exact vs the oracle) → tiled receiver, tiles of 128 samples after a 256-sample warm-up, against the
    oracle's fir_filter → exact serial receiver from the same acquisition state."""
    import bench
    coeffs, decim = bench.c2_filter(capi)
    x, _ = synth.qpsk_baseband(120 * 65536, 120, seed=11, rms=1.0, snr_db=20.0)
    y_ref, _ = oracle.fir_filter(coeffs, decim, oracle.scaler(75.0, x))
    fir = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0)
    y, _ = fir.run(x)
    fir.close()
    assert bits_equal(y, y_ref)
    omega = 240e6 / decim / 2e6
    p = po.rx_params(sampler=1, cstln=1, omega=omega, meas_decimation=8192)
    acq = 32768
    a = oracle.rx(p, y_ref[:acq + 1])
    ref = oracle.rx(p, y_ref[acq:], state_in=a["state"])
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=omega, meas_decimation=8192, mode=capi.RX_TILED,
                           tile_len=128, tile_warmup=256)
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    out = r.run(y[acq:])
    stats = r.tiled_stats()
    r.close()
    assert out["consumed"] == ref["consumed"] and len(out["sym"]) == len(ref["sym"]) > 50000, stats
    same = (out["sym"]["symbol"] == ref["sym"]["symbol"]).mean()
    dcost = np.abs(out["sym"]["cost"].astype(int) - ref["sym"]["cost"].astype(int)).mean()
    assert same >= bench.TOL["min_equal_decisions"] and dcost <= bench.TOL["max_mean_abs_dcost"], (same, dcost, stats)
    assert stats["bad_seams"] == 0
    assert bits_equal(out["sym"]["cost"][:56], ref["sym"]["cost"][:56])
    assert np.allclose(out["ss"], ref["ss"], rtol=SS_RTOL) and np.max(np.abs(out["mer"] - ref["mer"])) <= MER_ATOL_DB


def test_tiled_fir_sampler_vs_serial(capi, ctx, oracle):
    """--sampler rrc (fir_sampler: polyphase matched filter, sdr.h:635-689) in the tiled mode: the tolerance tiles take
    their shifted taps from a table rebuilt from the carried frequency at the start of every run."""
    from conftest import gold
    g = gold("cstln_receiver.npz")
    rrc = g["rrc_rx"]
    x, _ = synth.qpsk_baseband(4 * 200000, 4, seed=7, rms=50.0, snr_db=20.0)
    p = po.rx_params(sampler=2, coeffs=rrc, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096)
    # The matched filter's gain is 1/S: est_insp has to come down from 75² to ≈ 8 at 1 % per chunk — the serial
    # loop needs ≈ 2000 chunks before its AGC stands still, and a tiled run keeps the AGC of its first chunk for all tiles.
    acq = 128 * 2400
    a = oracle.rx(p, x[:acq + len(rrc) - 1])
    assert a["consumed"] == acq
    ref = oracle.rx(p, x[acq:], state_in=a["state"])
    r = capi.CstlnReceiver(ctx, sampler=2, coeffs=rrc, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096,
                           mode=capi.RX_TILED, tile_len=256, tile_warmup=1024)
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    out = r.run(x[acq:])
    stats = r.tiled_stats()
    r.close()
    assert out["consumed"] == ref["consumed"] and len(out["sym"]) == len(ref["sym"]), (len(out["sym"]), len(ref["sym"]), stats)
    same = (out["sym"]["symbol"] == ref["sym"]["symbol"]).mean()
    dcost = np.abs(out["sym"]["cost"].astype(int) - ref["sym"]["cost"].astype(int)).mean()
    assert same >= 0.999 and dcost <= 0.05 * 11236 and stats["bad_seams"] == 0, (same, dcost, stats)

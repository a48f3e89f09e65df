"""LSDR_RX_TILED (throughput mode of cstln_receiver) against the exact serial oracle.

The tolerance is stated ONCE, in leansdr_amd/tolerance.py (TOL at the bench condition, LOW_SNR at 10-12 dB): same symbol count,
first tile bit-exact, >= 99.9 % identical decisions, mean / 99th percentile / maximum of |Δcost| bounded, no unreconciled seam,
reports within ss_rtol / mer_atol_db.  bench.py's `verified` objects use the same module.
"""
import numpy as np
import pytest
from conftest import bits_equal
import pyoracle as po
from leansdr_amd import synth
from leansdr_amd.tolerance import TOL, LOW_SNR, check_tiled

pytestmark = pytest.mark.gpu

SS_RTOL = TOL["ss_rtol"]          # signal-strength report and carried AGC state vs the serial receiver
MER_ATOL_DB = TOL["mer_atol_db"]  # MER report vs the serial receiver (the tiles' residual timing/AGC settling shows at ≈ 20 dB MER)


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
    # first tile: exact continuation of the carried state (tile 0 is one warm-up long; (0, 0) = library defaults: 256 samples)
    rep = check_tiled(out["sym"], ref["sym"], stats, first_exact=(warm or 256) // 4 - 8)
    assert rep["pass"] and stats["tiles"] > 10, (rep, TOL)
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


def _c2_chain(capi, ctx, oracle, snr_db, n_sym=65536, seed=11):
    """Fs 240 MS/s cf32 at 120 samples/symbol → scaler(×75) + fir_filter(313, /30) on the GPU (bit-exact vs the oracle)."""
    import bench
    coeffs, decim = bench.c2_filter(capi)
    x, _ = synth.qpsk_baseband(120 * n_sym, 120, seed=seed, rms=1.0, snr_db=snr_db)
    y_ref, _ = oracle.fir_filter(coeffs, decim, oracle.scaler(75.0, x))
    fir = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0)
    y, _ = fir.run(x)
    fir.close()
    assert bits_equal(y, y_ref)
    return y, 240e6 / decim / 2e6


def _tiled_vs_serial(capi, ctx, oracle, y, omega, acq, tile):
    p = po.rx_params(sampler=1, cstln=1, omega=omega, meas_decimation=8192)
    a = oracle.rx(p, y[:acq + 1])
    ref = oracle.rx(p, y[acq:], state_in=a["state"])
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=omega, meas_decimation=8192, mode=capi.RX_TILED, tile_len=tile[0], tile_warmup=tile[1])
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    out = r.run(y[acq:])
    stats = r.tiled_stats()
    r.close()
    assert out["consumed"] == ref["consumed"]
    return out, ref, stats


def test_bench_geometry_on_the_c2_chain(capi, ctx, oracle):
    """The configuration bench.py times (its DEFAULT_TILE, imported): the C2 chain → tiled receiver against the oracle's
    fir_filter → exact serial receiver from the same acquisition state, under TOL."""
    import bench
    # (the serial loop acquires for 1000 chunks, like bench.py's pipelines do before their tiled receivers take over: the AGC
    # estimator has a 100-chunk time constant and a tiled run keeps the gain it starts with — after 256 chunks the mean |Δcost|
    # was 395 instead of ≈ 200)
    y, omega = _c2_chain(capi, ctx, oracle, 20.0, n_sym=98304)
    out, ref, stats = _tiled_vs_serial(capi, ctx, oracle, y, omega, 128 * 1000, bench.DEFAULT_TILE)
    rep = check_tiled(out["sym"], ref["sym"], stats, first_exact=bench.DEFAULT_TILE[1] // 4 - 8)
    assert rep["pass"] and len(out["sym"]) > 50000, (rep, TOL)
    assert np.allclose(out["ss"], ref["ss"], rtol=SS_RTOL) and np.max(np.abs(out["mer"] - ref["mer"])) <= MER_ATOL_DB


@pytest.mark.parametrize("snr_db", [12.0, 10.0])
def test_c2_chain_at_low_snr(capi, ctx, oracle, snr_db):
    """10-12 dB on the C2 chain, bench geometry, under LOW_SNR.  The serial loop is given 1000 chunks to acquire: the AGC
    estimator has a 100-chunk time constant (sdr.h:867-870) and a tiled run keeps the gain it starts with, so a run that begins
    before the AGC has settled carries that error through all of its tiles (2.5 time constants gave -1.5 dB of MER here)."""
    import bench
    y, omega = _c2_chain(capi, ctx, oracle, snr_db, n_sym=98304)
    out, ref, stats = _tiled_vs_serial(capi, ctx, oracle, y, omega, 128 * 1000, bench.DEFAULT_TILE)
    rep = check_tiled(out["sym"], ref["sym"], stats, first_exact=bench.DEFAULT_TILE[1] // 4 - 8, tol=LOW_SNR)
    assert rep["pass"], (rep, LOW_SNR)
    assert np.allclose(out["ss"], ref["ss"], rtol=LOW_SNR["ss_rtol"]), np.max(np.abs(out["ss"] / ref["ss"] - 1))
    assert np.max(np.abs(out["mer"] - ref["mer"])) <= LOW_SNR["mer_atol_db"], (out["mer"] - ref["mer"])


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
    rep = check_tiled(out["sym"], ref["sym"], stats)
    assert rep["pass"], (rep, TOL)


@pytest.mark.parametrize("steps,tile,odd", [(16, (384, 256), False), (16, (384, 256), True), (128, (256, 512), False)])
def test_tiled_fir_sampler_taps_in_lds_or_hbm_same_as_serial(capi, ctx, oracle, steps, tile, odd):
    """The RRC tiles read their taps out of LDS (≤ 1024 taps) or, for longer filters, out of the table in HBM, and cf32 samples in pairs — at the bench's
    tile geometry, from an input that starts on an odd sample (8-byte-aligned 16-byte loads), and with a 128-steps-per-sample filter (1329 taps: the HBM path):
    the sequential loop's decisions under TOL."""
    order = int(10 * 8e6 * steps / (22 * 1e6 * 0.35))
    rrc = capi.root_raised_cosine(order, float(np.float32(2e6) / np.float32(8e6 * steps)), 0.35)
    assert (len(rrc) > 1024) == (steps == 128)
    x, _ = synth.qpsk_baseband(4 * 150000, 4, seed=17, rms=50.0, snr_db=20.0)
    p = po.rx_params(sampler=2, coeffs=rrc, subsampling=steps, cstln=1, omega=4.0, meas_decimation=4096)
    acq = 128 * 2400
    a = oracle.rx(p, x[:acq + len(rrc) - 1])
    ref = oracle.rx(p, x[acq:], state_in=a["state"])
    r = capi.CstlnReceiver(ctx, sampler=2, coeffs=rrc, subsampling=steps, cstln=1, omega=4.0, meas_decimation=4096,
                           mode=capi.RX_TILED, tile_len=tile[0], tile_warmup=tile[1])
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    tail = x[acq:]
    d_in = ctx.alloc((len(tail) + 1) * 8)
    off = 1 if odd else 0
    capi.check(capi.lib.lsdr_memcpy_h2d(ctx.h, d_in.at(8 * off), np.ascontiguousarray(tail).ctypes.data_as(capi.vp), tail.nbytes))
    ctx.sync()
    d_out = ctx.alloc((len(tail) // 4 + 8192) * 4)
    o = r.run_dev(d_in.at(8 * off), len(tail), d_out.ptr, len(tail) // 4 + 8192, meas=False)
    sym = ctx.download(d_out, ref["sym"].dtype, o["produced"])
    stats = r.tiled_stats()
    r.close(); d_in.free(); d_out.free()
    assert o["consumed"] == ref["consumed"] and len(sym) == len(ref["sym"]), (len(sym), len(ref["sym"]), stats)
    rep = check_tiled(sym, ref["sym"], stats)
    assert rep["pass"], (rep, TOL)


@pytest.mark.parametrize("fmt", ["cf32", "cu8"])
def test_multi_capture_runs_equal_separate_queued_runs(capi, ctx, oracle, fmt):
    """lsdr_rx_run_multi_async: three independent captures (own signal, own loop state) share their launches — same symbols,
    counts, seam statistics and final loop state as three separate lsdr_rx_run_async calls, bit for bit, over two queued runs;
    receivers that cannot share (different tile geometry) fall back to separate launches with the same results."""
    n = 60000
    xs = []
    for k in range(3):
        x, _ = synth.qpsk_baseband(4 * (n // 4 + 9000), 4, seed=20 + k, rms=50.0, snr_db=18.0 + k)
        xs.append(x)
    kw = dict(sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=4.0, mode=capi.RX_TILED, tile_len=256, tile_warmup=256)
    if fmt == "cu8":
        kw["in_format"] = capi.IN_CU8
        xs = [np.clip(np.round(np.stack([x.real, x.imag], -1) * 0.5) + 128, 0, 255).astype(np.uint8) for x in xs]
        isz, acq_kw = 2, dict(in_format=capi.IN_CU8)
    else:
        isz, acq_kw = 8, {}
    sts = []
    for x in xs:
        acq = capi.CstlnReceiver(ctx, sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=4.0, **acq_kw)
        acq.run(x[:32768], meas=False)
        sts.append(acq.state()); acq.close()
    dins = [ctx.upload(x[32768:]) for x in xs]
    m = len(xs[0]) - 32768
    half = 16000

    def run(multi, kws):
        rxs = [capi.CstlnReceiver(ctx, **k) for k in kws]
        for r, st in zip(rxs, sts):
            r.set_state(st)
        outs = [[ctx.alloc(m * 4 + 1024) for _ in range(2)] for _ in rxs]
        pos = 0
        for q in range(2):
            if multi:
                used = capi.CstlnReceiver.run_multi_async(rxs, [d.at(pos * isz) for d in dins], half + 1, [o[q].ptr for o in outs], m)
                assert len(used) == len(rxs) and len(set(used)) == 1       # one entry per capture; alike receivers take the same
                used = used[0]
            else:
                for r, d, o in zip(rxs, dins, outs):
                    used = r.run_async(d.at(pos * isz), half + 1, o[q].ptr, m)
            pos += used
        res = []
        for r, o in zip(rxs, outs):
            syms = [ctx.download(o[q], capi.SOFTSYM, r.wait()).copy() for q in range(2)]
            res.append((syms, r.tiled_stats(), r.state().as_dict()))
            r.close()
        for o in outs:
            for b in o:
                b.free()
        return pos, res

    pos_a, sep = run(False, [kw] * 3)
    pos_b, mul = run(True, [kw] * 3)
    assert pos_a == pos_b and pos_a > 0
    for k in range(3):
        for q in range(2):
            assert len(mul[k][0][q]) > 3000 and bits_equal(mul[k][0][q], sep[k][0][q]), (k, q)
        assert mul[k][1] == sep[k][1] and mul[k][2] == sep[k][2], k
    assert not bits_equal(mul[0][0][0], mul[1][0][0][:len(mul[0][0][0])]) or len(mul[0][0][0]) != len(mul[1][0][0])
    # not alike: one receiver has another tile length → queued one by one, same results as on its own
    kws = [kw, dict(kw, tile_len=512), kw]
    _, mixed = run(True, kws)
    _, mixed_sep = run(False, kws)
    for k in range(3):
        for q in range(2):
            assert bits_equal(mixed[k][0][q], mixed_sep[k][0][q]), (k, q)
    for d in dins:
        d.free()


_STAGED_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import leansdr_amd.capi as capi
from leansdr_amd import synth
ctx = capi.Ctx(0)
x, _ = synth.qpsk_baseband(4 * 60000, 4, seed=11, rms=50.0, snr_db=18.0)
out = {}
for sampler in (0, 1):
    for shift in (0, 1):                        # shift 1: the stream starts on the odd sample of a 16-byte piece
        r = capi.CstlnReceiver(ctx, sampler=sampler, cstln=1, omega=4.0, meas_decimation=4096, mode=capi.RX_TILED, tile_len=256, tile_warmup=256)
        n = len(x) - 8
        d_in = ctx.upload(x); d_out = ctx.alloc(n * 4)
        used = r.run_async(d_in.at(8 * shift), n, d_out.ptr, n)
        prod = r.wait()
        s = ctx.download(d_out, capi.SOFTSYM, prod)
        out[f"c{sampler}{shift}"] = s["cost"].copy(); out[f"s{sampler}{shift}"] = s["symbol"].copy(); out[f"u{sampler}{shift}"] = np.array([used, prod])
        st = r.state()
        out[f"t{sampler}{shift}"] = np.array([st.mu, st.phase, st.freqw, st.agc_gain, st.est_insp], np.float32)
        r.close(); d_in.free(); d_out.free()
np.savez(sys.argv[2], **out)
"""


def test_staged_cf32_tiles_equal_direct_loads(tmp_path):
    """cf32 input, nearest / linear sampler, 64 tiles per wavefront: the tolerance tiles take their samples out of LDS stages
    (rx_stage<LSDR_IN_CF32>) instead of per-symbol global loads — the same samples, the same arithmetic, bit for bit the same run,
    whether the stream starts on a 16-byte boundary or not.  And 32 tiles per wavefront give that run too: the estimator maps are
    composed in groups of 32 tiles either way.  (Loads and lanes are chosen once per process: three processes.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "staged.py"
    script.write_text(_STAGED_SCRIPT)
    res = []
    for tag, extra in (("staged", {"LSDR_RX_LANES": "64"}), ("direct", {"LSDR_RX_NO_LDS": "2", "LSDR_RX_LANES": "64"}),
                       ("direct32", {"LSDR_RX_NO_LDS": "2", "LSDR_RX_LANES": "32"})):
        f = str(tmp_path / f"{tag}.npz")
        p = subprocess.run([sys.executable, str(script), root, f], env=dict(os.environ, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        res.append(np.load(f))
    a = res[0]
    assert len(a.files) == 16 and a["u10"][1] > 50000
    for b in res[1:]:
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert a[k].tobytes() == b[k].tobytes(), k


@pytest.mark.parametrize("ppm,snr_db,tile", [(30.0, 20.0, (0, 0)), (-50.0, 20.0, (256, 256)), (40.0, 15.0, (1024, 512))])
def test_tiles_find_the_symbol_timing_wherever_they_start(capi, ctx, oracle, ppm, snr_db, tile):
    """A receiver whose omega is `ppm` off the stream's symbol rate (a real sample clock): the symbol timing drifts against the tile
    grid, so tiles start at EVERY timing phase — including half a symbol off, the unstable point of the timing detector, where a
    few tiles in a thousand used to leave their warm-up unconverged (the reference benchmark's 4.2-sps series lost its lock in the
    tiled mode: profiles/r06_sensitivity/README.md).  With the fed-forward estimate (rx_wave_timing) no seam stays unreconciled and the
    decisions are the sequential loop's."""
    x, _ = synth.qpsk_baseband(4 * 400000, 4, seed=21, rms=50.0, snr_db=snr_db)
    omega = float(np.float32(4.0 * (1 + ppm * 1e-6)))
    p = po.rx_params(sampler=1, cstln=1, omega=omega, meas_decimation=4096)
    acq = 40960
    a = oracle.rx(p, x[:acq + 1])
    ref = oracle.rx(p, x[acq:], state_in=a["state"])
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=omega, meas_decimation=4096, mode=capi.RX_TILED, tile_len=tile[0], tile_warmup=tile[1])
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    # in runs of 256 Ki samples, like a scheduler with --buf-factor 64 pipes: the state a run leaves is the next one's start
    syms, pos, bad, tiles = [], acq, 0, 0
    while pos + 129 <= len(x):
        out = r.run(x[pos:pos + 262144])
        if not out["consumed"]:
            break
        syms.append(out["sym"]); pos += out["consumed"]
        stt = r.tiled_stats(); bad += stt["bad_seams"]; tiles += stt["tiles"]
    r.close()
    got = np.concatenate(syms)
    # the timing swept through more than a whole symbol over the stream
    assert abs(ppm) * 1e-6 * len(x) > 4.0
    rep = check_tiled(got, ref["sym"][: len(got)] if len(got) <= len(ref["sym"]) else ref["sym"], dict(tiles=tiles, bad_seams=bad, dup=0, miss=0),
                      tol=TOL if snr_db >= 18 else LOW_SNR)
    assert len(got) == len(ref["sym"]) and rep["pass"], rep

"""notch_fir — auto_notch (one slot) fused into fir_filter (lsdr_notch_fir_*, the reference's default graph `--anf 1`,
leandvb.cc:296-301) — against the oracle's chain auto_notch (sdr.h:46-138) → fir_filter (dsp.h:233-262) in the reference's arithmetic.
A tolerance mode (include/lsdr_hip.h): same bins; every output within 2e-5 of full scale of the reference's arithmetic for bins
below 2048, within 1e-3 for bins from 2048 on — where the reference's phasor table is cosf/sinf of a float-ROUNDED angle of up to
25 736 rad (±1e-3 rad of table noise; measured effect up to 3.8e-4 of full scale) — and, for EVERY bin, within 1e-5 of the float64
restatement of the block's own stated arithmetic (oracle.notch_fir_ideal: exact phases)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_TAPS, DECIM = 313, 30


def signal(n, seed, tones, noise=12.0):
    """noise + CW tones: [(first sample, cycles per sample, amplitude)], each lasting until the next one starts"""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * noise
    for i, (s0, f, a) in enumerate(tones):
        s1 = tones[i + 1][0] if i + 1 < len(tones) else n
        x[s0:s1] += a * np.exp(2j * np.pi * f * t[s0:s1])
    return x.astype(np.complex64)


def oracle_chain(oracle, x, coeffs, decimation, k=0.002, scale=None):
    xs = oracle.scaler(scale, x) if scale else x
    xn, bins = oracle.auto_notch(xs, 1, decimation, k, 0.0)
    y, cons = oracle.fir_filter(coeffs, DECIM, xn)
    return y, bins[0], len(xn)


def c2_taps(capi):
    return capi.lowpass(N_TAPS - 1, np.float32((2e6 / 2) * (1 + 0.35 / 2) / 240e6))


def rel_err(y, ref):
    return float(np.abs(y.astype(np.complex128) - ref.astype(np.complex128)).max() / np.abs(ref).max())


@pytest.mark.parametrize("case", ["positive_bin", "negative_bin", "bin_changes", "tone_in_stopband"])
def test_one_run_matches_the_oracle_chain(capi, ctx, oracle, case):
    """The whole stream in one run: pass-through before the first detect, the first detect (a bin appears), later detects with the
    same bin or a new one — every output of fir_filter(auto_notch(x)) within the stated bound, same final bin."""
    dec = 4096 * 8
    n = 4096 * 60 + 1234
    tones = {"positive_bin": [(0, 0.0031, 40.0)], "negative_bin": [(0, -0.0021, 55.0)],
             "bin_changes": [(0, 0.0031, 40.0), (4096 * 21 + 100, -0.0021, 55.0), (4096 * 43, 0.0007, 30.0)],
             "tone_in_stopband": [(0, 0.0137, 36.0)]}[case]
    x = signal(n, 3, tones)
    c = c2_taps(capi)
    ref, ref_bin, n_notched = oracle_chain(oracle, x, c, dec)
    nf = capi.NotchFir(ctx, c, DECIM, decimation=dec)
    y, cons = nf.run(x)
    assert nf.bin() == ref_bin
    assert len(y) == (n_notched - N_TAPS) // DECIM and cons == len(y) * DECIM
    err = rel_err(y, ref[:len(y)])
    bound = 2e-5 if case in ("positive_bin", "tone_in_stopband") else 1e-3
    assert err <= bound, (case, err)
    ideal, _ = oracle.notch_fir_ideal(x, c, DECIM, dec)
    assert rel_err(y, ideal[:len(y)]) <= 1e-5, (case, rel_err(y, ideal[:len(y)]))
    nf.close()


@pytest.mark.parametrize("step", [4096 + 400, 3 * 4096 + 17, 50000])
def test_many_small_runs_give_the_same_stream(capi, ctx, oracle, step):
    """The block carries its state from run to run (last output, the raw samples in front of the read pointer, the bin, the
    estimator at the notch's frontier — needed when a LATER run's detect changes the bin): runs of a block or two, bin changes
    included, against the oracle chain over the whole stream."""
    dec = 4096 * 4
    n = 4096 * 50
    x = signal(n, 11, [(0, 0.0024, 35.0), (4096 * 17 + 2000, -0.0035, 45.0), (4096 * 34, 0.0024, 35.0)])
    c = c2_taps(capi)
    ref, ref_bin, n_notched = oracle_chain(oracle, x, c, dec)
    nf = capi.NotchFir(ctx, c, DECIM, decimation=dec)
    y, cons = nf.run(x, step=step)
    assert nf.bin() == ref_bin
    assert len(y) >= (n_notched - 4096 - N_TAPS) // DECIM          # (the last run may stop a block short of the oracle's)
    assert rel_err(y, ref[:len(y)]) <= 1e-3                         # (bin 4082 in the middle segment: the reference's table noise)
    ideal, _ = oracle.notch_fir_ideal(x, c, DECIM, dec)
    assert rel_err(y, ideal[:len(y)]) <= 1e-5
    quiet = np.r_[0:(4096 * 17) // DECIM, (4096 * 40) // DECIM:len(y)]       # the two segments with bin 10
    assert rel_err(y[quiet], ref[quiet]) <= 2e-5
    nf.close()


def test_fused_scaler_and_default_decimation(capi, ctx, oracle):
    """in_scale rides on the taps (scaler → auto_notch → fir_filter, leandvb.cc:259-301); the reference's decimation (1024·4096): the
    first 4 190 208 samples pass through, the detect at block 1023 starts the notch."""
    n = 4096 * 1100
    x = signal(n, 5, [(0, 0.0137, 2.1)], noise=0.5)
    c = c2_taps(capi)
    ref, ref_bin, n_notched = oracle_chain(oracle, x, c, 1024 * 4096, scale=75.0)
    nf = capi.NotchFir(ctx, c, DECIM, in_scale=75.0)
    y, cons = nf.run(x)
    assert nf.bin() == ref_bin == 56
    assert len(y) == (n_notched - N_TAPS) // DECIM
    assert rel_err(y, ref[:len(y)]) <= 2e-5
    nf.close()


def test_unsupported_geometries_are_refused(capi, ctx):
    c = c2_taps(capi)
    for kw in (dict(nslots=2), dict(decim=10), dict(coeffs=np.ones(340, np.float32))):
        with pytest.raises(capi.LsdrError):
            capi.NotchFir(ctx, kw.get("coeffs", c), kw.get("decim", DECIM), nslots=kw.get("nslots", 1))


def test_overlapped_runs_give_the_same_bits(capi, ctx, oracle):
    """lsdr_notch_fir_set_overlap: detect chain and filter pass of run k+1 on the block's own streams next to run k's tail — the same
    output bit for bit, over runs of a few blocks each with bin changes in them."""
    dec = 4096 * 4
    x = signal(4096 * 50, 11, [(0, 0.0024, 35.0), (4096 * 17 + 2000, -0.0035, 45.0), (4096 * 34, 0.0024, 35.0)])
    c = c2_taps(capi)
    outs = []
    for ov in (False, True):
        nf = capi.NotchFir(ctx, c, DECIM, decimation=dec)
        if ov:
            nf.set_overlap(True)
        y, cons = nf.run(x, step=3 * 4096 + 17)
        outs.append((y, cons, nf.bin()))
        nf.close()
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    assert np.array_equal(outs[0][0].view(np.uint64), outs[1][0].view(np.uint64))


def test_taps_follow_the_carrier_like_fir_filter(capi, ctx, oracle):
    """fir_filter's tracking (dsp.h:236-244): the taps are re-shifted between two runs — twice here, with a bin change in between — and
    every output, the first ones after each shift included, is what the oracle's fir_filter with the same shift schedule gives over
    the oracle's notched stream (the recurrence is re-anchored under the new taps; complex taps through k_nf_taps / k_nf_fix)."""
    dec = 4096 * 4
    n = 4096 * 48
    x = signal(n, 21, [(0, 0.0021, 30.0), (4096 * 22 + 500, 0.0032, 38.0)])
    c = c2_taps(capi)
    xn, bins = oracle.auto_notch(x, 1, dec, 0.002, 0.0)
    nf = capi.NotchFir(ctx, c, DECIM, decimation=dec)
    din = ctx.upload(x)
    dout = ctx.alloc((n // DECIM + 16) * 8)
    sched = [(4096 * 15 + 77, 0.0), (4096 * 33, 0.0023), (n, -0.0011)]       # (feed up to here, with this shift)
    pos = nout = 0
    got_freqs = []
    for upto, freq in sched:
        if freq != nf.current_freq:
            assert nf.track(freq * 4, 0.25, 1e-4) and abs(nf.current_freq - np.float32(freq * 4) * np.float32(0.25)) < 1e-9
        got_freqs.append((nout, nf.current_freq))
        cons, prod = nf.run_dev(din.at(pos * 8), upto - pos, dout.at(nout * 8), n // DECIM + 16 - nout)
        pos += cons; nout += prod
    y = ctx.download(dout, np.complex64, nout)
    # the oracle: the same shift schedule by output index over the notched stream
    ref = np.zeros(nout, np.complex64)
    for k, (m0, f) in enumerate(got_freqs):
        m1 = got_freqs[k + 1][0] if k + 1 < len(got_freqs) else nout
        seg, _ = oracle.fir_filter(c, DECIM, xn[m0 * DECIM:], freq=f)
        ref[m0:m1] = seg[:m1 - m0]
    assert nout > 6000 and len(got_freqs) == 3 and got_freqs[1][0] > 0 and got_freqs[2][0] > got_freqs[1][0]
    assert nf.bin() == bins[0]
    assert rel_err(y, ref) <= 2e-5, rel_err(y, ref)
    nf.close(); din.free(); dout.free()

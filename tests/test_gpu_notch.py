"""auto_notch / cnr_fft on the GPU (and the host cfft they share) against the reference's goldens and
the oracle.  auto_notch is bit-exact: time tiles are verified seam by seam, unverified spans are redone
sequentially."""
import hashlib
import os
import ctypes as C
import numpy as np
import pytest
from conftest import gold, bits_equal, iq16_to_cf32


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_cfft_host_golden(capi, oracle):
    """(CPU) the product's host FFT — used by detect() and cnr_fft — equals the reference's."""
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    assert bits_equal(capi.cfft_host(x[:4096], True), g["fft4096_rev"])
    assert bits_equal(capi.cfft_host(x[:1024], False), g["fft1024_fwd"])


@pytest.mark.gpu
def test_auto_notch_golden(capi, ctx, oracle):
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    for ns in (1, 2):
        a = capi.AutoNotch(ctx, ns, 0.0, 4096 * 3)
        y = a.run(x)
        assert a.bins() == g[f"anf{ns}_bins"].tolist()
        assert sha(y) == bytes(g[f"anf{ns}_sha"]).hex() and bits_equal(y[-512:], g[f"anf{ns}_tail"])
        a.close()
    a = capi.AutoNotch(ctx, 1, 30.0, 4096 * 3)
    y = a.run(x)
    assert sha(y) == bytes(g["anf_agc_sha"]).hex()
    a.close()


@pytest.mark.gpu
def test_auto_notch_passthrough_before_first_detect(capi, ctx, oracle):
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    a = capi.AutoNotch(ctx, 1, 0.0)            # default decimation: no detect within this input
    y = a.run(x[:4096 * 5 + 100])
    assert len(y) == 4096 * 5 and bits_equal(y, x[:4096 * 5])
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nslots", [1, 3])
def test_auto_notch_long_stream_vs_oracle(capi, ctx, oracle, nslots):
    """Many tiles between detections, state carried across calls: still bit-exact, and the verification
    finds (almost) every speculative tile converged."""
    rng = np.random.default_rng(11)
    n = 4096 * 300
    t = np.arange(n)
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 12 + 70 * np.exp(2j * np.pi * 0.0713 * t)
         + 40 * np.exp(-2j * np.pi * 0.27 * t) + 25 * np.exp(2j * np.pi * 0.4 * t)).astype(np.complex64)
    want, wbins = oracle.auto_notch(x, nslots, 4096 * 100)
    a = capi.AutoNotch(ctx, nslots, 0.0, 4096 * 100)
    y1 = a.run(x[: 4096 * 130 + 17])
    st1 = a.stats()
    y2 = a.run(x[len(y1):])
    st2 = a.stats()
    assert a.bins() == wbins
    a.close()
    got = np.concatenate([y1, y2])
    assert len(got) == len(want) and bits_equal(got, want)
    assert st1["tiles"] + st2["tiles"] > 50
    assert st1["bad_seams"] + st2["bad_seams"] <= (st1["tiles"] + st2["tiles"]) // 4, (st1, st2)


@pytest.mark.gpu
def test_cnr_fft_golden(capi, ctx, oracle):
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    c = capi.CnrFft(ctx, 0.2, 4096, 4096 * 2)
    out, cons = c.run(x, 0.01, 0.5)
    c.close()
    assert bits_equal(out, g["cnr"]) and cons == len(x)
    with pytest.raises(capi.LsdrError):
        capi.CnrFft(ctx, 0.3)                  # "CNR estimator requires Fsampling > 4x Fsignal"


@pytest.mark.gpu
def test_spectrum_golden(capi, ctx, oracle):
    g = gold("auto_notch.npz")
    s = gold("spectrum.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    for dec, k, key in [(4096, 0.5, "d4096_k05"), (3000, 0.1, "d3000_k01")]:
        sp = capi.Spectrum(ctx, dec, k)
        out, cons = sp.run(x)
        sp.close()
        assert bits_equal(out, s[key]) and cons == len(x)
    # state carries across calls: two halves == one call
    sp = capi.Spectrum(ctx, 3000, 0.1)
    a, ca = sp.run(x[:10240])
    b, cb = sp.run(x[ca:])
    sp.close()
    assert bits_equal(np.concatenate([a, b]), s["d3000_k01"])


@pytest.mark.gpu
def test_rotator_golden(capi, ctx, oracle):
    import hashlib
    g = gold("rotator.npz")
    t = np.arange(70000)
    x = ((t % 251) - 125 + 1j * ((t * 7) % 199 - 99)).astype(np.complex64)
    for name, f in (("p01", 0.01), ("m123", -0.123)):
        r = capi.Rotator(ctx, f)
        y = np.concatenate([r.run(x[:5]), r.run(x[5:40000]), r.run(x[40000:])])
        r.close()
        assert hashlib.sha256(y.tobytes()).digest() == bytes(g[name + "_sha"])
        assert bits_equal(y, oracle.rotator(x, f))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 1024, 64, 8192])
def test_cfft_on_device_is_the_reference_fft(capi, ctx, oracle, n):
    """k_cfft (what detect()/cnr_fft/spectrum run): same bits as cfft_engine<float>::inplace, both directions."""
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))[:n]
    for rev in (True, False):
        assert bits_equal(capi.cfft_dev(ctx, x, rev), oracle.cfft(x, rev))
    if n == 4096:
        assert bits_equal(capi.cfft_dev(ctx, x, True), g["fft4096_rev"])


# ---- throughput mode (LSDR_NOTCH_SCAN): the recurrence as a single-pass scan, detect() on the device --------------------
NOTCH_RTOL = 2e-5     # max |out − oracle| relative to the largest |oracle| sample (re-associated carries, device libm)


@pytest.mark.gpu
@pytest.mark.parametrize("nslots", [1, 2, 3])
def test_scan_mode_vs_oracle(capi, ctx, oracle, nslots):
    """Strong CW interferers on noise: same detected bins as the reference, output within NOTCH_RTOL of the exact result,
    for any cut of the stream into calls (state, bins and block phase are carried on the device)."""
    rng = np.random.default_rng(11)
    n = 4096 * 300
    t = np.arange(n)
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 12 + 70 * np.exp(2j * np.pi * 0.0713 * t)
         + 40 * np.exp(-2j * np.pi * 0.27 * t) + 25 * np.exp(2j * np.pi * 0.4 * t)).astype(np.complex64)
    want, wbins = oracle.auto_notch(x, nslots, 4096 * 100)
    a = capi.AutoNotch(ctx, nslots, 0.0, 4096 * 100, mode=capi.NOTCH_SCAN)
    parts, pos = [], 0
    for cut in (4096 * 130 + 17, 4096 * 3, 4096 * 1 + 5, n):
        y = a.run(x[pos:cut if cut > pos else n])
        parts.append(y)
        pos += len(y)
        if pos >= n // 4096 * 4096:
            break
    assert a.bins() == wbins
    a.close()
    got = np.concatenate(parts)
    assert len(got) == len(want)
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err <= NOTCH_RTOL, err
    # the interferers are gone: what is left is the noise (power 2·12²), as in the exact result
    tail = slice(4096 * 120, None)
    assert abs(np.mean(np.abs(got[tail]) ** 2) / np.mean(np.abs(want[tail]) ** 2) - 1) < 1e-4


@pytest.mark.gpu
def test_scan_mode_is_a_passthrough_before_the_first_detect(capi, ctx, oracle):
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    a = capi.AutoNotch(ctx, 1, 0.0, mode=capi.NOTCH_SCAN)            # default decimation: no detect within this input
    y = a.run(x[:4096 * 5 + 100])
    a.close()
    assert len(y) == 4096 * 5 and bits_equal(y, x[:4096 * 5])


@pytest.mark.gpu
def test_scan_mode_golden_bins_and_tolerance(capi, ctx, oracle):
    """The reference's own fixture (tests/golden/auto_notch.npz): same bins; output within tolerance of the golden tail."""
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    for ns in (1, 2):
        want, _ = oracle.auto_notch(x, ns, 4096 * 3)
        a = capi.AutoNotch(ctx, ns, 0.0, 4096 * 3, mode=capi.NOTCH_SCAN)
        y = a.run(x)
        assert a.bins() == g[f"anf{ns}_bins"].tolist()
        a.close()
        assert np.max(np.abs(y - want)) / np.max(np.abs(want)) <= NOTCH_RTOL
        assert np.max(np.abs(y[-512:] - g[f"anf{ns}_tail"])) / np.max(np.abs(want)) <= NOTCH_RTOL


@pytest.mark.gpu
def test_scan_mode_refuses_what_it_does_not_implement(capi, ctx):
    with pytest.raises(capi.LsdrError):
        capi.AutoNotch(ctx, 1, 30.0, mode=capi.NOTCH_SCAN)      # AGC set point: exact mode only
    with pytest.raises(capi.LsdrError):
        capi.AutoNotch(ctx, 6, 0.0, mode=capi.NOTCH_SCAN)


@pytest.mark.gpu
@pytest.mark.parametrize("nslots", [1, 3])
def test_scan_hand_off_never_reads_a_stale_total(nslots):
    """Stress test of k_notch_scan's cross-workgroup hand-off (tools/notch_poison_stress.py: garbage into the hand-off buffers
    before each of 1000 runs, every output bit-identical).  The poison hook exists only in the measure build of the library
    (-DLSDR_MEASURE), so the stress runs in its own process on tools/variants/liblsdr_hip_measure.so."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "variants", "liblsdr_hip_measure.so")
    assert os.path.exists(lib), "make -C leansdr_amd/csrc measure"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "notch_poison_stress.py"), str(nslots)],
                       env=dict(os.environ, LSDR_HIP_LIB=lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "stress OK: 1000 runs" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.gpu
def test_scan_overlapped_detect_chain_gives_the_same_output(capi, ctx):
    """lsdr_auto_notch_set_overlap: the detect chain of run k+1 on a side stream next to the scan of run k — bit-identical output
    and bins over eight queued runs with a detect point in each (two buffer sets used alternately, bins carried from chain to chain)."""
    rng = np.random.default_rng(31)
    n = 4096 * 40
    t = np.arange(8 * n)
    x = ((rng.standard_normal(8 * n) + 1j * rng.standard_normal(8 * n)) * 12 + 70 * np.exp(2j * np.pi * (0.0713 + 1e-8 * t) * t)).astype(np.complex64)
    d_in = ctx.upload(x)
    outs = {}
    for ov in (0, 1):
        a = capi.AutoNotch(ctx, 2, 0.0, 4096 * 16, mode=capi.NOTCH_SCAN)
        capi.check(capi.lib.lsdr_auto_notch_set_overlap(a.h, ov))
        d_out = ctx.alloc(8 * n * 8)
        for k in range(8):
            a.run_dev(d_in.at(k * n * 8), n, d_out.at(k * n * 8), n)
        outs[ov] = (ctx.download(d_out, np.complex64, 8 * n).copy(), a.bins())
        a.close(); d_out.free()
    assert outs[0][1] == outs[1][1] and np.array_equal(outs[0][0].view(np.uint64), outs[1][0].view(np.uint64))
    d_in.free()

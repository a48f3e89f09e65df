"""The oracle against the REAL reference run here (oracle/_ref, built from
/root/reference by oracle/Makefile).  Skipped where oracle/_ref is absent —
the committed goldens (test_oracle_golden.py) cover that case."""
import numpy as np
import pytest
from conftest import bits_equal
import pyoracle as po


@pytest.fixture(scope="module")
def sig():
    rng = np.random.default_rng(7)
    n = 30000
    # QPSK-like stream at 4 samples/symbol with noise, amplitude ~75 (enough to lock)
    syms = (rng.integers(0, 2, n // 4 + 8) * 2 - 1) + 1j * (rng.integers(0, 2, n // 4 + 8) * 2 - 1)
    up = np.zeros(len(syms) * 4, np.complex64)
    up[::4] = syms
    h = np.hanning(9).astype(np.float32)
    x = np.convolve(up, h)[:n] * 40 + (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 6
    return x.astype(np.complex64)


def test_tables_all_constellations(oracle, ref):
    assert bits_equal(oracle.trig16(), ref.trig16())
    for pre, fec in [(0, 0), (1, 0), (2, 1), (3, 1), (3, 3), (3, 8), (4, 3), (4, 8), (5, 0), (6, 0), (7, 0), (8, 0)]:
        a, b = oracle.cstln_lut(pre, fec), ref.cstln_lut(pre, fec)
        for k in ("symbols", "cost", "symbol", "phase_error"):
            assert bits_equal(a[k], b[k]), (pre, fec, k)
    for a in [-1.5, 70000.7, 0.0, -65536.5, 32767.9, -32768.2, 1e6, -1e6 + 0.3, 123456.7]:
        assert oracle.lib.lo_trig16_index(a) == ref.lib.ref_trig16_index(a)


@pytest.mark.parametrize("order,fcut", [(312, (2e6 / 2) * (1 + 0.35 / 2) / 240e6), (14, 0.4895), (40, 0.1), (1, 0.3), (99, 0.02)])
def test_lowpass(oracle, ref, order, fcut):
    assert bits_equal(oracle.lowpass(order, np.float32(fcut)), ref.lowpass(order, np.float32(fcut)))
    assert bits_equal(oracle.lowpass(order, np.float32(fcut), False), ref.lowpass(order, np.float32(fcut), False))


@pytest.mark.parametrize("order,fs,ro", [(166, 2e6 / (8e6 * 16), 0.35), (41, 0.25, 0.35), (64, 0.25, 0.25), (100, 0.5, 0.2), (33, 1 / 3.0, 0.35)])
def test_rrc(oracle, ref, order, fs, ro):
    assert bits_equal(oracle.rrc(order, np.float32(fs), np.float32(ro)), ref.rrc(order, np.float32(fs), np.float32(ro)))


def test_elementwise(oracle, ref, sig):
    rng = np.random.default_rng(1)
    u8 = rng.integers(0, 256, 2 * 5000, dtype=np.uint8)
    assert bits_equal(oracle.cconverter_u8(u8), ref.cconverter_u8(u8))
    assert bits_equal(oracle.scaler(0.0123, sig), ref.scaler(0.0123, sig))


@pytest.mark.parametrize("n,d,freq", [(313, 30, 0.0), (313, 30, 0.001), (313, 30, -0.0123), (21, 1, 0.0), (21, 7, 0.05), (2, 3, 0.2), (64, 64, 0.0)])
def test_fir_filter(oracle, ref, sig, n, d, freq):
    c = oracle.lowpass(n - 1, np.float32(0.4 / d))
    a, cons = oracle.fir_filter(c, d, sig, freq)
    b, sc = ref.fir_filter(c, d, sig, freq)
    assert bits_equal(oracle.fir_shift(c, freq), sc)
    assert bits_equal(a, b) and cons == len(a) * d


def test_fir_resampler(oracle, ref, sig):
    rr = oracle.rrc(41, np.float32(0.25), np.float32(0.35))
    for f in (0.0, 0.01):
        a, _ = oracle.fir_resampler(rr, 4, sig[:3000], f)
        assert bits_equal(a, ref.fir_resampler(rr, 4, sig[:3000], f))


def test_fft_notch_cnr(oracle, ref, sig):
    assert bits_equal(oracle.cfft(sig[:4096]), ref.cfft(sig[:4096]))
    assert bits_equal(oracle.cfft(sig[:1024], True), ref.cfft(sig[:1024], True))
    t = np.arange(4096 * 7)
    x = (np.resize(sig, len(t)) * 0.2 + 60 * np.exp(2j * np.pi * 0.123 * t) + 30 * np.exp(-2j * np.pi * 0.31 * t)).astype(np.complex64)
    for ns, sp in [(1, 0.0), (2, 0.0), (3, 0.0), (1, 30.0)]:
        a, ba = oracle.auto_notch(x, ns, 4096 * 3, setpoint=sp)
        b, bb = ref.auto_notch(x, ns, 4096 * 3, setpoint=sp)
        assert ba == [int(v) for v in bb] and bits_equal(a, b)
    assert bits_equal(oracle.cnr_fft(x, 0.2, 4096, 8192, 0.01, 0.5), ref.cnr_fft(x, 0.2, 4096, 8192, 0.01, 0.5))
    for dec, k in [(2048, 0.5), (5000, 0.1), (1024, 0.9)]:
        assert bits_equal(oracle.spectrum(x, dec, k), ref.spectrum(x, dec, k))


def same_rx(a, b):
    assert bits_equal(a["sym"]["cost"], b["sym"]["cost"]) and bits_equal(a["sym"]["symbol"], b["sym"]["symbol"])
    for k in ("freq", "ss", "mer", "cstln"):
        assert bits_equal(a[k], b[k]), k
    sa, sb = a["state"].as_dict(), b["state"].as_dict()
    for k in sa:
        if k == "hist":
            assert np.array(sa[k], np.float32).tobytes() == np.array(sb[k], np.float32).tobytes()
        elif k == "meas_count":
            assert sa[k] == sb[k]
        else:
            assert np.float32(sa[k]).tobytes() == np.float32(sb[k]).tobytes(), k


@pytest.mark.parametrize("kw", [
    dict(sampler=1, cstln=1, omega=4.0, meas_decimation=4096),
    dict(sampler=0, cstln=1, omega=4.0, meas_decimation=1000),
    dict(sampler=1, cstln=1, omega=4.0, freq=0.01, allow_drift=1, meas_decimation=4096),
    dict(sampler=1, cstln=1, omega=4.0, freq=-0.03, meas_decimation=4096),
    dict(sampler=1, cstln=2, fec=1, omega=4.0, meas_decimation=4096),
    dict(sampler=1, cstln=0, omega=4.0, meas_decimation=4096),
    dict(sampler=1, cstln=3, fec=3, omega=4.0, meas_decimation=4096),
    dict(sampler=1, cstln=1, omega=3.7, pll_adjustment=1 / 6.0, meas_decimation=256),
    dict(sampler=1, cstln=1, omega=4.0, kest=0.05, meas_decimation=4096),
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items() if a not in ("meas_decimation",)))
def test_cstln_receiver(oracle, ref, sig, kw):
    p = po.rx_params(**kw)
    same_rx(oracle.rx(p, sig), ref.rx(p, sig))


def test_cstln_receiver_fir_sampler(oracle, ref, sig):
    rr = oracle.rrc(int(10 * 8e6 * 16 / (22 * (2e6 / 2) * 0.35)), np.float32(2e6 / (8e6 * 16)), np.float32(0.35))
    p = po.rx_params(sampler=2, coeffs=rr, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096)
    same_rx(oracle.rx(p, sig * np.float32(16)), ref.rx(p, sig * np.float32(16)))
    rr1 = oracle.rrc(40, np.float32(0.25), np.float32(0.35))
    p = po.rx_params(sampler=2, coeffs=rr1, subsampling=1, cstln=1, omega=4.0, meas_decimation=4096)
    same_rx(oracle.rx(p, sig), ref.rx(p, sig))


def test_rotator(oracle, ref, sig):
    for f in (0.01, -0.123, 0.4999, 1e-5):
        assert bits_equal(oracle.rotator(sig, f, splits=(7, 3000)), ref.rotator(sig, f))


@pytest.mark.parametrize("nslots", [1, 2, 3])
def test_auto_notch_long_interferer_stream_pinned(oracle, ref, nslots):
    """(CPU) the stream the GPU's scan-mode notch is held against (tests/test_gpu_notch.py::test_scan_mode_vs_oracle: three CW
    interferers on noise, 300 blocks, detect every 100): the oracle's output and bins are the compiled reference's, bit for bit."""
    rng = np.random.default_rng(11)
    n = 4096 * 300
    t = np.arange(n)
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 12 + 70 * np.exp(2j * np.pi * 0.0713 * t)
         + 40 * np.exp(-2j * np.pi * 0.27 * t) + 25 * np.exp(2j * np.pi * 0.4 * t)).astype(np.complex64)
    a, ba = oracle.auto_notch(x, nslots, 4096 * 100)
    b, bb = ref.auto_notch(x, nslots, 4096 * 100)
    assert ba == [int(v) for v in bb] and bits_equal(a, b)

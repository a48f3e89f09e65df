"""The plain-C oracle against the golden vectors generated from the REAL
reference (oracle/make_golden.py).  Bit-exact for every block."""
import hashlib
import json
import os
import numpy as np
import pytest
from conftest import gold, bits_equal, iq16_to_cf32, GOLD
import pyoracle as po


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_tables(oracle):
    g = gold("tables.npz")
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    trig = oracle.trig16()
    assert sha(trig) == man["trig16"]["sha256"]
    assert bits_equal(trig[g["trig_sample_idx"]], g["trig_sample"])
    for name, ent in man.items():
        if not name.startswith("cstln_"):
            continue
        t = oracle.cstln_lut(ent["predef"], ent["fec"])
        assert t["nsymbols"] == ent["nsymbols"]
        assert t["symbols"].tolist() == ent["symbols"]
        assert sha(t["cost"]) == ent["cost"] and sha(t["symbol"]) == ent["symbol"]
        assert sha(t["phase_error"]) == ent["phase_error"], name
    q = oracle.cstln_lut(po.Oracle and 1, 0)
    assert bits_equal(q["cost"], g["qpsk_cost"]) and bits_equal(q["phase_error"], g["qpsk_pe"])
    assert bits_equal(oracle.lowpass(312, float(g["lowpass_c2_fcut"])), g["lowpass_c2"])
    assert bits_equal(oracle.lowpass(14, np.float32(0.4895)), g["lowpass_small"])
    assert bits_equal(oracle.rrc(int(10 * 8e6 * 16 / (22 * (2e6 / 2) * 0.35)), np.float32(2e6 / (8e6 * 16)),
                                 np.float32(0.35)), g["rrc_rx"])
    assert bits_equal(oracle.rrc(41, np.float32(0.25), np.float32(0.35)), g["rrc_tx"])


def test_known_answers_survey_appendix(oracle):
    """Known-answer values recorded in SURVEY.md Appendix A (dumped from the compiled reference)."""
    assert oracle.lib.lo_trig16_index(-1.5) == 65535 and oracle.lib.lo_trig16_index(70000.7) == 4464
    trig = oracle.trig16()
    assert trig[1].real == np.float32(1) and abs(trig[1].imag - 9.58738019e-05) < 1e-12
    q = oracle.cstln_lut(1, 0)
    assert q["symbols"].tolist() == [[53, 53], [53, -53], [-53, 53], [-53, -53]]

    def look(i, qq):
        idx = (i & 255) * 256 + (qq & 255)
        return int(q["symbol"][idx]), int(q["cost"][idx]), int(q["phase_error"][idx])
    assert look(53, 53) == (0, -11236, 0)
    assert look(0, 0) == (0, 0, -8192)
    assert look(127, -128) == (1, -21666, -40)
    assert look(-1, 1) == (2, -212, 0)
    assert look(10, -90) == (1, -2120, -7037)
    i = oracle.lib.lo_cstln_lookup_index(300.0, -20.0)  # halved twice to (75,-5)
    assert (int(q["symbol"][i]), int(q["cost"][i]), int(q["phase_error"][i])) == (1, -1060, 7497)
    c = oracle.lowpass(312, np.float32((2e6 / 2) * (1 + 0.35 / 2) / 240e6))
    assert len(c) == 313 and abs(c[0] + 0.00201203022) < 1e-10 and abs(c[156] - 0.00969144143) < 1e-10
    r = oracle.rrc(int(10 * 8e6 * 16 / (22 * (2e6 / 2) * 0.35)), np.float32(2e6 / (8e6 * 16)), np.float32(0.35))
    assert len(r) == 167 and abs(r[0] + 0.00279977219) < 1e-10 and abs(r[83] - 0.0163024738) < 1e-9


def test_fir_filter(oracle):
    g = gold("fir_filter.npz")
    tab = gold("tables.npz")
    xs = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq120"]))
    for tag, freq in [("f0", 0.0), ("fshift", 0.0123), ("fneg", -0.004)]:
        assert bits_equal(oracle.fir_shift(tab["lowpass_c2"], freq), g[f"c2_{tag}_sc"])
        y, cons = oracle.fir_filter(tab["lowpass_c2"], 30, xs, freq)
        assert bits_equal(y, g[f"c2_{tag}_out"]) and cons == len(y) * 30
    y, _ = oracle.fir_filter(tab["lowpass_small"], 1, xs[:3000])
    assert bits_equal(y, g["small_d1_out"])
    y, _ = oracle.fir_filter(tab["lowpass_small"], 7, xs[:3001], 0.05)
    assert bits_equal(y, g["small_d7_shift_out"])
    y, _ = oracle.fir_filter(tab["lowpass_small"], 2, oracle.cconverter_u8(g["u8"]))
    assert bits_equal(y, g["u8_d2_out"])


def test_fir_filter_edges(oracle):
    c = np.array([0.25, 0.5, 0.25], np.float32)
    x = np.arange(10, dtype=np.float32).astype(np.complex64)
    y, cons = oracle.fir_filter(c, 1, x[:2])          # fewer than ncoeffs samples: no progress
    assert len(y) == 0 and cons == 0
    y, cons = oracle.fir_filter(c, 1, x[:3])          # exactly ncoeffs: (n-N)/D = 0 outputs
    assert len(y) == 0 and cons == 0
    y, cons = oracle.fir_filter(c, 4, x)              # (10-3)/4 = 1 output, consumes 4
    assert len(y) == 1 and cons == 4
    assert y[0] == np.complex64(0.25 * 3 + 0.5 * 2 + 0.25 * 1)  # x[N+0-i], i=0..2


def test_fir_resampler(oracle):
    g = gold("fir_resampler.npz")
    y, _ = oracle.fir_resampler(g["rrc_tx"], 4, g["sym"])
    assert bits_equal(y, g["out"])
    y, _ = oracle.fir_resampler(g["rrc_tx"], 4, g["sym"], 0.01)
    assert bits_equal(y, g["out_shift"])


RX_CASES = [
    ("lin4", dict(sampler=1, cstln=1, omega=4.0, meas_decimation=4096), "iq4", None),
    ("near4", dict(sampler=0, cstln=1, omega=4.0, meas_decimation=4096), "iq4", None),
    ("rrc4", dict(sampler=2, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096, pll_adjustment=1 / 6.0), "iq4", None),
    ("lin4_drift", dict(sampler=1, cstln=1, omega=4.0, freq=0.01, allow_drift=1, meas_decimation=4096), "iq4", None),
    ("lin4_psk8", dict(sampler=1, cstln=2, fec=1, omega=4.0, meas_decimation=4096), "iq4", 16384),
    ("lin4_bpsk", dict(sampler=1, cstln=0, omega=4.0, meas_decimation=4096), "iq4", 16384),
    ("lin4_loud", dict(sampler=1, cstln=1, omega=4.0, meas_decimation=4096), "iq4x7", 16384),
    ("lin1p2_u8", dict(sampler=1, cstln=1, omega=float(np.float32(2400e3 / 2000e3)), meas_decimation=2400), "u8", None),
]


def rx_input(oracle, g, src, limit):
    if src == "u8":
        x = oracle.cconverter_u8(g["u8"])
    else:
        x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq4"]))
        if src == "iq4x7":
            x = x[:limit] * np.float32(7)
    return x[:limit] if limit else x


def state_vec(st):
    d = st.as_dict()
    keys = ["mu", "phase", "freqw", "agc_gain", "est_insp", "est_sp", "est_ep", "freq_tap", "min_freqw", "max_freqw"]
    return np.array([d[k] for k in keys] + d["hist"], np.float32), int(d["meas_count"])


def check_rx_against_golden(r, g, tag):
    assert bits_equal(r["sym"]["cost"], g[f"{tag}_cost"]), tag
    assert bits_equal(r["sym"]["symbol"], g[f"{tag}_symbol"]), tag
    for k in ("freq", "ss", "mer", "cstln"):
        assert bits_equal(r[k], g[f"{tag}_{k}"]), (tag, k)
    sv, mc = state_vec(r["state"])
    assert bits_equal(sv, g[f"{tag}_state"]), tag
    assert mc == int(g[f"{tag}_meas_count"])


@pytest.mark.parametrize("tag,kw,src,limit", RX_CASES, ids=[c[0] for c in RX_CASES])
def test_cstln_receiver(oracle, tag, kw, src, limit):
    g = gold("cstln_receiver.npz")
    kw = dict(kw)
    if kw["sampler"] == 2:
        kw["coeffs"] = g["rrc_rx"]
    r = oracle.rx(po.rx_params(**kw), rx_input(oracle, g, src, limit))
    check_rx_against_golden(r, g, tag)


def test_cstln_receiver_chunking_invariance(oracle):
    """Feeding the stream in pieces (state carried over) gives the same symbols:
    the reference's output is --buf-factor independent."""
    g = gold("cstln_receiver.npz")
    x = rx_input(oracle, g, "iq4", None)
    p = po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=4096)
    whole = oracle.rx(p, x)
    import ctypes as C
    h = oracle.lib.lo_rx_new(C.byref(p))
    outs, pos = [], 0
    for piece in (5000, 129, 128, 17000, 1 << 30):
        seg = np.ascontiguousarray(x[pos:pos + piece])
        buf = np.zeros(len(seg) + 256, po.SOFTSYM)
        cons = C.c_size_t()
        n = oracle.lib.lo_rx_run(h, seg.ctypes.data, len(seg), buf.ctypes.data, len(buf), C.byref(cons),
                                 None, None, None, 0, None, None, 0, None)
        outs.append(buf[:n])
        pos += cons.value
    oracle.lib.lo_rx_free(h)
    got = np.concatenate(outs)
    assert bits_equal(got["cost"], whole["sym"]["cost"]) and bits_equal(got["symbol"], whole["sym"]["symbol"])


def test_auto_notch_fft_cnr(oracle):
    g = gold("auto_notch.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    for ns in (1, 2):
        y, bins = oracle.auto_notch(x, ns, 4096 * 3)
        assert bins == g[f"anf{ns}_bins"].tolist()
        assert sha(y) == bytes(g[f"anf{ns}_sha"]).hex()
        assert bits_equal(y[-512:], g[f"anf{ns}_tail"])
    y, _ = oracle.auto_notch(x, 1, 4096 * 3, setpoint=30.0)
    assert sha(y) == bytes(g["anf_agc_sha"]).hex()
    # pass-through contract before the first detect
    y, _ = oracle.auto_notch(x, 1, 4096 * 1000)
    assert bits_equal(y, x[: len(y)])
    assert bits_equal(oracle.cfft(x[:4096], True), g["fft4096_rev"])
    assert bits_equal(oracle.cfft(x[:1024], False), g["fft1024_fwd"])
    assert bits_equal(oracle.cnr_fft(x, 0.2, 4096, 4096 * 2, 0.01, 0.5), g["cnr"])


def test_spectrum(oracle):
    """spectrum<f32> (sdr.h:1347-1404): dB rows, fftshifted, EMA across spectra."""
    g = gold("auto_notch.npz")
    s = gold("spectrum.npz")
    x = oracle.scaler(float(g["scale"]), iq16_to_cf32(g["iq"]))
    assert bits_equal(oracle.spectrum(x, 4096, 0.5), s["d4096_k05"])
    assert bits_equal(oracle.spectrum(x, 3000, 0.1), s["d3000_k01"])
    assert s["d4096_k05"].shape == (8, 1024)


def rotator_input():
    t = np.arange(70000)
    return ((t % 251) - 125 + 1j * ((t * 7) % 199 - 99)).astype(np.complex64)


def test_rotator(oracle):
    """rotator<f32> (sdr.h:1226-1259): table of cosf/sinf(2π·i·ifreq/65536), 16-bit index carried across calls."""
    g = gold("rotator.npz")
    x = rotator_input()
    for name, f in (("p01", 0.01), ("m123", -0.123)):
        y = oracle.rotator(x, f, splits=(5, 40000, 65536))
        assert sha(y) == bytes(g[name + "_sha"]).hex()
        assert bits_equal(y[:64], g[name + "_head"]) and bits_equal(y[65530:65546], g[name + "_wrap"])

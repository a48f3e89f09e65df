"""Parity of the HIP fir_filter / elementwise kernels (through the C ABI) with
the oracle and the golden vectors.  Bit-exact in LSDR_FIR_EXACT mode; the
LSDR_FIR_FMA variant is held to a stated tolerance."""
import os
import numpy as np
import pytest
from conftest import gold, bits_equal, iq16_to_cf32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sig():
    rng = np.random.default_rng(3)
    n = 300000
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 40).astype(np.complex64)


def test_elementwise(capi, ctx, oracle, sig):
    rng = np.random.default_rng(1)
    for n in (0, 1, 3, 4, 5, 1000, 65537):
        u8 = rng.integers(0, 256, 2 * n, dtype=np.uint8)
        assert bits_equal(ctx.cconverter_u8(u8), oracle.cconverter_u8(u8))
        assert bits_equal(ctx.scaler(0.0123, sig[:n]), oracle.scaler(0.0123, sig[:n]))
    assert bits_equal(ctx.decimator(7, sig[:10001]), sig[:10001:7][:10001 // 7])


def test_golden_c2_geometry(capi, ctx):
    """fir_filter at the C2 geometry (N=313, D=30) incl. the fused scaler, vs the reference's output."""
    g, tab = gold("fir_filter.npz"), gold("tables.npz")
    x = iq16_to_cf32(g["iq120"])
    for tag, freq in [("f0", 0.0), ("fshift", 0.0123), ("fneg", -0.004)]:
        f = capi.FirFilter(ctx, tab["lowpass_c2"], 30, in_scale=float(g["scale"]))
        if freq:
            f.set_freq(freq)
        assert bits_equal(f.shifted_coeffs(), g[f"c2_{tag}_sc"])
        y, cons = f.run(x)
        assert bits_equal(y, g[f"c2_{tag}_out"]), tag
        assert cons == 30 * len(y)
        f.close()
    f = capi.FirFilter(ctx, tab["lowpass_small"], 2, in_format=capi.IN_CU8)   # fused cconverter
    y, _ = f.run(g["u8"])
    assert bits_equal(y, g["u8_d2_out"])
    f.close()


@pytest.mark.parametrize("arith", ["fma", "mfma", "blk"])
def test_golden_c2_geometry_tolerance_modes(capi, ctx, arith):
    """The tolerance arithmetics against the REFERENCE-produced fixture (not only against their own restatements): at the C2 geometry,
    real and shifted taps, every output within the stated bound 4e-6·Σ|c|·max|x| of the reference's — and the headline's
    kernel (k_fir_mfma_stream for blk at D=30) is what runs."""
    g, tab = gold("fir_filter.npz"), gold("tables.npz")
    x = iq16_to_cf32(g["iq120"])
    scale = float(g["scale"])
    c = tab["lowpass_c2"]
    bound = 4e-6 * float(np.abs(c).sum()) * float(np.abs(x).max()) * scale * np.sqrt(2)
    for tag, freq in [("f0", 0.0), ("fshift", 0.0123), ("fneg", -0.004)]:
        f = capi.FirFilter(ctx, c, 30, in_scale=scale, arith={"fma": capi.FIR_FMA, "mfma": capi.FIR_MFMA, "blk": capi.FIR_MFMA_BLK}[arith])
        if freq:
            f.set_freq(freq)
        y, cons = f.run(x)
        want = g[f"c2_{tag}_out"]
        assert len(y) == len(want) and cons == 30 * len(y)
        err = float(np.abs(y.astype(np.complex128) - want.astype(np.complex128)).max())
        assert err <= bound, (arith, tag, err, bound)
        assert err <= 1e-5 * float(np.abs(want).max()), (arith, tag, err)      # bench.py's in-run bound: 1e-5 of full scale
        f.close()


GEOMS = [(313, 30), (313, 1), (21, 1), (21, 7), (2, 3), (64, 64), (1, 1), (1, 5), (40, 4), (100, 10), (33, 16),
         (257, 8), (500, 30), (31, 31), (200, 2), (75, 5), (640, 64), (90, 9)]


@pytest.mark.parametrize("n,d", GEOMS, ids=[f"N{n}_D{d}" for n, d in GEOMS])
@pytest.mark.parametrize("freq", [0.0, 0.0123])
def test_vs_oracle(capi, ctx, oracle, sig, n, d, freq):
    c = oracle.lowpass(n - 1, np.float32(0.4 / d)) if n > 1 else np.array([0.75], np.float32)
    x = sig[: min(len(sig), 2000 * d + n + 17)]
    f = capi.FirFilter(ctx, c, d)
    if freq:
        f.set_freq(freq)
    y, cons = f.run(x)
    ref, rcons = oracle.fir_filter(c, d, x, freq)
    assert cons == rcons and bits_equal(y, ref)
    f.close()


@pytest.mark.parametrize("force", ["generic", "complex"])
def test_kernel_variants_agree(capi, ctx, oracle, sig, force):
    """Run-time-D kernel and complex-coefficient kernel on a real-coefficient filter: same bits."""
    env = {"generic": "LSDR_FIR_GENERIC", "complex": "LSDR_FIR_FORCE_COMPLEX"}[force]
    os.environ[env] = "1"
    try:
        c = oracle.lowpass(312, np.float32(0.0049))
        f = capi.FirFilter(ctx, c, 30)
        y, _ = f.run(sig[:100000])
        f.close()
    finally:
        del os.environ[env]
    ref, _ = oracle.fir_filter(c, 30, sig[:100000])
    assert bits_equal(y, ref)


def test_edges(capi, ctx, oracle, sig):
    c = oracle.lowpass(20, np.float32(0.2))
    f = capi.FirFilter(ctx, c, 4)
    for n in (0, 5, 20, 21, 24, 25, 28, 29):            # around ncoeffs and one decimation step
        y, cons = f.run(sig[:n])
        ref, rcons = oracle.fir_filter(c, 4, sig[:n])
        assert cons == rcons and bits_equal(y, ref), n
    # cap_out smaller than what the input allows: produce exactly cap, consume cap*decim
    x = sig[:1000]
    din = ctx.upload(x)
    dout = ctx.alloc(8 * 10)
    cons, prod = f.run_dev(din.ptr, len(x), dout.ptr, 10)
    assert (cons, prod) == (40, 10)
    ref, _ = oracle.fir_filter(c, 4, x)
    assert bits_equal(ctx.download(dout, np.complex64, 10), ref[:10])
    din.free(); dout.free(); f.close()


def test_streaming_equals_one_shot(capi, ctx, oracle, sig):
    """pipebuf-style use: consume what run() reports, keep the tail, feed more — same stream."""
    c = oracle.lowpass(312, np.float32(0.0049))
    f = capi.FirFilter(ctx, c, 30)
    x = sig[:200000]
    whole, _ = oracle.fir_filter(c, 30, x)
    outs, pos, avail = [], 0, 0
    for step in (5000, 313, 40000, 7, 100000, 1 << 30):
        avail = min(len(x), avail + step)
        y, cons = f.run(x[pos:avail])
        outs.append(y)
        pos += cons
    assert bits_equal(np.concatenate(outs), whole)
    f.close()


def test_track_semantics(capi, ctx):
    """freq_tap prologue of run(): re-shift only when |current-new| > tol (dsp.h:236-244)."""
    c = np.ones(9, np.float32) / 9
    f = capi.FirFilter(ctx, c, 1)
    assert not f.track(0.001, 0.5, 0.01) and f.current_freq == 0.0
    assert f.track(0.1, 0.5, 0.01) and abs(f.current_freq - 0.05) < 1e-9
    assert not f.track(0.11, 0.5, 0.01)
    f.close()


def test_fma_variant_tolerance(capi, ctx, oracle, sig):
    """LSDR_FIR_FMA: same order, fused ops.  Tolerance: |err| <= 4e-6 * sum|c|*max|x| (stated bound)."""
    c = oracle.lowpass(312, np.float32(0.0049))
    for freq in (0.0, 0.0123):
        f = capi.FirFilter(ctx, c, 30, arith=capi.FIR_FMA)
        if freq:
            f.set_freq(freq)
        y, _ = f.run(sig[:100000])
        ref, _ = oracle.fir_filter(c, 30, sig[:100000], freq)
        bound = 4e-6 * np.abs(c).sum() * np.abs(sig[:100000]).max() * 2
        assert np.abs(y - ref).max() <= bound
        f.close()


def test_large_full_config_properties(capi, ctx, oracle):
    """BASELINE config-2 size (N=313, D=30, 16 Mi samples): linearity-free exactness is checked
    on random windows against the oracle, plus the DC-gain property (sum of taps = 1)."""
    n = 1 << 24
    rng = np.random.default_rng(5)
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 30).astype(np.complex64)
    c = oracle.lowpass(312, np.float32((2e6 / 2) * (1 + 0.35 / 2) / 240e6))
    f = capi.FirFilter(ctx, c, 30)
    y, cons = f.run(x)
    assert len(y) == (n - 313) // 30 and cons == len(y) * 30
    for m0 in (0, 1234, len(y) // 2, len(y) - 700):
        ref, _ = oracle.fir_filter(c, 30, x[m0 * 30: m0 * 30 + 313 + 30 * 600 + 29])
        assert bits_equal(y[m0: m0 + len(ref)], ref)
    f.close()
    f = capi.FirFilter(ctx, c, 30)
    y, _ = f.run(np.full(40000, 3 - 2j, np.complex64))
    assert np.allclose(y, 3 - 2j, rtol=2e-5)
    f.close()


def test_run_multi_equals_separate_runs(capi, ctx, oracle):
    """lsdr_fir_filter_run_multi: three independent buffers in one launch == three run() calls (and == the oracle)."""
    rng = np.random.default_rng(11)
    n, D = 300000, 30
    coeffs = capi.lowpass(312, np.float32(0.0049))
    xs = [((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 20).astype(np.complex64) for _ in range(3)]
    f = capi.FirFilter(ctx, coeffs, D, in_scale=0.5)
    cap = (n - len(coeffs)) // D
    dins = [ctx.upload(x) for x in xs]
    douts = [ctx.alloc(cap * 8 + 64) for _ in xs]
    cons, prod = f.run_multi_dev([d.ptr for d in dins], n, [d.ptr for d in douts], cap)
    assert prod == cap and cons == cap * D
    for x, dout in zip(xs, douts):
        got = ctx.download(dout, np.complex64, prod)
        want, wcons = oracle.fir_filter(coeffs, D, oracle.scaler(0.5, x))
        assert wcons == cons and bits_equal(got, want[:prod])
    for d in dins + douts:
        d.free()
    f.close()


@pytest.mark.parametrize("d,nt", [(1, 9), (2, 23), (4, 40), (5, 61), (8, 65), (10, 101), (16, 161), (30, 313)])
def test_long_inputs_all_specialised_decimations(capi, ctx, oracle, d, nt):
    """A million samples through every compile-time-D kernel (R = 4 / 2 / 1 outputs per lane): many tiles per persistent
    workgroup, prefetch running ahead across tiles, the tail tile — bit for bit the oracle's output."""
    rng = np.random.default_rng(5)
    n = 4096 * 260
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 12).astype(np.complex64)
    co = capi.lowpass(nt - 1, 0.4 / d)
    f = capi.FirFilter(ctx, co, d)
    y, cons = f.run(x)
    f.close()
    ref, rcons = oracle.fir_filter(co, d, x)
    assert cons == rcons and bits_equal(y, ref)


MFMA_GEOMS = [(313, 30), (200, 30), (101, 10), (161, 16), (65, 8), (40, 4), (17, 16), (330, 30)]


@pytest.mark.parametrize("w", [2, 4])
@pytest.mark.parametrize("n,d", MFMA_GEOMS, ids=[f"N{n}_D{d}" for n, d in MFMA_GEOMS])
def test_mfma_is_the_fma_chain_bit_for_bit(capi, ctx, oracle, w, n, d):
    """LSDR_FIR_MFMA (k_fir_mfma: the taps as a banded Toeplitz block on v_mfma_f32_16x16x4_f32) computes LSDR_FIR_FMA's
    arithmetic — the reference's loop with fused multiply-adds, oracle.fir_filter(fma=True) — bit for bit: real taps and
    shifted (complex) taps, fused scaler, many tiles per persistent workgroup, ragged tail, the stream start."""
    rng = np.random.default_rng(n * 31 + d)
    ns = 4096 * 90 + 77
    x = ((rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) * 12).astype(np.complex64)
    co = capi.lowpass(n - 1, 0.4 / d)
    os.environ["LSDR_MFMA_W"] = str(w)
    try:
        for freq in (0.0, 0.0123):
            f = capi.FirFilter(ctx, co, d, in_scale=75.0, arith=capi.FIR_MFMA)
            g = capi.FirFilter(ctx, co, d, in_scale=75.0, arith=capi.FIR_FMA)
            if freq:
                f.set_freq(freq); g.set_freq(freq)
            y, cons = f.run(x)
            yg, _ = g.run(x)
            f.close(); g.close()
            ref, rcons = oracle.fir_filter(co, d, oracle.scaler(75.0, x), freq, fma=True)
            assert cons == rcons and len(y) == len(ref)
            assert np.array_equal(y, ref), (freq, int((y != ref).sum()), float(np.abs(y - ref).max()))
            assert np.array_equal(yg, ref), ("fma kernel", freq)
            exact, _ = oracle.fir_filter(co, d, oracle.scaler(75.0, x), freq)
            bound = 4e-6 * np.abs(co).sum() * np.abs(x).max() * 75 * 2
            assert np.abs(y - exact).max() <= bound
    finally:
        del os.environ["LSDR_MFMA_W"]


def test_mfma_run_multi_and_short_inputs(capi, ctx, oracle):
    """Several buffers per launch, inputs shorter than one tile, cap_out below what the input allows."""
    rng = np.random.default_rng(12)
    co = capi.lowpass(312, np.float32(0.0049))
    f = capi.FirFilter(ctx, co, 30, in_scale=0.5, arith=capi.FIR_MFMA)
    n = 250000
    xs = [((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 20).astype(np.complex64) for _ in range(3)]
    cap = (n - 313) // 30
    dins = [ctx.upload(x) for x in xs]
    douts = [ctx.alloc(cap * 8 + 64) for _ in xs]
    cons, prod = f.run_multi_dev([d.ptr for d in dins], n, [d.ptr for d in douts], cap)
    assert prod == cap and cons == cap * 30
    for x, dout in zip(xs, douts):
        want, _ = oracle.fir_filter(co, 30, oracle.scaler(0.5, x), fma=True)
        assert np.array_equal(ctx.download(dout, np.complex64, prod), want[:prod])
    for m in (0, 312, 313, 342, 343, 344, 1000, 4153, 8000):
        y, cons = f.run(xs[0][:m])
        want, wcons = oracle.fir_filter(co, 30, oracle.scaler(0.5, xs[0][:m]), fma=True)
        assert cons == wcons and np.array_equal(y, want), m
    cons, prod = f.run_dev(dins[0].ptr, n, douts[0].ptr, 1000)
    assert (cons, prod) == (30000, 1000)
    want, _ = oracle.fir_filter(co, 30, oracle.scaler(0.5, xs[0]), fma=True)
    assert np.array_equal(ctx.download(douts[0], np.complex64, 1000), want[:1000])
    for d in dins + douts:
        d.free()
    f.close()


BLK_GEOMS = [(313, 30), (200, 30), (330, 30), (480, 30), (30, 30), (101, 10), (160, 10), (161, 16), (65, 8), (40, 4), (17, 16), (7, 8),
             # the sweep (fir_stream_sweep.hip): every residue of D mod 8, odd D, padded rows of every size, both wave-tile shapes (D ≤ 34 / above),
             # the shapes leandvb.cc:353-378 derives (order ≈ 10.4·D … 12·D), the extremes of N (1, 16·D) and of D (2, 64)
             (21, 2), (32, 2), (1, 2), (37, 3), (48, 4), (53, 5), (67, 6), (81, 7), (89, 8), (97, 9), (115, 11), (127, 12), (150, 13), (161, 14),
             (170, 15), (200, 17), (230, 20), (233, 21), (260, 24), (270, 25), (352, 32), (347, 33), (362, 34), (371, 35), (410, 36), (440, 40),
             (495, 45), (530, 50), (633, 60), (960, 60), (660, 63), (676, 64), (1024, 64), (3, 64),
             # leandvb's own designs (eleven tap blocks: the sweep's compile-time form) at the remaining residues
             (43, 4), (63, 6), (73, 7), (85, 8), (157, 15), (167, 16), (209, 20), (251, 24), (105, 10),
             # no decimation at all (fir_filter with decim = 1: leandvb's --resample at Fs < 8·Fm)
             (1, 1), (11, 1), (16, 1)]


@pytest.mark.parametrize("kern", ["stream", "blk_w2", "blk_w4"])
@pytest.mark.parametrize("n,d", BLK_GEOMS, ids=[f"N{n}_D{d}" for n, d in BLK_GEOMS])
def test_mfma_blk_is_its_stated_arithmetic_bit_for_bit(capi, ctx, oracle, kern, n, d):
    """LSDR_FIR_MFMA_BLK (block-polyphase dense product on the matrix pipe: k_fir_mfma_stream — LDS-direct refill, the default,
    every decimation 1 … 64 — and k_fir_mfma_blk, register-staged, decimations 4, 8, 10, 16, 30) against its stated
    arithmetic, oracle.fir_filter(fma="blk") — the reference's loop with the taps in blocks of D, an fmaf chain per block,
    block sums added in order — bit for bit, and against the reference's arithmetic under LSDR_FIR_FMA's error bound."""
    if kern != "stream" and d not in (4, 8, 10, 16, 30):
        pytest.skip("k_fir_mfma_blk (register-staged) exists for decimations 4, 8, 10, 16, 30; k_fir_mfma_stream for 1 … 64")
    w = 4 if kern == "blk_w4" else 2
    os.environ["LSDR_MFMA_STREAM"] = "1" if kern == "stream" else "0"
    rng = np.random.default_rng(n * 37 + d)
    ns = 4096 * 90 + 77
    x = ((rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) * 12).astype(np.complex64)
    co = capi.lowpass(n - 1, 0.4 / d) if n > 1 else np.array([0.75], np.float32)
    os.environ["LSDR_MFMA_W"] = str(w)
    try:
        for freq in (0.0, 0.0123):
            f = capi.FirFilter(ctx, co, d, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
            if freq:
                f.set_freq(freq)
            y, cons = f.run(x)
            f.close()
            ref, rcons = oracle.fir_filter(co, d, x, freq, fma="blk", scale=75.0)
            assert cons == rcons and len(y) == len(ref)
            assert np.array_equal(y, ref), (freq, int((y != ref).sum()), int(np.flatnonzero(y != ref)[0]), float(np.abs(y - ref).max()))
            exact, _ = oracle.fir_filter(co, d, oracle.scaler(75.0, x), freq)
            bound = 4e-6 * np.abs(co).sum() * np.abs(x).max() * 75 * 2
            assert np.abs(y - exact).max() <= bound
    finally:
        del os.environ["LSDR_MFMA_W"]
        del os.environ["LSDR_MFMA_STREAM"]


@pytest.mark.parametrize("n,d", [(313, 30), (105, 10), (85, 8), (251, 24), (417, 40)])
def test_mfma_blk_run_time_and_compile_time_tap_blocks_give_the_same_bits(capi, ctx, oracle, n, d, monkeypatch):
    """Eleven tap blocks (leandvb's own designs) run k_fir_mfma_stream's compile-time-NQ form; LSDR_MFMA_NQT=0 forces the run-time-NQ
    kernel every other ncoeffs runs: same outputs, real and complex taps."""
    rng = np.random.default_rng(n + d)
    ns = 4096 * 40 + 31
    x = ((rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) * 9).astype(np.complex64)
    co = capi.lowpass(n - 1, 0.45 / d)
    outs = []
    for nqt in ("1", "0"):
        monkeypatch.setenv("LSDR_MFMA_NQT", nqt)
        for freq in (0.0, -0.021):
            f = capi.FirFilter(ctx, co, d, in_scale=3.0, arith=capi.FIR_MFMA_BLK)
            if freq:
                f.set_freq(freq)
            outs.append(f.run(x)[0])
            f.close()
    want = oracle.fir_filter(co, d, x, 0.0, fma="blk", scale=3.0)[0]
    assert np.array_equal(outs[0], want) and np.array_equal(outs[2], want)
    assert np.array_equal(outs[1], outs[3]) and np.array_equal(outs[1], oracle.fir_filter(co, d, x, -0.021, fma="blk", scale=3.0)[0])


@pytest.mark.parametrize("swpc", ["1", "48", "192"])
def test_mfma_blk_carried_ring_rows_do_not_depend_on_where_the_lists_are_cut(capi, ctx, oracle, swpc, monkeypatch):
    """Complex taps at the C2 geometry run the folded 64-row tiles that CARRY the ring's last rows into the next tile of a workgroup's list
    (fir_stream.h): no halo, a warm-up pair at the head of every list.  Output counts around one tile, two tiles, and many — with one
    workgroup per CU queued (long lists: the steady state), the default, and 192 (lists of one or two tiles: warm-ups everywhere) — and three
    streams in one launch (a list that crosses into the next stream): the oracle's bits each time."""
    monkeypatch.setenv("LSDR_MFMA_SWPC", swpc)
    rng = np.random.default_rng(int(swpc))
    co = capi.lowpass(312, 0.45 / 30)
    f = capi.FirFilter(ctx, co, 30, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
    f.set_freq(0.0123)
    for nout in (1, 53, 54, 55, 64, 65, 118, 129, 1100, 40000):
        ns = nout * 30 + len(co) + 11
        x = ((rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) * 0.7).astype(np.complex64)
        y, cons = f.run(x)
        want = oracle.fir_filter(co, 30, x, f.current_freq, fma="blk", scale=75.0)[0]
        assert len(y) == len(want) == nout and cons == 30 * nout and np.array_equal(y, want), nout
    n = 30 * 700 + len(co)
    xs = [((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64) for _ in range(3)]
    dins = [ctx.upload(x) for x in xs]
    douts = [ctx.alloc(700 * 8 + 64) for _ in xs]
    cons, prod = f.run_multi_dev([d.ptr for d in dins], n, [d.ptr for d in douts], 700)
    assert prod == 700 and cons == 700 * 30
    for x, dout in zip(xs, douts):
        assert np.array_equal(ctx.download(dout, np.complex64, prod), oracle.fir_filter(co, 30, x, f.current_freq, fma="blk", scale=75.0)[0][:prod])
    for d in dins + douts:
        d.free()
    f.close()


@pytest.mark.parametrize("n,d", [(81, 7), (343, 33), (16, 1), (625, 60)])
def test_mfma_blk_sweep_kernels_short_inputs_capped_outputs_and_run_multi(capi, ctx, oracle, n, d):
    """The sweep's kernels (padded LDS rows, 128- and 64-row wave tiles) at the edges: inputs shorter than one tile, one output, an output cap that
    cuts a tile, three streams in one launch — against lo_fir_filter_blk, bit for bit."""
    rng = np.random.default_rng(1000 * d + n)
    co = capi.lowpass(n - 1, 0.4 / d) if n > 1 else np.array([0.5], np.float32)
    f = capi.FirFilter(ctx, co, d, in_scale=0.5, arith=capi.FIR_MFMA_BLK)
    big = 70000 * d // 7 + 13
    xs = [((rng.standard_normal(big) + 1j * rng.standard_normal(big)) * 20).astype(np.complex64) for _ in range(3)]
    for m in (0, n - 1, n, n + d - 1, n + d, n + 17 * d + 3, 5000, big):
        m = min(max(m, 0), big)
        y, cons = f.run(xs[0][:m])
        want, wcons = oracle.fir_filter(co, d, xs[0][:m], fma="blk", scale=0.5)
        assert cons == wcons and np.array_equal(y, want), m
    cap = (big - n) // d
    dins = [ctx.upload(x) for x in xs]
    douts = [ctx.alloc(cap * 8 + 64) for _ in xs]
    cons, prod = f.run_multi_dev([q.ptr for q in dins], big, [q.ptr for q in douts], cap)
    assert prod == cap and cons == cap * d
    for x, dout in zip(xs, douts):
        want, _ = oracle.fir_filter(co, d, x, fma="blk", scale=0.5)
        assert np.array_equal(ctx.download(dout, np.complex64, prod), want[:prod])
    cons, prod = f.run_dev(dins[1].ptr, big, douts[1].ptr, 777)          # cap_out binds inside a tile
    assert (cons, prod) == (777 * d, 777)
    want, _ = oracle.fir_filter(co, d, xs[1], fma="blk", scale=0.5)
    assert np.array_equal(ctx.download(douts[1], np.complex64, 777), want[:777])
    f.set_freq(0.0371)                                                   # complex taps: the other kernel of the pair
    y, _ = f.run(xs[2][:20000])
    assert np.array_equal(y, oracle.fir_filter(co, d, xs[2][:20000], 0.0371, fma="blk", scale=0.5)[0])
    for q in dins + douts:
        q.free()
    f.close()


@pytest.mark.parametrize("stream", ["1", "0"])
def test_mfma_blk_run_multi_short_inputs_and_refusals(capi, ctx, oracle, stream, monkeypatch):
    monkeypatch.setenv("LSDR_MFMA_STREAM", stream)
    rng = np.random.default_rng(13)
    co = capi.lowpass(312, np.float32(0.0049))
    f = capi.FirFilter(ctx, co, 30, in_scale=0.5, arith=capi.FIR_MFMA_BLK)
    n = 250000
    xs = [((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 20).astype(np.complex64) for _ in range(3)]
    cap = (n - 313) // 30
    dins = [ctx.upload(x) for x in xs]
    douts = [ctx.alloc(cap * 8 + 64) for _ in xs]
    cons, prod = f.run_multi_dev([d.ptr for d in dins], n, [d.ptr for d in douts], cap)
    assert prod == cap and cons == cap * 30
    for x, dout in zip(xs, douts):
        want, _ = oracle.fir_filter(co, 30, x, fma="blk", scale=0.5)
        assert np.array_equal(ctx.download(dout, np.complex64, prod), want[:prod])
    for m in (0, 312, 313, 342, 343, 344, 1000, 4153, 8000):
        y, cons = f.run(xs[0][:m])
        want, wcons = oracle.fir_filter(co, 30, xs[0][:m], fma="blk", scale=0.5)
        assert cons == wcons and np.array_equal(y, want), m
    cons, prod = f.run_dev(dins[0].ptr, n, douts[0].ptr, 1000)
    assert (cons, prod) == (30000, 1000)
    want, _ = oracle.fir_filter(co, 30, xs[0], fma="blk", scale=0.5)
    assert np.array_equal(ctx.download(douts[0], np.complex64, 1000), want[:1000])
    for d in dins + douts:
        d.free()
    f.close()
    # its own arithmetic: no other kernel has the same bits, so geometries without a kernel are refused, not re-routed
    for nn, dd, fmt in ((313, 65, capi.IN_CF32), (17, 1, capi.IN_CF32), (600, 30, capi.IN_CF32), (313, 30, capi.IN_CU8)) + (((81, 7, capi.IN_CF32),) if stream == "0" else ()):
        with pytest.raises(Exception):
            capi.FirFilter(ctx, capi.lowpass(nn - 1, 0.01), dd, in_format=fmt, arith=capi.FIR_MFMA_BLK)

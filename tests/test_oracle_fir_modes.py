"""(CPU) The oracle's restatements of the tolerance arithmetics of the HIP fir_filter — lo_fir_filter_fma (LSDR_FIR_FMA / LSDR_FIR_MFMA:
the reference's loop, dsp.h:246-262, with fused multiply-adds) and lo_fir_filter_blk (LSDR_FIR_MFMA_BLK: the taps in blocks of `decim`,
an fmaf chain per block, block sums added in order, scaler on the taps) — against the exact restatement (itself pinned to the
reference build in test_oracle_vs_ref.py): identical where the arithmetic is exact anyway, one block = one chain, and within the
error bound the GPU tests assert for the kernels."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def sig():
    rng = np.random.default_rng(7)
    n = 60000
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 40).astype(np.complex64)


def test_exact_arithmetic_inputs_agree_bit_for_bit(oracle):
    """Small integers times powers of two: every product and sum is exact in f32, so rounding order cannot matter."""
    rng = np.random.default_rng(1)
    x = (rng.integers(-64, 64, 5000) + 1j * rng.integers(-64, 64, 5000)).astype(np.complex64)
    c = (2.0 ** rng.integers(-6, 3, 45)).astype(np.float32) * rng.choice([-1, 1], 45).astype(np.float32)
    for d in (1, 4, 10, 45):
        a, ca = oracle.fir_filter(c, d, x)
        b, cb = oracle.fir_filter(c, d, x, fma=True)
        e, ce = oracle.fir_filter(c, d, x, fma="blk")
        s, _ = oracle.fir_filter(c, d, x, fma="blk", scale=4.0)
        assert ca == cb == ce and np.array_equal(a, b) and np.array_equal(a, e)
        assert np.array_equal(s, oracle.fir_filter(c, d, oracle.scaler(4.0, x))[0])


def test_one_block_is_one_chain(oracle, sig):
    """decim ≥ ncoeffs, real taps: the blocked form has a single block — the fmaf chain of lo_fir_filter_fma.  (Complex taps: the chain
    takes its taps four at a time — next test.)"""
    c = oracle.lowpass(30, np.float32(0.1))
    for d in (31, 40):
        a, _ = oracle.fir_filter(c, d, sig, 0.0, fma=True)
        b, _ = oracle.fir_filter(c, d, sig, 0.0, fma="blk")
        assert np.array_equal(a, b)


def _fma32(a, b, c):
    """fmaf on float32 values through the 64-bit mantissa of np.longdouble: the product is exact, the sum is rounded once to 64 bits and
    once to 24 (a double rounding needs a tie pattern 40 bits long: not on random data)."""
    return np.float32(np.longdouble(a) * np.longdouble(b) + np.longdouble(c))


@pytest.mark.skipif(np.finfo(np.longdouble).nmant < 63, reason="needs an 80-bit long double")
@pytest.mark.parametrize("n,d", [(45, 10), (23, 30), (9, 4)])
def test_complex_taps_go_through_a_block_four_at_a_time(oracle, sig, n, d):
    """The stated order of a block's chain under complex taps (k_fir_mfma_stream's four MFMAs per step): the re-part products of four
    taps, then their im-part products; blocks added in order."""
    c = oracle.lowpass(n - 1, np.float32(0.4 / d))
    sc = oracle.fir_shift(c, 0.0123)
    x = sig[:n + 40 * d]
    y, _ = oracle.fir_filter(c, d, x, 0.0123, fma="blk")
    for m in range(len(y)):
        p0 = n + m * d
        acc = None
        for q0 in range(0, n, d):
            zr = zi = np.float32(0)
            i1 = min(q0 + d, n)
            for g in range(q0, i1, 4):
                idx = range(g, min(g + 4, i1))
                for i in idx:
                    zr = _fma32(sc[i].real, x[p0 - i].real, zr); zi = _fma32(sc[i].real, x[p0 - i].imag, zi)
                for i in idx:
                    zr = _fma32(-sc[i].imag, x[p0 - i].imag, zr); zi = _fma32(sc[i].imag, x[p0 - i].real, zi)
            acc = (zr, zi) if acc is None else (np.float32(acc[0] + zr), np.float32(acc[1] + zi))
        assert acc[0] == y[m].real and acc[1] == y[m].imag, m


def test_exact_arithmetic_complex_taps_agree_bit_for_bit(oracle):
    """Complex taps of small integers times powers of two: rounding order cannot matter, so the grouped chain equals the reference loop."""
    rng = np.random.default_rng(2)
    x = (rng.integers(-64, 64, 4000) + 1j * rng.integers(-64, 64, 4000)).astype(np.complex64)
    sc = ((2.0 ** rng.integers(-4, 3, 45)) * rng.choice([-1, 1], 45) + 1j * (2.0 ** rng.integers(-4, 3, 45)) * rng.choice([-1, 1], 45)).astype(np.complex64)
    for d in (1, 4, 10, 45):
        a, _ = oracle.fir_filter(np.zeros(45, np.float32), d, x, shifted=sc)
        e, _ = oracle.fir_filter(np.zeros(45, np.float32), d, x, shifted=sc, fma="blk")
        assert np.array_equal(a, e)


@pytest.mark.parametrize("n,d", [(313, 30), (101, 10), (65, 8), (17, 16)])
def test_within_the_stated_error_bound_of_the_reference_arithmetic(oracle, sig, n, d):
    c = oracle.lowpass(n - 1, np.float32(0.4 / d))
    for freq in (0.0, 0.0123):
        exact, _ = oracle.fir_filter(c, d, oracle.scaler(75.0, sig), freq)
        bound = 4e-6 * np.abs(c).sum() * np.abs(sig).max() * 75 * 2
        fma, _ = oracle.fir_filter(c, d, oracle.scaler(75.0, sig), freq, fma=True)
        blk, _ = oracle.fir_filter(c, d, sig, freq, fma="blk", scale=75.0)
        assert len(fma) == len(blk) == len(exact)
        assert 0 < np.abs(fma - exact).max() <= bound and 0 < np.abs(blk - exact).max() <= bound
        # and against float64: all three are f32-roundoff-class evaluations of the same sum
        sc = oracle.fir_shift(c, freq).astype(np.complex128)
        x64 = sig.astype(np.complex128) * 75.0
        m = np.arange(0, len(exact), max(1, len(exact) // 200))
        ref = np.array([np.dot(sc, x64[n + k * d - np.arange(n)]) for k in m])
        for y in (exact, fma, blk):
            assert np.abs(y[m] - ref).max() <= bound

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
    """decim ≥ ncoeffs: the blocked form has a single block — the fmaf chain of lo_fir_filter_fma."""
    c = oracle.lowpass(30, np.float32(0.1))
    for freq in (0.0, 0.0123):
        a, _ = oracle.fir_filter(c, 31, sig, freq, fma=True)
        b, _ = oracle.fir_filter(c, 31, sig, freq, fma="blk")
        assert np.array_equal(a, b)
        a, _ = oracle.fir_filter(c, 40, sig, freq, fma=True)
        b, _ = oracle.fir_filter(c, 40, sig, freq, fma="blk")
        assert np.array_equal(a, b)


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

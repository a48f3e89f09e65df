"""Shared helpers for the FEC-tail tests: the deterministic soft-symbol input of the goldens
(same integer formulas as oracle/make_golden.py:fec_input)."""
import hashlib
import numpy as np
from conftest import gold

SOFTSYM = np.dtype([("cost", "<i2"), ("symbol", "u1"), ("pad", "u1")])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def hard_symbols():
    g = gold("fec.npz")
    n = int(g["nsym"])
    bits = np.unpackbits(g["hard_packed"])[: 2 * n].reshape(-1, 2)
    return (bits[:, 0] * 2 + bits[:, 1]).astype(np.uint8)


def fec_input(hard, err_permille):
    n = len(hard)
    i = np.arange(n, dtype=np.uint64)
    sym = np.zeros(n, SOFTSYM)
    sym["symbol"] = hard
    sym["cost"] = -(1000 + (i * np.uint64(37)) % np.uint64(10237)).astype(np.int64)
    if err_permille:
        h = (i * np.uint64(2654435761)) % np.uint64(1 << 32)
        bad = h < np.uint64((err_permille << 32) // 1000)
        sym["symbol"][bad] = (hard[bad] ^ (1 + (h[bad] >> np.uint64(7)) % np.uint64(3)).astype(np.uint8)) & 3
        sym["cost"][bad] = -(h[bad] % np.uint64(300)).astype(np.int64)
    return sym


CASES = [("clean", 0), ("noisy", 40)]

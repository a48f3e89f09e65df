"""Channel-simulator oracle (oracle/lsdr_oracle_chan.c) against the output of the real `leanchansim` binary and the reference's
wgn_c / drand48 / logf (tests/golden/chan.npz) and, where the reference build is present, against oracle/_ref directly."""
import hashlib
import os
import numpy as np
import pytest
import pyoracle as po
from conftest import gold, bits_equal


def test_drand48_and_logf_golden(oracle):
    g = gold("chan.npz")
    d, _ = oracle.drand48(100000)
    assert np.array_equal(d[[0, 1, 2, 3, 1000, 99999]], g["drand48"])
    d, _ = oracle.drand48(3, seed=1234)
    assert np.array_equal(d, g["drand48_seed1234"])
    assert bits_equal(oracle.logf(g["logf_x"]), g["logf_y"])


def test_wgn_golden(oracle):
    g = gold("chan.npz")
    w, _ = oracle.wgn(200000, 0.7)
    assert bits_equal(w[:256], g["wgn_head"]) and hashlib.sha256(w.tobytes()).digest() == bytes(g["wgn_sha"])
    w, _ = oracle.wgn(5000, 2.0, seed=77)
    assert bits_equal(w[:256], g["wgn77_head"]) and hashlib.sha256(w.tobytes()).digest() == bytes(g["wgn77_sha"])
    # the stream does not depend on how it is cut into calls
    a, st = oracle.wgn(1234, 0.7)
    b, _ = oracle.wgn(777, 0.7, state=st)
    assert bits_equal(np.concatenate([a, b]), w0 := oracle.wgn(2011, 0.7)[0]) and len(w0) == 2011


@pytest.mark.parametrize("name,args,kw", po.CHAN_CASES)
def test_chansim_is_leanchansim(oracle, name, args, kw):
    g = gold("chan.npz")
    y = oracle.chansim(po.chan_test_input(), **kw).reshape(-1)
    assert len(y) == int(g[name + "_n"])
    assert bits_equal(y[:512], g[name + "_head"]) and bits_equal(y[-512:], g[name + "_tail"])
    assert hashlib.sha256(y.tobytes()).digest() == bytes(g[name + "_sha"])


def test_chan_blocks_vs_ref(oracle, ref):
    w, _ = oracle.wgn(50000, 1.3, seed=5)
    assert bits_equal(w, ref.wgn(50000, 1.3, seed=5))
    x = po.chan_test_input(20000)
    assert bits_equal(oracle.adder(x, x[::-1].copy()), ref.adder(x, x[::-1].copy()))
    with np.errstate(all="ignore"):
        x3 = x * 3
    assert np.array_equal(oracle.cconv_f32_u8(x3), ref.cconv_f32_u8(x3))
    # logf: a slice of the domain by default, every float in (0,1) with LSDR_EXHAUSTIVE=1 (≈ 15 s)
    lo, hi = (0x00800000, 0x3f800000) if os.environ.get("LSDR_EXHAUSTIVE") else (0x3e000000, 0x3e400000)
    assert ref.logf_mismatches(oracle, lo, hi)[0] == 0

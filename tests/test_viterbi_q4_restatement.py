"""(CPU) The arithmetic of k_viterbi_q4 (leansdr_amd/csrc/viterbi.hip), restated in numpy and run against the oracle's
viterbi_sync: metrics kept x16 with the tie rule in the four low bits (the number of branches into the state with a higher
coded symbol, or NUS on the labelled branch when its cost is negative; a positive cost clamped to 0), a strict minimum, the low
bits cleared afterwards, lowest-index best state, register-exchange paths.  Inputs with many exact ties (costs 0 / -1 / -2),
positive costs and int16 extremes: the claim that this IS viterbi_dec::update's choice (viterbi.h:202-260) does not need a GPU."""
import ctypes as C
import numpy as np
import pytest

G1, G2 = 0o171, 0o133


def parity(x):
    return bin(x).count("1") & 1


def trellis(nus):
    """(pred[s'][e], label[s'][e], us[s']) as the kernel's constexpr functions give them (q4::branch_label, q4::state_us)."""
    sh = {2: 1, 4: 2}[nus]
    pred = np.zeros((64, nus), np.int64); lab = np.zeros((64, nus), np.int64); us = np.zeros(64, np.int64)
    for s in range(64):
        lowmask = (64 >> sh) - 1
        for e in range(nus):
            reg = ((s & lowmask) << sh) | e | ((s >> (6 - sh)) << 6)
            if nus == 2:
                cs = (parity(reg & G1) << 1) | parity(reg & G2)
            else:
                cs = (parity(reg & G1) << 2) | (parity(reg & G2) << 1) | parity(reg & (G2 << 1))
            pred[s, e] = (s & lowmask) * nus + e
            lab[s, e] = cs
        top = s >> (6 - sh)
        us[s] = top if nus == 2 else ((top & 1) << 1) | (top >> 1)
    return pred, lab, us


def q4_decode(cs_seq, cost_seq, nus):
    """The kernel's recurrence for one alignment, one symbol per FEC block; returns the decoded input symbols per step."""
    pred, lab, us = trellis(nus)
    ncs, nbits, depth = 2 * nus, {2: 1, 4: 3}[nus], {2: 32, 4: 21}[nus]
    le = [int(lab[0, e]) for e in range(nus)]
    tie = np.array([sum(1 for e in range(1, nus) if (y ^ le[e]) > y) for y in range(ncs)], np.int64)
    c = np.zeros(64, np.int64)
    path = np.zeros(64, np.uint64)
    mask = np.uint64((1 << (depth * nbits)) - 1) if depth * nbits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    out = np.empty(len(cs_seq), np.int64)
    usv = us.astype(np.uint64)
    for t in range(len(cs_seq)):
        cs, cost = int(cs_seq[t]), min(int(cost_seq[t]), 0)
        A = tie.copy()
        if cost < 0 and cs < ncs:
            A[cs] = 16 * cost + nus
        cand = c[pred] + A[lab]                                  # [64][nus]
        assert all(len(set(row)) == nus for row in cand[:4])      # no two candidates of a state are ever equal
        k = np.argmin(cand, axis=1)
        best = cand[np.arange(64), k]
        c = best & ~np.int64(15)
        p = path[pred[np.arange(64), k]]
        path = ((p << np.uint64(nbits)) | usv) & mask
        b = int(np.argmin(c))                                     # lowest index among the minima
        out[t] = int(path[b] >> np.uint64((depth - 1) * nbits)) & ((1 << nbits) - 1)
        if (t & 127) == 127:
            c -= c.min()
    return out


def encode(bits_in, nus, n_steps, rng):
    """Random input symbols through the convolutional encoder (trellis::init_convolutional's shift register): coded symbols."""
    sh = {2: 1, 4: 2}[nus]
    s, cs_seq = 0, np.empty(n_steps, np.int64)
    for t in range(n_steps):
        u = int(rng.integers(0, nus))
        rev = int("{:0{w}b}".format(u, w=sh)[::-1], 2)
        reg = s | (rev << 6)
        if nus == 2:
            cs = (parity(reg & G1) << 1) | parity(reg & G2)
        else:
            cs = (parity(reg & G1) << 2) | (parity(reg & G2) << 1) | parity(reg & (G2 << 1))
        cs_seq[t] = cs
        s = reg >> sh
    return cs_seq


@pytest.mark.parametrize("cstln,rate,nus", [(1, 0, 2), (2, 1, 4)])
@pytest.mark.parametrize("costs", ["ties", "int16", "positive"])
def test_q4_arithmetic_is_the_reference_decoder(oracle, cstln, rate, nus, costs):
    rng = np.random.default_rng(7 + nus)
    n = 128 * 48
    cs_seq = encode(None, nus, n, rng)
    # symbol -> coded symbol map of alignment 0 (init_map, dvb.h:1336-1351), inverted to place coded symbols on the constellation
    h = oracle.lib.lo_viterbi_new(cstln, rate)
    m = np.zeros(256, np.uint8)
    oracle.lib.lo_viterbi_map(h, 0, m.ctypes.data)
    oracle.lib.lo_viterbi_free(h)
    nsym = {1: 4, 2: 8}[cstln]
    inv = {int(m[i]): i for i in range(nsym)}
    assert len(inv) == nsym
    sym = np.zeros(n, [("cost", "<i2"), ("symbol", "u1"), ("pad", "u1")])
    sym["symbol"] = [inv[int(c)] for c in cs_seq]
    flip = rng.random(n) < 0.03                                   # a few channel errors
    sym["symbol"][flip] = rng.integers(0, nsym, int(flip.sum()))
    if costs == "ties":
        sym["cost"] = -rng.integers(0, 3, n)
    elif costs == "int16":
        sym["cost"] = np.where(rng.random(n) < 0.5, -32768, -rng.integers(0, 32768, n))
    else:
        c = -rng.integers(0, 400, n); c[::5] = rng.integers(1, 300, len(c[::5]))
        sym["cost"] = c
    want, cons, cur = oracle.viterbi_sync(sym, cstln, rate)
    assert cur == 0 and cons == n                                  # the stream stays on alignment 0: one decoder, no switch
    cs_rx = m[sym["symbol"]].astype(np.int64)
    got = q4_decode(cs_rx, sym["cost"].astype(np.int64), nus)
    bits_in = {2: 1, 4: 2}[nus]
    bits = np.zeros(n * bits_in, np.uint8)
    for b in range(bits_in):
        bits[b::bits_in] = (got >> (bits_in - 1 - b)) & 1
    assert np.packbits(bits).tobytes() == np.asarray(want, np.uint8).tobytes()

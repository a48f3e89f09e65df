"""leansdr_amd/synth_dvbs.py — framed DVB-S (EN 300 421) test-signal generator, numpy only.

The build's own equivalent of `leantsgen | leandvbtx | leanchansim`: counter-pattern TS
packets → energy dispersal → RS(204,188) → Forney interleaver (I=12, M=17) → K=7 rate-1/2 convolutional
code (G1=171, G2=133 octal) → QPSK (Gray, EN 300 421 fig. 5) → RRC(0.35) at sps = num/den → AWGN →
cf32 or offset-128 cu8.  It is a signal SOURCE for tests and bench.py, not part of the timed path;
tests check that the reference `leandvb` decodes what it produces.
"""
import numpy as np
from scipy import signal

from .synth import rrc_taps


def ts_packets(n, start=0):
    """leantsgen's pattern (leantsgen.cc:37-49): byte 4k = 4k, then a 24-bit packet counter; byte 0 = 0x47."""
    t = np.arange(start, start + n, dtype=np.uint32)
    pk = np.zeros((n, 188), np.uint8)
    pk[:, 0::4] = np.arange(0, 188, 4, dtype=np.uint8)
    pk[:, 1::4] = (t >> 16).astype(np.uint8)[:, None]
    pk[:, 2::4] = (t >> 8).astype(np.uint8)[:, None]
    pk[:, 3::4] = t.astype(np.uint8)[:, None]
    pk[:, 0] = 0x47
    return pk


def dispersal_pattern():
    """PRBS 1+x^14+x^15, init 100101010000000, restarted every 8 packets; byte 0 flips the first sync,
    the generator keeps running but is not applied on the other seven sync bytes."""
    pat = np.zeros(1504, np.uint8)
    pat[0] = 0xff
    st = 0o000251
    for i in range(1, 1504):
        o = 0
        for _ in range(8):
            bit = ((st >> 13) ^ (st >> 14)) & 1
            o = ((o << 1) | bit) & 0xff
            st = ((st << 1) | bit) & 0xffff
        pat[i] = o if i % 188 else 0
    return pat


def _gf():
    exp = np.zeros(512, np.int32)
    log = np.zeros(256, np.int32)
    a = 1
    for i in range(255):
        exp[i] = a
        log[a] = i
        a <<= 1
        if a & 256:
            a ^= 0x11d
    exp[255:510] = exp[:255]
    return exp, log


def rs_encode(pk188):
    """Systematic RS(204,188) over GF(256)/0x11d, generator Π_{i=0..15}(x − α^i); vectorised over packets."""
    exp, log = _gf()
    g = np.array([1], np.int32)
    for d in range(16):   # multiply by (x + α^d), coefficients highest degree first
        ad = exp[d]
        ng = np.zeros(len(g) + 1, np.int32)
        ng[:-1] ^= g
        ng[1:] ^= np.where(g == 0, 0, exp[(log[g] + log[ad]) % 255])
        g = ng
    n = len(pk188)
    rem = np.zeros((n, 16), np.int32)
    glog = log[g[1:]]
    for k in range(188):
        fb = pk188[:, k].astype(np.int32) ^ rem[:, 0]
        rem[:, :-1] = rem[:, 1:]
        rem[:, -1] = 0
        nz = fb != 0
        term = exp[(log[fb[nz]][:, None] + glog[None, :]) % 255]
        rem[nz] ^= term
    return np.concatenate([pk188, rem.astype(np.uint8)], axis=1)


def interleave(bytes_):
    """Forney convolutional interleaver: byte n (branch n mod 12, sync bytes on branch 0) is delayed by
    17·12·(n mod 12) positions; the pipeline starts filled with zeros."""
    n = len(bytes_)
    idx = np.arange(n)
    out = np.zeros(n + 17 * 12 * 11, np.uint8)
    out[idx + 17 * 12 * (idx % 12)] = bytes_
    return out[:n]


def conv_encode(bytes_):
    """K=7 rate 1/2, X = G1 = 171o, Y = G2 = 133o, bits MSB first, newest bit at register bit 6."""
    bits = np.unpackbits(bytes_)
    pad = np.concatenate([np.zeros(6, np.uint8), bits])
    x = np.zeros(len(bits), np.uint8)
    y = np.zeros(len(bits), np.uint8)
    for tap in range(7):   # register bit b holds the input bit (6 − b) steps back
        sel = pad[6 - (6 - tap): 6 - (6 - tap) + len(bits)] if False else pad[tap: tap + len(bits)]
        # pad[tap + t] = bits[t − 6 + tap] → register bit `tap` at time t
        if (0o171 >> tap) & 1:
            x ^= sel
        if (0o133 >> tap) & 1:
            y ^= sel
    return x, y


def modulate(ts, sps_num=6, sps_den=5, rolloff=0.35):
    """TS packets → complex baseband (unit average power) at sps_num/sps_den samples per symbol."""
    pat = dispersal_pattern()
    n = len(ts)
    rand = ts ^ np.tile(pat.reshape(8, 188), (n // 8 + 1, 1))[:n]
    rs = rs_encode(rand)
    il = interleave(rs.reshape(-1))
    x, y = conv_encode(il)
    sym = ((1.0 - 2.0 * x) + 1j * (1.0 - 2.0 * y)) / np.sqrt(2.0)
    h = rrc_taps(sps_num, rolloff, span=10)
    bb = signal.upfirdn(h, sym, up=sps_num, down=sps_den)
    return bb * np.sqrt(sps_num)   # unit average power (the decimation keeps the power)


def capture_u8(n_packets=140, sps_num=6, sps_den=5, seed=1, amp=75.0, noise_std=7.5):
    """Config-1 shaped capture: offset-128 cu8, RMS amplitude `amp`, AWGN `noise_std` per component
    (leandvbtx --power 37.5 | leanchansim --awgn 17.5 --ou8).  Returns (uint8 IQ array, TS packets)."""
    assert n_packets % 8 == 0 or True
    ts = ts_packets(n_packets)
    bb = modulate(ts, sps_num, sps_den)
    rng = np.random.default_rng(seed)
    x = bb * amp + (rng.standard_normal(len(bb)) + 1j * rng.standard_normal(len(bb))) * noise_std
    iq = np.empty(2 * len(x), np.uint8)
    iq[0::2] = np.clip(np.rint(x.real + 128), 0, 255)
    iq[1::2] = np.clip(np.rint(x.imag + 128), 0, 255)
    return iq, ts


def capture_f32(n_packets=140, sps=4, seed=1, rms=1.0, snr_db=20.0):
    """cf32 capture at an integer number of samples per symbol (Es/N0 as in synth.qpsk_baseband)."""
    ts = ts_packets(n_packets)
    bb = modulate(ts, sps, 1)
    rng = np.random.default_rng(seed)
    nstd = np.sqrt(0.5 * sps / (10 ** (snr_db / 10)))
    x = bb + (rng.standard_normal(len(bb)) + 1j * rng.standard_normal(len(bb))) * nstd
    x *= rms / np.sqrt(1 + 2 * nstd ** 2)
    return x.astype(np.complex64), ts

"""leansdr_amd/synth.py — deterministic synthetic QPSK baseband for bench.py and tests.

The build's own equivalent of the reference's generator pipeline
`leantsgen | leandvbtx -f I/D --agc | leanchansim --awgn`:
random QPSK symbols, root-raised-cosine pulse shaping (roll-off 0.35) at
`sps` samples/symbol, complex AWGN, fixed RMS amplitude.  numpy/scipy on the
host; it is a signal SOURCE, not part of the timed path.  The payload here is
random bits (enough for the fir_filter + cstln_receiver workload, BASELINE
config 2); the framed DVB-S generator for the TS-exact configs lives in
synth_dvbs (added with the FEC tail).
"""
import numpy as np
from scipy import signal


def rrc_taps(sps, rolloff=0.35, span=10):
    """Continuous-time RRC sampled at `sps` samples/symbol, `span` symbols each side, unit energy."""
    n = np.arange(-span * sps, span * sps + 1, dtype=np.float64)
    t = n / sps
    b = rolloff
    with np.errstate(divide="ignore", invalid="ignore"):
        num = np.sin(np.pi * t * (1 - b)) + 4 * b * t * np.cos(np.pi * t * (1 + b))
        den = np.pi * t * (1 - (4 * b * t) ** 2)
        h = num / den
    h[n == 0] = 1 - b + 4 * b / np.pi
    sing = np.isclose(np.abs(4 * b * t), 1.0)
    h[sing] = b / np.sqrt(2) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * b)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * b)))
    return h / np.sqrt((h ** 2).sum())


def qpsk_baseband(n_samples, sps, seed=1, rms=1.0, snr_db=20.0, rolloff=0.35, circular=True, freq=0.0):
    """cf32 array of n_samples (n_samples % sps == 0 when circular): QPSK at sps samples/symbol.

    circular=True shapes the pulse train circularly so the buffer can be replayed
    back-to-back as an endless stream (used by bench.py).  snr_db is Es/N0 in the
    symbol-rate bandwidth.  Returns (samples, symbols) with symbols in {0..3}
    using the reference's QPSK labelling (sdr.h:340-347: 0:(+,+) 1:(+,-) 2:(-,+) 3:(-,-)).
    """
    rng = np.random.default_rng(seed)
    nsym = n_samples // sps
    assert nsym * sps == n_samples or not circular
    lab = rng.integers(0, 4, nsym + (0 if circular else 32))
    pts = np.array([1 + 1j, 1 - 1j, -1 + 1j, -1 - 1j]) / np.sqrt(2)
    s = pts[lab]
    h = rrc_taps(sps, rolloff)
    if circular:
        ext = 10
        se = np.concatenate([s[-ext:], s, s[:ext]])
        y = signal.upfirdn(h, se, up=sps)
        start = ext * sps + (len(h) - 1) // 2
        y = y[start:start + n_samples]
    else:
        y = signal.upfirdn(h, s, up=sps)[(len(h) - 1) // 2:][:n_samples]
    y = y * np.sqrt(sps)                       # unit average power
    nstd = np.sqrt(0.5 * sps / (10 ** (snr_db / 10)))   # noise power sps/(Es/N0) spread over the full band
    y = y + (rng.standard_normal(len(y)) + 1j * rng.standard_normal(len(y))) * nstd
    if freq:
        y = y * np.exp(2j * np.pi * freq * np.arange(len(y)))
    y *= rms / np.sqrt(1 + 2 * nstd ** 2)
    return y.astype(np.complex64), lab[:nsym]

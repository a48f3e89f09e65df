"""oracle/pyoracle.py — ctypes bindings for the two CPU checkers.  TEST INFRASTRUCTURE.

  * `Oracle`  -> oracle/liblsdr_oracle.so   (our plain-C restatement, travels)
  * `Ref`     -> oracle/_ref/libleansdr_ref.so (the real reference behind
                 oracle/ref_harness.cc; exists only where it was built)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liblsdr_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libleansdr_ref.so")
REF_DIR = os.path.join(HERE, "_ref")

c_f = C.c_float
c_fp = C.POINTER(C.c_float)
c_sz = C.c_size_t


def build(ref=True):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liblsdr_oracle.so"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def cf32(a):
    """complex64 ndarray (contiguous) view helper."""
    return np.ascontiguousarray(a, dtype=np.complex64)


SOFTSYM = np.dtype([("cost", "<i2"), ("symbol", "u1"), ("pad", "u1")])


class RxParams(C.Structure):
    _fields_ = [("sampler", C.c_int), ("ncoeffs", C.c_int), ("coeffs", C.c_void_p),
                ("subsampling", C.c_int), ("cstln", C.c_int), ("fec", C.c_int),
                ("omega", c_f), ("freq", c_f), ("pll_adjustment", c_f),
                ("allow_drift", C.c_int), ("meas_decimation", C.c_ulong), ("kest", c_f)]


class RxState(C.Structure):
    _fields_ = [("mu", c_f), ("phase", c_f), ("freqw", c_f), ("agc_gain", c_f),
                ("est_insp", c_f), ("est_sp", c_f), ("est_ep", c_f), ("freq_tap", c_f),
                ("min_freqw", c_f), ("max_freqw", c_f), ("meas_count", C.c_ulong),
                ("hist", c_f * 12)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "hist"}
        d["hist"] = list(self.hist)
        return d


def rx_params(sampler=1, coeffs=None, subsampling=1, cstln=1, fec=0, omega=4.0, freq=0.0,
              pll_adjustment=1.0, allow_drift=0, meas_decimation=1048576, kest=0.01):
    p = RxParams()
    p.sampler = sampler
    if coeffs is not None:
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float32)
        p._keep = coeffs
        p.ncoeffs = len(coeffs)
        p.coeffs = coeffs.ctypes.data
    p.subsampling = subsampling
    p.cstln, p.fec = cstln, fec
    p.omega, p.freq, p.pll_adjustment = omega, freq, pll_adjustment
    p.allow_drift, p.meas_decimation, p.kest = allow_drift, meas_decimation, kest
    return p


class CstlnLut(C.Structure):
    _fields_ = [("nsymbols", C.c_int), ("nrotations", C.c_int),
                ("symbols", (C.c_int8 * 2) * 256),
                ("cost", C.c_int16 * 65536), ("symbol", C.c_uint8 * 65536),
                ("phase_error", C.c_int16 * 65536)]


def chan_test_input(n=50000):
    """Deterministic cf32 test signal for the channel-simulator fixtures (exact small integers/halves, plus the float → u8
    corner cases of cconverter<f32,0,u8,128,1,1> in its first samples)."""
    i = np.arange(n, dtype=np.int64)
    x = (((i * 37) % 251 - 125) * 0.5 + 1j * (((i * 91) % 241 - 120) * 0.5)).astype(np.complex64)
    x[:8] = np.array([1e10, -1e10, 3e9 + 3e9j, np.nan, np.inf, -129.5, 127.9, -0.0], np.complex64)
    return x


CHAN_CASES = [("awgn", "--awgn -3 --deterministic", dict(awgn_db=-3.0)),
              ("drift_u8", "--awgn 2.5 --deterministic -f 2e6 --lo 10e9 --ppm 3 --drift-period 0.01 --drift2-amp 500 --drift2-freq 70 --ou8 --scale 0.5",
               dict(awgn_db=2.5, Fs=2e6, lo=10e9, ppm=3, drift_period=0.01, drift2_amp=500, drift2_freq=70, ou8=True, scale=0.5)),
              ("driftrate", "--deterministic -f 2e6 --lo 10e9 --ppm -3 --drift-rate 40000", dict(Fs=2e6, lo=10e9, ppm=-3, drift_rate=40000))]


def chansim_drifts(Fs=0.0, lo=0.0, ppm=-1.0, drift_period=0.0, drift_rate=0.0, drift2_amp=0.0, drift2_freq=0.0):
    """The drifter components leanchansim derives from its options (leanchansim.cc:155-170; config fields are floats, the
    expressions mix in double constants)."""
    f32, f64 = np.float32, np.float64
    Fs, lo, ppm, drift_period, drift_rate, drift2_amp, drift2_freq = (f32(v) for v in (Fs, lo, ppm, drift_period, drift_rate, drift2_amp, drift2_freq))
    with np.errstate(all="ignore"):
        maxoffs = f32(f64(lo * ppm) * 1e-6)
        amp = [float(maxoffs / Fs), 0.0, 0.0]
        freq = [0.0, 0.0, 0.0]
        if drift_period:
            freq[0] = float(f32((1.0 / f64(drift_period)) / f64(Fs)))
        if drift_rate:
            freq[0] = float(f32((f64(drift_rate) / (2 * np.pi * f64(ppm))) / f64(Fs)))
        if drift2_amp and drift2_freq:
            amp[1] = float(drift2_amp / Fs); freq[1] = float(drift2_freq / Fs)
    return amp, freq


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.lo_trig16_index.restype = C.c_uint
        L.lo_trig16_index.argtypes = [c_f]
        L.lo_cstln_lookup_index.restype = C.c_uint
        L.lo_cstln_lookup_index.argtypes = [c_f, c_f]
        L.lo_cstln_lut_init.argtypes = [C.c_void_p, C.c_int, c_f, c_f, c_f]
        L.lo_make_dvbs2_constellation.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.lo_lowpass.argtypes = [C.c_int, c_f, C.c_void_p, c_f]
        L.lo_root_raised_cosine.argtypes = [C.c_int, c_f, c_f, C.c_void_p]
        L.lo_normalize_dcgain.argtypes = [C.c_int, C.c_void_p, c_f]
        L.lo_normalize_power.argtypes = [C.c_int, C.c_void_p, c_f]
        L.lo_cconverter_u8.argtypes = [C.c_void_p, c_sz, C.c_void_p]
        L.lo_scaler.argtypes = [c_f, C.c_void_p, c_sz, C.c_void_p]
        L.lo_decimator.restype = c_sz
        L.lo_decimator.argtypes = [C.c_uint, C.c_void_p, c_sz, C.c_void_p, c_sz]
        L.lo_fir_shift_coeffs.argtypes = [C.c_uint, C.c_void_p, c_f, C.c_void_p]
        L.lo_fir_filter.restype = c_sz
        L.lo_fir_filter.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, c_sz,
                                    C.c_void_p, c_sz, C.POINTER(c_sz)]
        L.lo_fir_filter_fma.restype = c_sz
        L.lo_fir_filter_fma.argtypes = L.lo_fir_filter.argtypes
        L.lo_fir_filter_blk.restype = c_sz
        L.lo_fir_filter_blk.argtypes = L.lo_fir_filter.argtypes
        L.lo_fir_resampler_shift_coeffs.argtypes = [C.c_uint, C.c_void_p, c_f, C.c_void_p]
        L.lo_fir_resampler.restype = c_sz
        L.lo_fir_resampler.argtypes = [C.c_uint, C.c_void_p, C.c_int, C.c_void_p, c_sz,
                                       C.c_void_p, c_sz, C.POINTER(c_sz)]
        L.lo_cfft.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.lo_auto_notch_new.restype = C.c_void_p
        L.lo_auto_notch_new.argtypes = [C.c_int, C.c_int, c_f, c_f]
        L.lo_auto_notch_free.argtypes = [C.c_void_p]
        L.lo_auto_notch_run.restype = c_sz
        L.lo_auto_notch_run.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p]
        L.lo_auto_notch_slot_bin.argtypes = [C.c_void_p, C.c_int]
        L.lo_cnr_fft_new.restype = C.c_void_p
        L.lo_cnr_fft_new.argtypes = [c_f, C.c_int, C.c_int]
        L.lo_cnr_fft_free.argtypes = [C.c_void_p]
        L.lo_cnr_fft_run.restype = c_sz
        L.lo_cnr_fft_run.argtypes = [C.c_void_p, c_f, c_f, C.c_void_p, c_sz, C.c_void_p, c_sz]
        L.lo_spectrum_new.restype = C.c_void_p
        L.lo_spectrum_new.argtypes = [C.c_int, c_f]
        L.lo_spectrum_free.argtypes = [C.c_void_p]
        L.lo_spectrum_run.restype = c_sz
        L.lo_spectrum_run.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p, c_sz]
        L.lo_rx_new.restype = C.c_void_p
        L.lo_rx_new.argtypes = [C.POINTER(RxParams)]
        L.lo_rx_free.argtypes = [C.c_void_p]
        L.lo_rx_run.restype = c_sz
        L.lo_rx_run.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p, c_sz, C.POINTER(c_sz),
                                C.c_void_p, C.c_void_p, C.c_void_p, c_sz, C.POINTER(c_sz),
                                C.c_void_p, c_sz, C.POINTER(c_sz)]
        L.lo_rx_get_state.argtypes = [C.c_void_p, C.POINTER(RxState)]
        L.lo_rx_set_state.argtypes = [C.c_void_p, C.POINTER(RxState)]
        L.lo_rx_readahead.argtypes = [C.c_void_p]
        # FEC tail
        vp = C.c_void_p
        L.lo_deconv_new.restype = vp; L.lo_deconv_new.argtypes = [C.c_int, C.c_int]
        L.lo_deconv_free.argtypes = [vp]
        L.lo_deconv_info.argtypes = [vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lo_deconv_next_sync.argtypes = [vp]
        L.lo_deconv_locked.argtypes = [vp]
        L.lo_deconv_run.restype = c_sz; L.lo_deconv_run.argtypes = [vp, vp, c_sz, vp, c_sz, C.POINTER(c_sz)]
        L.lo_viterbi_new.restype = vp; L.lo_viterbi_new.argtypes = [C.c_int, C.c_int]
        L.lo_viterbi_free.argtypes = [vp]
        L.lo_viterbi_set_resync_period.argtypes = [vp, C.c_int]
        L.lo_viterbi_current_sync.argtypes = [vp]
        L.lo_viterbi_nsyncs.argtypes = [vp]
        L.lo_viterbi_map.argtypes = [vp, C.c_int, vp]
        L.lo_viterbi_run.restype = c_sz; L.lo_viterbi_run.argtypes = [vp, vp, c_sz, vp, c_sz, C.POINTER(c_sz)]
        L.lo_mpeg_sync_new.restype = vp; L.lo_mpeg_sync_new.argtypes = [C.c_int]
        L.lo_mpeg_sync_free.argtypes = [vp]
        L.lo_mpeg_sync_locked.argtypes = [vp]
        L.lo_mpeg_sync_run.restype = c_sz
        L.lo_mpeg_sync_run.argtypes = [vp, vp, c_sz, vp, c_sz, C.POINTER(c_sz), vp, c_sz, C.POINTER(c_sz), vp, c_sz,
                                       C.POINTER(c_sz), C.POINTER(C.c_int)]
        L.lo_deinterleaver.restype = c_sz; L.lo_deinterleaver.argtypes = [vp, c_sz, vp, c_sz, C.POINTER(c_sz)]
        L.lo_rs_tables.argtypes = [vp, vp, vp]
        L.lo_rs_encode.argtypes = [vp]
        L.lo_rs_decoder.restype = c_sz; L.lo_rs_decoder.argtypes = [vp, c_sz, vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.lo_derandomizer_pattern.argtypes = [vp]
        L.lo_derandomizer_new.restype = vp
        L.lo_derandomizer_free.argtypes = [vp]
        L.lo_derandomizer_run.restype = c_sz; L.lo_derandomizer_run.argtypes = [vp, vp, c_sz, vp]
        L.lo_fec_chain.restype = c_sz
        L.lo_fec_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, c_sz, vp, c_sz, C.POINTER(C.c_long), C.POINTER(C.c_long)]

    # -- tables
    def trig16(self):
        out = np.empty(65536, np.complex64)
        self.lib.lo_trig16(_p(out))
        return out

    def cstln_lut(self, predef, fec=0):
        c = CstlnLut()
        n = self.lib.lo_make_dvbs2_constellation(C.byref(c), predef, fec)
        assert n > 0
        return dict(nsymbols=c.nsymbols, nrotations=c.nrotations,
                    symbols=np.ctypeslib.as_array(c.symbols).reshape(256, 2)[:n].copy(),
                    cost=np.ctypeslib.as_array(c.cost).copy(),
                    symbol=np.ctypeslib.as_array(c.symbol).copy(),
                    phase_error=np.ctypeslib.as_array(c.phase_error).copy())

    def lowpass(self, order, fcut, renormalize=True):
        out = np.empty(order + 1, np.float32)
        n = self.lib.lo_lowpass(order, fcut, _p(out), 1.0)
        if renormalize:
            self.lib.lo_normalize_dcgain(n, _p(out), 1.0)
        return out[:n]

    def rrc(self, order, fs, rolloff):
        out = np.empty(order + 3, np.float32)
        n = self.lib.lo_root_raised_cosine(order, fs, rolloff, _p(out))
        return out[:n].copy()

    # -- blocks
    def cconverter_u8(self, x):
        x = np.ascontiguousarray(x, np.uint8).reshape(-1, 2)
        out = np.empty(len(x), np.complex64)
        self.lib.lo_cconverter_u8(_p(x), len(x), _p(out))
        return out

    def scaler(self, scale, x):
        x = cf32(x)
        out = np.empty_like(x)
        self.lib.lo_scaler(scale, _p(x), len(x), _p(out))
        return out

    def fir_shift(self, coeffs, freq):
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        out = np.empty(len(coeffs), np.complex64)
        self.lib.lo_fir_shift_coeffs(len(coeffs), _p(coeffs), freq, _p(out))
        return out

    def fir_filter(self, coeffs, decim, x, freq=0.0, shifted=None, fma=False, scale=None):
        """fma=True: the fused-multiply-add restatement (the arithmetic of LSDR_FIR_FMA / LSDR_FIR_MFMA); fma="blk": the
        block-polyphase restatement (LSDR_FIR_MFMA_BLK)."""
        x = cf32(x)
        sc = self.fir_shift(coeffs, freq) if shifted is None else cf32(shifted)
        if scale is not None:      # LSDR_FIR_MFMA_BLK: the fused scaler rides on the taps, one f32 rounding per component
            assert fma == "blk"
            sc = (sc.view(np.float32) * np.float32(scale)).view(np.complex64)
        cap = max(0, (len(x) - len(sc)) // decim) + 1
        out = np.empty(cap, np.complex64)
        consumed = c_sz()
        fn = self.lib.lo_fir_filter_blk if fma == "blk" else self.lib.lo_fir_filter_fma if fma else self.lib.lo_fir_filter
        n = fn(len(sc), _p(sc), decim, _p(x), len(x), _p(out), cap, C.byref(consumed))
        return out[:n], consumed.value

    def normalize_power(self, coeffs, gain):
        c = np.ascontiguousarray(coeffs, np.float32).copy()
        self.lib.lo_normalize_power(len(c), _p(c), gain)
        return c

    def decimator(self, d, x):
        x = cf32(x)
        out = np.empty(len(x) // d + 1, np.complex64)
        n = self.lib.lo_decimator(d, _p(x), len(x), _p(out), len(out))
        return out[:n].copy()

    def fir_resampler(self, coeffs, interp, x, freq=0.0):
        x = cf32(x)
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        sc = np.empty(len(coeffs), np.complex64)
        self.lib.lo_fir_resampler_shift_coeffs(len(coeffs), _p(coeffs), freq, _p(sc))
        cap = len(x) * interp
        out = np.empty(cap, np.complex64)
        consumed = c_sz()
        n = self.lib.lo_fir_resampler(len(sc), _p(sc), interp, _p(x), len(x), _p(out), cap, C.byref(consumed))
        return out[:n], consumed.value

    def cfft(self, x, reverse=False):
        x = cf32(x).copy()
        self.lib.lo_cfft(len(x), _p(x), int(reverse))
        return x

    def auto_notch(self, x, nslots=1, decimation=1024 * 4096, k=0.002, setpoint=0.0):
        x = cf32(x)
        h = self.lib.lo_auto_notch_new(nslots, decimation, k, setpoint)
        out = np.empty_like(x)
        n = self.lib.lo_auto_notch_run(h, _p(x), len(x), _p(out))
        bins = [self.lib.lo_auto_notch_slot_bin(h, s) for s in range(nslots)]
        self.lib.lo_auto_notch_free(h)
        return out[:n], bins

    def auto_notch_bins(self, x, decimation=1024 * 4096, k=0.002):
        """The slot's bin after every 4096-sample block of auto_notch (one slot), from the oracle run block by block: [−1, …, bin, …]."""
        x = cf32(x)
        h = self.lib.lo_auto_notch_new(1, decimation, k, 0.0)
        out = np.empty(4096, np.complex64)
        bins = []
        for b in range(len(x) // 4096):
            blk = np.ascontiguousarray(x[b * 4096:(b + 1) * 4096])
            self.lib.lo_auto_notch_run(h, _p(blk), 4096, _p(out))
            bins.append(self.lib.lo_auto_notch_slot_bin(h, 0))
        self.lib.lo_auto_notch_free(h)
        return bins

    def notch_fir_ideal(self, x, coeffs, decim, decimation=1024 * 4096, k=0.002, scale=None):
        """The arithmetic lsdr_notch_fir states (leansdr_amd/csrc/notch.hip), in float64: auto_notch's recurrence (sdr.h:119-138) as the
        LTI filter sub[n] = p·sub[n−1] + k·x[n], out = x − sub, p = (1−k)·exp(j2π·bin/4096) with EXACT phases — not the reference's
        float-rounded phasor table — restarted from 0 wherever a detect (sdr.h:76-118; bins taken from the oracle's own detect) changes the
        bin, then fir_filter (dsp.h:246-262) over it.  What the GPU block must equal to float32 accuracy for every bin; its distance
        from the reference's arithmetic for bins ≥ 2048 is the reference's table noise."""
        from scipy.signal import lfilter
        x = cf32(x)
        kf = np.float32(k)
        omk = float(np.float32(1) - kf)
        bins = self.auto_notch_bins(x, decimation, k)
        n = len(bins) * 4096
        xs = x[:n].astype(np.complex128) * (float(np.float32(scale)) if scale else 1.0)
        out = xs.copy()
        b = 0
        while b < len(bins):
            e = b
            while e < len(bins) and bins[e] == bins[b]:
                e += 1
            if bins[b] >= 0:
                p = omk * np.exp(2j * np.pi * bins[b] / 4096)
                out[b * 4096:e * 4096] -= lfilter([float(kf)], [1, -p], xs[b * 4096:e * 4096])
            b = e
        c = np.ascontiguousarray(coeffs, np.float32).astype(np.float64)
        N = len(c)
        count = (n - N) // decim if n >= N else 0
        idx = N + decim * np.arange(count)
        y = np.zeros(count, np.complex128)
        for i in range(N):            # y[m] = Σ_i c[i]·out[N + m·D − i]
            y += c[i] * out[idx - i]
        return y, bins

    def cnr_fft(self, x, bandwidth, nfft=4096, decimation=1048576, freq_tap=0.0, tap_multiplier=1.0):
        x = cf32(x)
        h = self.lib.lo_cnr_fft_new(bandwidth, nfft, decimation)
        out = np.empty(len(x) // nfft + 1, np.float32)
        n = self.lib.lo_cnr_fft_run(h, freq_tap, tap_multiplier, _p(x), len(x), _p(out), len(out))
        self.lib.lo_cnr_fft_free(h)
        return out[:n]

    def rotator(self, x, freq, splits=()):
        """rotator<f32>; `splits` cuts the stream into several run() calls (the 16-bit index is carried)."""
        x = cf32(x)
        L = self.lib
        L.lo_rotator_new.restype = C.c_void_p
        L.lo_rotator_new.argtypes = [c_f]
        L.lo_rotator_free.argtypes = [C.c_void_p]
        L.lo_rotator_run.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p]
        h = L.lo_rotator_new(freq)
        out = np.empty_like(x)
        pos = 0
        for e in list(splits) + [len(x)]:
            L.lo_rotator_run(h, x[pos:].ctypes.data, e - pos, out[pos:].ctypes.data)
            pos = e
        L.lo_rotator_free(h)
        return out

    def spectrum(self, x, decimation=1048576, kavg=0.1):
        x = cf32(x)
        h = self.lib.lo_spectrum_new(decimation, kavg)
        out = np.empty((len(x) // 1024 + 1, 1024), np.float32)
        n = self.lib.lo_spectrum_run(h, _p(x), len(x), _p(out), len(out))
        self.lib.lo_spectrum_free(h)
        return out[:n]

    def fast_qpsk(self, iq_u8, omega, freq=0.0, pll_adjustment=1.0, allow_drift=0, meas_decimation=0, return_tables=False):
        """fast_qpsk_receiver<u8> over an interleaved u8 I/Q array.  Returns dict(sym, consumed, freq, cstln, mu, phase, freqw)."""
        iq = np.ascontiguousarray(iq_u8, np.uint8)
        n = len(iq) // 2
        L = self.lib
        L.lo_fastqpsk_new.restype = C.c_void_p
        L.lo_fastqpsk_new.argtypes = [c_f, c_f, c_f, C.c_int, C.c_ulong]
        L.lo_fastqpsk_free.argtypes = [C.c_void_p]
        L.lo_fastqpsk_run.restype = c_sz
        L.lo_fastqpsk_run.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p, c_sz, C.POINTER(c_sz), C.c_void_p, c_sz, C.POINTER(c_sz), C.c_void_p, c_sz, C.POINTER(c_sz)]
        L.lo_fastqpsk_get_state.argtypes = [C.c_void_p, C.POINTER(c_f), C.POINTER(C.c_uint), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.lo_fastqpsk_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        h = L.lo_fastqpsk_new(omega, freq, pll_adjustment, allow_drift, meas_decimation)
        out = np.empty(n + 256, np.uint8)
        fo = np.empty(n // 64 + 16, np.float32)
        co = np.empty((n // 64 + 16, 2), np.uint8)
        cons, nf, nc = c_sz(), c_sz(), c_sz()
        k = L.lo_fastqpsk_run(h, _p(iq), n, _p(out), len(out), C.byref(cons), _p(fo), len(fo), C.byref(nf), _p(co), len(co), C.byref(nc))
        mu, ph, fw, mn, mx = c_f(), C.c_uint(), C.c_long(), C.c_long(), C.c_long()
        L.lo_fastqpsk_get_state(h, C.byref(mu), C.byref(ph), C.byref(fw), C.byref(mn), C.byref(mx))
        res = dict(sym=out[:k].copy(), consumed=cons.value, freq=fo[:nf.value].copy(), cstln=co[:nc.value].copy(),
                   mu=mu.value, phase=ph.value, freqw=fw.value, min_freqw=mn.value, max_freqw=mx.value)
        if return_tables:
            pa, pr = np.empty(65536, np.uint16), np.empty(65536, np.uint8)
            re, sc = np.empty((256, 256, 2), np.uint8), np.empty((65536, 2), np.uint8)
            L.lo_fastqpsk_tables(h, _p(pa), _p(pr), _p(re), _p(sc))
            res.update(polar_a=pa, polar_r=pr, rect=re, sincos=sc)
        L.lo_fastqpsk_free(h)
        return res

    # ---- transmit chain (leandvbtx)
    def randomizer(self, ts):
        ts = np.ascontiguousarray(ts, np.uint8).reshape(-1, 188)
        out = np.empty_like(ts)
        pos = C.c_uint(0)
        self.lib.lo_randomizer.argtypes = [C.POINTER(C.c_uint), C.c_void_p, c_sz, C.c_void_p]
        self.lib.lo_randomizer(C.byref(pos), _p(ts), len(ts), _p(out))
        return out

    def rs_encoder(self, ts):
        ts = np.ascontiguousarray(ts, np.uint8).reshape(-1, 188)
        out = np.zeros((len(ts), 204), np.uint8)
        out[:, :188] = ts
        self.lib.lo_rs_encode.argtypes = [C.c_void_p]
        for k in range(len(out)):
            self.lib.lo_rs_encode(out[k].ctypes.data)
        return out

    def interleaver(self, packets):
        pk = np.ascontiguousarray(packets, np.uint8).reshape(-1, 204)
        out = np.empty(len(pk) * 204, np.uint8)
        c = c_sz()
        self.lib.lo_interleaver.restype = c_sz
        self.lib.lo_interleaver.argtypes = [C.c_void_p, c_sz, C.c_void_p, c_sz, C.POINTER(c_sz)]
        n = self.lib.lo_interleaver(_p(pk), len(pk), _p(out), len(out), C.byref(c))
        return out[:n].copy()

    def dvb_convol(self, data, rate=0, bps=2):
        data = np.ascontiguousarray(data, np.uint8)
        L = self.lib
        L.lo_convol_new.restype = C.c_void_p
        L.lo_convol_new.argtypes = [C.c_int, C.c_int]
        L.lo_convol_free.argtypes = [C.c_void_p]
        L.lo_convol_run.restype = c_sz
        L.lo_convol_run.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p, c_sz, C.POINTER(c_sz)]
        h = L.lo_convol_new(rate, bps)
        assert h
        out = np.empty(len(data) * 16 + 64, np.uint8)
        c = c_sz()
        n = L.lo_convol_run(h, _p(data), len(data), _p(out), len(out), C.byref(c))
        L.lo_convol_free(h)
        return out[:n].copy(), c.value

    def cstln_transmitter(self, sym, cstln=1, rate=0):
        sym = np.ascontiguousarray(sym, np.uint8)
        c = CstlnLut()
        assert self.lib.lo_make_dvbs2_constellation(C.byref(c), cstln, rate) > 0
        out = np.empty(len(sym), np.complex64)
        self.lib.lo_cstln_transmitter.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p]
        self.lib.lo_cstln_transmitter(C.byref(c), _p(sym), len(sym), _p(out))
        return out

    def tx_chain(self, ts, interp=2, decim=1, amp=1.0, rolloff=0.35, rrc_rej=10.0, agc=False, cstln=1, rate=0):
        """leandvbtx (leandvbtx.cc:79-175): TS packets → cf32 baseband."""
        r = self.randomizer(ts)
        pk = self.rs_encoder(r)
        il = self.interleaver(pk)
        bps = {0: 1, 1: 2, 2: 3, 3: 4, 4: 5, 5: 6, 6: 4, 7: 6, 8: 8}[cstln]
        conv_rate = 2 if (rate == 1 and bps in (2, 6)) else rate   # 2/3 on QPSK / 64-ary runs as 4/6 (leandvbtx.cc:117-121)
        sym, _ = self.dvb_convol(il, conv_rate, bps)
        iq = self.cstln_transmitter(sym, cstln, rate)
        order = int(interp * rrc_rej)
        co = self.rrc(order, float(np.float32(1.0) / np.float32(interp)), rolloff)
        co = self.normalize_power(co, float(np.float32(amp) / np.float32(75.0)))
        y, _ = self.fir_resampler(co, interp, iq)
        y = self.decimator(decim, y)
        if agc:
            y, _ = self.simple_agc(y, float(np.float32(amp) / np.sqrt(np.float32(np.float32(interp) / decim))), float(np.float32(0.001 * decim / interp)))
        return y

    def simple_agc(self, x, out_rms=1.0, bw=0.001, estimated=0.0):
        x = cf32(x)
        out = np.empty_like(x)
        est = c_f(estimated)
        self.lib.lo_simple_agc.restype = c_sz
        self.lib.lo_simple_agc.argtypes = [C.POINTER(c_f), c_f, c_f, C.c_void_p, c_sz, C.c_void_p]
        n = self.lib.lo_simple_agc(C.byref(est), out_rms, bw, _p(x), len(x), _p(out))
        return out[:n].copy(), est.value

    # ---- channel simulator (lsdr_oracle_chan.c) ----
    def drand48(self, n, seed=None):
        """n consecutive drand48() results and the final 48-bit state."""
        L = self.lib
        L.lo_drand48_default.restype = C.c_uint64
        L.lo_srand48.restype = C.c_uint64
        L.lo_srand48.argtypes = [C.c_long]
        L.lo_drand48.restype = C.c_double
        L.lo_drand48.argtypes = [C.POINTER(C.c_uint64)]
        x = C.c_uint64(L.lo_drand48_default() if seed is None else L.lo_srand48(seed))
        out = np.array([L.lo_drand48(C.byref(x)) for _ in range(n)], np.float64)
        return out, x.value

    def logf(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.lib.lo_logf.restype = c_f
        self.lib.lo_logf.argtypes = [c_f]
        return np.array([self.lib.lo_logf(float(v)) for v in x], np.float32)

    def wgn(self, n, stddev=1.0, seed=None, state=None):
        """wgn_c<f32>: n samples.  Returns (samples, drand48 state after)."""
        L = self.lib
        L.lo_drand48_default.restype = C.c_uint64
        L.lo_srand48.restype = C.c_uint64
        L.lo_srand48.argtypes = [C.c_long]
        x = C.c_uint64(state if state is not None else (L.lo_drand48_default() if seed is None else L.lo_srand48(seed)))
        out = np.empty(n, np.complex64)
        L.lo_wgn.argtypes = [C.POINTER(C.c_uint64), c_f, C.c_void_p, c_sz]
        L.lo_wgn(C.byref(x), stddev, _p(out), n)
        return out, x.value

    def adder(self, a, b):
        a = cf32(a); b = cf32(b)
        n = min(len(a), len(b))
        out = np.empty(n, np.complex64)
        self.lib.lo_adder.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p]
        self.lib.lo_adder(_p(a), _p(b), n, _p(out))
        return out

    def cconv_f32_u8(self, x):
        x = cf32(x)
        out = np.empty((len(x), 2), np.uint8)
        self.lib.lo_cconv_f32_u8.argtypes = [C.c_void_p, c_sz, C.c_void_p]
        self.lib.lo_cconv_f32_u8(_p(x), len(x), _p(out))
        return out

    def cconv_f32_s16(self, x):
        x = cf32(x)
        out = np.empty((len(x), 2), np.int16)
        self.lib.lo_cconv_f32_s16.argtypes = [C.c_void_p, c_sz, C.c_void_p]
        self.lib.lo_cconv_f32_s16(_p(x), len(x), _p(out))
        return out

    def drifter_trig(self):
        lut = np.empty(65536, np.complex64)
        self.lib.lo_drifter_trig.argtypes = [C.c_void_p]
        self.lib.lo_drifter_trig(_p(lut))
        return lut

    def drifter(self, x, amp, freq, a=(0, 0, 0), chunk=4096):
        """drifter<float> over x cut into run() calls of `chunk` samples (leanchansim's pipes hold 4096).
        Returns (out, a after)."""
        x = cf32(x)
        lut = self.drifter_trig()
        out = np.empty_like(x)
        amp = (c_f * 3)(*amp); freq = (c_f * 3)(*freq); aa = (C.c_long * 3)(*a)
        self.lib.lo_drifter_run.argtypes = [C.c_void_p, c_f * 3, c_f * 3, C.c_long * 3, C.c_void_p, c_sz, C.c_void_p]
        chunk = chunk or max(len(x), 1)
        for p in range(0, len(x), chunk):
            m = min(chunk, len(x) - p)
            self.lib.lo_drifter_run(_p(lut), amp, freq, aa, x[p:].ctypes.data, m, out[p:].ctypes.data)
        return out, tuple(aa)

    def chansim(self, x, awgn_db=None, scale=1.0, Fs=0.0, lo=0.0, ppm=-1.0, drift_period=0.0, drift_rate=0.0, drift2_amp=0.0,
                drift2_freq=0.0, ou8=False, seed=None, chunk=4096):
        """leanchansim (leanchansim.cc:120-176) on cf32 input: scaler → + wgn_c → drifter → [cconverter to u8]."""
        x = cf32(x)
        y = self.scaler(scale, x)
        self.lib.lo_db_to_amp.restype = c_f
        self.lib.lo_db_to_amp.argtypes = [C.c_double]
        stddev = self.lib.lo_db_to_amp(awgn_db) if awgn_db is not None else 0.0
        noise, _ = self.wgn(len(y), stddev, seed)
        y = self.adder(y, noise)
        amp, freq = chansim_drifts(Fs, lo, ppm, drift_period, drift_rate, drift2_amp, drift2_freq)
        y, _ = self.drifter(y, amp, freq, chunk=chunk)
        return self.cconv_f32_u8(y) if ou8 else y

    def hs_chain(self, iq_u8, omega, fastlock=0):
        """leandvb --hs (leandvb.cc:727-969): fast_qpsk_receiver → dvb_deconvol_sync_hard → mpeg_sync(fastlock, resync) →
        deinterleaver → rs_decoder → derandomizer.  Returns TS packets."""
        sym = self.fast_qpsk(iq_u8, omega)["sym"]
        by = self.hs_deconvol(sym, 1 if fastlock else 32)
        mb = self.mpeg_sync(by, fastlock=1, resync_period=1 if fastlock else 32)[0]
        pk = self.deinterleaver(mb)
        ts = self.rs_decoder(pk)[0]
        return self.derandomizer(ts)

    def hs_deconvol(self, symbols, resync_period=32, pipe=None, room=None):
        """dvb_deconvol_sync<u8> over hard symbols; optional call pattern (pipe symbols per call, room bytes)."""
        sym = np.ascontiguousarray(symbols, np.uint8)
        L = self.lib
        L.lo_hsdeconv_new.restype = C.c_void_p
        L.lo_hsdeconv_new.argtypes = [C.c_int]
        L.lo_hsdeconv_free.argtypes = [C.c_void_p]
        L.lo_hsdeconv_run.restype = c_sz
        L.lo_hsdeconv_run.argtypes = [C.c_void_p, C.c_void_p, c_sz, C.c_void_p, c_sz, C.POINTER(c_sz)]
        h = L.lo_hsdeconv_new(resync_period)
        out = np.empty(len(sym) // 8 + 64, np.uint8)
        pos = nout = 0
        while True:
            c = c_sz()
            avail = len(sym) - pos if pipe is None else min(pipe, len(sym) - pos)
            cap = len(out) - nout if room is None else min(room, len(out) - nout)
            k = L.lo_hsdeconv_run(h, sym[pos:].ctypes.data, avail, out[nout:].ctypes.data, cap, C.byref(c))
            if not k:
                break
            pos += c.value
            nout += k
        L.lo_hsdeconv_free(h)
        return out[:nout].copy()

    def rx(self, params, x, state_in=None, chunks=None):
        """Run cstln_receiver over x.  Returns dict(sym, consumed, freq, ss, mer, cstln, state)."""
        x = cf32(x)
        h = self.lib.lo_rx_new(C.byref(params))
        if state_in is not None:
            self.lib.lo_rx_set_state(h, C.byref(state_in))
        cap = len(x) + 256
        out = np.zeros(cap, SOFTSYM)
        mcap = len(x) // max(1, params.meas_decimation) + 8
        ccap = len(x) // 128 + 8
        fr, ss, mer = (np.empty(mcap, np.float32) for _ in range(3))
        cst = np.empty(ccap, np.complex64)
        consumed, nm, nc = c_sz(), c_sz(), c_sz()
        n = self.lib.lo_rx_run(h, _p(x), len(x), _p(out), cap, C.byref(consumed),
                               _p(fr), _p(ss), _p(mer), mcap, C.byref(nm),
                               _p(cst), ccap, C.byref(nc))
        st = RxState()
        self.lib.lo_rx_get_state(h, C.byref(st))
        self.lib.lo_rx_free(h)
        return dict(sym=out[:n], consumed=consumed.value, freq=fr[:nm.value], ss=ss[:nm.value],
                    mer=mer[:nm.value], cstln=cst[:nc.value], state=st)


    # -- FEC tail ------------------------------------------------------------
    @staticmethod
    def softsyms(symbol, cost=None):
        out = np.zeros(len(symbol), SOFTSYM)
        out["symbol"] = symbol
        if cost is not None:
            out["cost"] = cost
        return out

    def deconv_info(self, rate):
        h = self.lib.lo_deconv_new(rate, 0)
        d, d2 = np.zeros(8, np.uint64), np.zeros(8, np.uint64)
        p, w = C.c_int(), C.c_int()
        self.lib.lo_deconv_info(h, _p(d), _p(d2), C.byref(p), C.byref(w))
        self.lib.lo_deconv_free(h)
        return d[:p.value].copy(), d2[:p.value].copy(), p.value, w.value

    def deconvol_sync(self, sym, rate=0, fastlock=0, next_syncs=0, pipe=4096):
        """Whole-stream deconvolution: repeated run() calls on windows of at most `pipe` symbols,
        output room 8192 like the reference pipes (leandvb.cc:185-202)."""
        sym = np.ascontiguousarray(sym, SOFTSYM)
        h = self.lib.lo_deconv_new(rate, fastlock)
        for _ in range(next_syncs):
            self.lib.lo_deconv_next_sync(h)
        out = np.empty(len(sym) + 64, np.uint8)
        pos, nout = 0, 0
        while True:
            c = c_sz()
            avail = min(pipe, len(sym) - pos)
            got = self.lib.lo_deconv_run(h, sym[pos:].ctypes.data, avail, out[nout:].ctypes.data, 8192, C.byref(c))
            if not got and not c.value:
                break
            pos += c.value
            nout += got
        self.lib.lo_deconv_free(h)
        return out[:nout].copy()

    def viterbi_sync(self, sym, cstln=1, rate=0, resync_period=0):
        sym = np.ascontiguousarray(sym, SOFTSYM)
        h = self.lib.lo_viterbi_new(cstln, rate)
        assert h
        if resync_period:
            self.lib.lo_viterbi_set_resync_period(h, resync_period)
        out = np.empty(len(sym) + 64, np.uint8)
        c = c_sz()
        n = self.lib.lo_viterbi_run(h, _p(sym), len(sym), _p(out), len(out), C.byref(c))
        cur = self.lib.lo_viterbi_current_sync(h)
        self.lib.lo_viterbi_free(h)
        return out[:n].copy(), c.value, cur

    def mpeg_sync(self, data, fastlock=0, resync_period=0):
        data = np.ascontiguousarray(data, np.uint8)
        h = self.lib.lo_mpeg_sync_new(fastlock)
        if resync_period:
            self.lib.lo_mpeg_sync_set_resync_period.argtypes = [C.c_void_p, C.c_int]
            self.lib.lo_mpeg_sync_set_resync_period(h, resync_period)
        out = np.empty(len(data) + 4096, np.uint8)
        st = np.empty(len(data) // 204 + 64, np.int32)
        lt = np.empty(len(data) // 204 + 64, np.uint64)
        pos, nout, nst, nlt = 0, 0, 0, 0
        while True:
            c, ns, nl, cns = c_sz(), c_sz(), c_sz(), C.c_int()
            got = self.lib.lo_mpeg_sync_run(h, data[pos:].ctypes.data, len(data) - pos, out[nout:].ctypes.data,
                                            len(out) - nout, C.byref(c), st[nst:].ctypes.data, len(st) - nst, C.byref(ns),
                                            lt[nlt:].ctypes.data, len(lt) - nlt, C.byref(nl), C.byref(cns))
            nst += ns.value; nlt += nl.value
            if not got and not c.value:
                break
            pos += c.value; nout += got
        self.lib.lo_mpeg_sync_free(h)
        return out[:nout].copy(), st[:nst].copy(), lt[:nlt].copy()

    def deinterleaver(self, data):
        data = np.ascontiguousarray(data, np.uint8)
        out = np.empty((len(data) // 204 + 1, 204), np.uint8)
        c = c_sz()
        n = self.lib.lo_deinterleaver(_p(data), len(data), _p(out), len(out), C.byref(c))
        return out[:n].copy()

    def rs_tables(self):
        e, l, g = np.empty(512, np.uint8), np.empty(256, np.uint8), np.empty(17, np.uint8)
        self.lib.lo_rs_tables(_p(e), _p(l), _p(g))
        return e, l, g

    def rs_encode(self, msg188):
        m = np.zeros(204, np.uint8)
        m[:188] = msg188
        self.lib.lo_rs_encode(_p(m))
        return m

    def rs_decoder(self, packets):
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 204).copy()
        out = np.empty((len(packets), 188), np.uint8)
        b, e = C.c_long(), C.c_long()
        self.lib.lo_rs_decoder(_p(packets), len(packets), _p(out), C.byref(b), C.byref(e))
        return out, b.value, e.value

    def derandomizer_pattern(self):
        p = np.empty(1504, np.uint8)
        self.lib.lo_derandomizer_pattern(_p(p))
        return p

    def derandomizer(self, packets):
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 188)
        h = self.lib.lo_derandomizer_new()
        out = np.empty_like(packets)
        n = self.lib.lo_derandomizer_run(h, _p(packets), len(packets), _p(out))
        self.lib.lo_derandomizer_free(h)
        return out[:n].copy()

    def fec_chain(self, sym, cstln=1, rate=0, viterbi=0, fastlock=0):
        sym = np.ascontiguousarray(sym, SOFTSYM)
        out = np.empty((len(sym) // 800 + 16, 188), np.uint8)
        b, e = C.c_long(), C.c_long()
        n = self.lib.lo_fec_chain(cstln, rate, viterbi, fastlock, _p(sym), len(sym), _p(out), len(out), C.byref(b), C.byref(e))
        return out[:n].copy(), b.value, e.value


class Ref:
    """The real reference, through oracle/ref_harness.cc.  Raises FileNotFoundError
    when oracle/_ref was not built (no /root/reference on this machine)."""

    def __init__(self, path=REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path)
        L.ref_trig16_index.restype = C.c_uint
        L.ref_trig16_index.argtypes = [c_f]
        L.ref_cstln_lut.argtypes = [C.c_int, c_f, c_f, c_f, C.c_int] + [C.c_void_p] * 5
        L.ref_cstln_lookup.argtypes = [C.c_int, c_f, c_f, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_make_dvbs2_constellation.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4
        L.ref_lowpass.argtypes = [C.c_int, c_f, C.c_int, C.c_void_p]
        L.ref_root_raised_cosine.argtypes = [C.c_int, c_f, c_f, C.c_void_p]
        L.ref_cconverter_u8.restype = C.c_long
        L.ref_cconverter_u8.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.ref_scaler.restype = C.c_long
        L.ref_scaler.argtypes = [c_f, C.c_void_p, C.c_long, C.c_void_p]
        L.ref_fir_filter.restype = C.c_long
        L.ref_fir_filter.argtypes = [C.c_int, C.c_void_p, C.c_uint, c_f, C.c_void_p, C.c_long,
                                     C.c_void_p, C.c_long, C.c_void_p]
        L.ref_fir_resampler.restype = C.c_long
        L.ref_fir_resampler.argtypes = [C.c_int, C.c_void_p, C.c_int, c_f, C.c_void_p, C.c_long,
                                        C.c_void_p, C.c_long]
        L.ref_auto_notch.restype = C.c_long
        L.ref_auto_notch.argtypes = [C.c_int, C.c_int, c_f, c_f, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.ref_cfft.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.ref_cnr_fft.restype = C.c_long
        L.ref_cnr_fft.argtypes = [c_f, C.c_int, C.c_int, c_f, c_f, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        L.ref_spectrum.restype = C.c_long
        L.ref_spectrum.argtypes = [C.c_int, c_f, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        L.ref_cstln_receiver.restype = C.c_long
        L.ref_cstln_receiver.argtypes = [C.POINTER(RxParams), C.c_void_p, C.c_long, C.c_void_p,
                                         C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_long, C.POINTER(C.c_long),
                                         C.POINTER(C.c_long), C.POINTER(RxState)]

    def _fec_sigs(self):
        L, vp = self.lib, C.c_void_p
        L.ref_deconvol_sync.restype = C.c_long
        L.ref_deconvol_sync.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_long, vp, C.c_long, vp, vp]
        L.ref_viterbi_sync.restype = C.c_long
        L.ref_viterbi_sync.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_long, vp, C.c_long, C.POINTER(C.c_int)]
        L.ref_mpeg_sync.restype = C.c_long
        L.ref_mpeg_sync.argtypes = [C.c_int, vp, C.c_long, vp, C.c_long, vp, C.c_long, C.POINTER(C.c_long), vp, C.POINTER(C.c_long)]
        L.ref_deinterleaver.restype = C.c_long
        L.ref_deinterleaver.argtypes = [vp, C.c_long, vp, C.c_long]
        L.ref_rs_decoder.restype = C.c_long
        L.ref_rs_decoder.argtypes = [vp, C.c_long, vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.ref_rs_tables.argtypes = [vp, vp, vp]
        L.ref_rs_encode.argtypes = [vp]
        L.ref_derandomizer.restype = C.c_long
        L.ref_derandomizer.argtypes = [vp, C.c_long, vp, vp]
        L.ref_fec_chain.restype = C.c_long
        L.ref_fec_chain.argtypes = [C.c_int] * 5 + [vp, vp, C.c_long, vp, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long)]

    def deconvol_sync(self, sym, rate=0, fastlock=0, next_syncs=0):
        self._fec_sigs()
        symbol = np.ascontiguousarray(sym["symbol"])
        out = np.empty(len(symbol) + 64, np.uint8)
        dc = np.zeros(8, np.uint64); pw = np.zeros(2, np.int32)
        n = self.lib.ref_deconvol_sync(rate, fastlock, next_syncs, _p(symbol), len(symbol), _p(out), len(out), _p(dc), _p(pw))
        return out[:n].copy(), dc[:pw[0]].copy(), int(pw[0]), int(pw[1])

    def viterbi_sync(self, sym, cstln=1, rate=0, resync_period=0):
        self._fec_sigs()
        cost, symbol = np.ascontiguousarray(sym["cost"]), np.ascontiguousarray(sym["symbol"])
        out = np.empty(len(symbol) + 64, np.uint8)
        cur = C.c_int()
        n = self.lib.ref_viterbi_sync(cstln, rate, resync_period, _p(cost), _p(symbol), len(symbol), _p(out), len(out), C.byref(cur))
        return out[:n].copy(), cur.value

    def mpeg_sync(self, data, fastlock=0):
        self._fec_sigs()
        data = np.ascontiguousarray(data, np.uint8)
        out = np.empty(len(data) + 4096, np.uint8)
        st = np.empty(len(data) // 204 + 64, np.int32)
        lt = np.empty(len(data) // 204 + 64, np.uint64)
        ns, nl = C.c_long(), C.c_long(len(lt))
        n = self.lib.ref_mpeg_sync(fastlock, _p(data), len(data), _p(out), len(out), _p(st), len(st), C.byref(ns), _p(lt), C.byref(nl))
        return out[:n].copy(), st[:ns.value].copy(), lt[:nl.value].copy()

    def deinterleaver(self, data):
        self._fec_sigs()
        data = np.ascontiguousarray(data, np.uint8)
        out = np.empty((len(data) // 204 + 1, 204), np.uint8)
        n = self.lib.ref_deinterleaver(_p(data), len(data), _p(out), len(out))
        return out[:n].copy()

    def rs_tables(self):
        self._fec_sigs()
        e, l, g = np.empty(512, np.uint8), np.empty(256, np.uint8), np.empty(17, np.uint8)
        self.lib.ref_rs_tables(_p(e), _p(l), _p(g))
        return e, l, g

    def rs_encode(self, msg188):
        self._fec_sigs()
        m = np.zeros(204, np.uint8)
        m[:188] = msg188
        self.lib.ref_rs_encode(_p(m))
        return m

    def rs_decoder(self, packets):
        self._fec_sigs()
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 204)
        out = np.empty((len(packets), 188), np.uint8)
        b, e = C.c_long(), C.c_long()
        self.lib.ref_rs_decoder(_p(packets), len(packets), _p(out), C.byref(b), C.byref(e))
        return out, b.value, e.value

    def derandomizer(self, packets):
        self._fec_sigs()
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 188)
        out = np.empty_like(packets)
        pat = np.empty(1504, np.uint8)
        n = self.lib.ref_derandomizer(_p(packets), len(packets), _p(out), _p(pat))
        return out[:n].copy(), pat

    def fec_chain(self, sym, cstln=1, rate=0, viterbi=0, fastlock=0, buf_factor=4):
        self._fec_sigs()
        cost, symbol = np.ascontiguousarray(sym["cost"]), np.ascontiguousarray(sym["symbol"])
        out = np.empty((len(symbol) // 800 + 16, 188), np.uint8)
        b, e = C.c_long(), C.c_long()
        n = self.lib.ref_fec_chain(cstln, rate, viterbi, fastlock, buf_factor, _p(cost), _p(symbol), len(symbol), _p(out), len(out), C.byref(b), C.byref(e))
        return out[:n].copy(), b.value, e.value

    def trig16(self):
        out = np.empty(65536, np.complex64)
        self.lib.ref_trig16(_p(out))
        return out

    def cstln_lut(self, predef, fec=0):
        cost = np.empty(65536, np.int16)
        sym = np.empty(65536, np.uint8)
        pe = np.empty(65536, np.int16)
        symbols = np.zeros((256, 2), np.int8)
        n = self.lib.ref_make_dvbs2_constellation(predef, fec, _p(cost), _p(sym), _p(pe), _p(symbols))
        return dict(nsymbols=n, symbols=symbols[:n].copy(), cost=cost, symbol=sym, phase_error=pe)

    def lowpass(self, order, fcut, renormalize=True):
        out = np.empty(order + 1, np.float32)
        n = self.lib.ref_lowpass(order, fcut, int(renormalize), _p(out))
        return out[:n]

    def rrc(self, order, fs, rolloff):
        out = np.empty(order + 3, np.float32)
        n = self.lib.ref_root_raised_cosine(order, fs, rolloff, _p(out))
        return out[:n].copy()

    def cconverter_u8(self, x):
        x = np.ascontiguousarray(x, np.uint8).reshape(-1, 2)
        out = np.empty(len(x), np.complex64)
        n = self.lib.ref_cconverter_u8(_p(x), len(x), _p(out))
        return out[:n]

    def scaler(self, scale, x):
        x = cf32(x)
        out = np.empty_like(x)
        n = self.lib.ref_scaler(scale, _p(x), len(x), _p(out))
        return out[:n]

    def fir_filter(self, coeffs, decim, x, freq=0.0):
        x = cf32(x)
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        cap = max(0, (len(x) - len(coeffs)) // decim) + 1
        out = np.empty(cap, np.complex64)
        sc = np.empty(len(coeffs), np.complex64)
        n = self.lib.ref_fir_filter(len(coeffs), _p(coeffs), decim, freq, _p(x), len(x), _p(out), cap, _p(sc))
        return out[:n], sc

    def fir_resampler(self, coeffs, interp, x, freq=0.0):
        x = cf32(x)
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        cap = len(x) * interp
        out = np.empty(cap, np.complex64)
        n = self.lib.ref_fir_resampler(len(coeffs), _p(coeffs), interp, freq, _p(x), len(x), _p(out), cap)
        return out[:n]

    def cfft(self, x, reverse=False):
        x = cf32(x).copy()
        self.lib.ref_cfft(len(x), _p(x), int(reverse))
        return x

    def auto_notch(self, x, nslots=1, decimation=1024 * 4096, k=0.002, setpoint=0.0):
        x = cf32(x)
        out = np.empty_like(x)
        bins = np.zeros(max(1, nslots), np.int32)
        n = self.lib.ref_auto_notch(nslots, decimation, k, setpoint, _p(x), len(x), _p(out), _p(bins))
        return out[:n], list(bins[:nslots])

    def cnr_fft(self, x, bandwidth, nfft=4096, decimation=1048576, freq_tap=0.0, tap_multiplier=1.0):
        x = cf32(x)
        out = np.empty(len(x) // nfft + 1, np.float32)
        n = self.lib.ref_cnr_fft(bandwidth, nfft, decimation, freq_tap, tap_multiplier, _p(x), len(x), _p(out), len(out))
        return out[:n]

    def rotator(self, x, freq):
        x = cf32(x)
        self.lib.ref_rotator.restype = C.c_long
        self.lib.ref_rotator.argtypes = [c_f, C.c_void_p, C.c_long, C.c_void_p]
        out = np.empty_like(x)
        n = self.lib.ref_rotator(freq, _p(x), len(x), _p(out))
        return out[:n]

    def spectrum(self, x, decimation=1048576, kavg=0.1):
        x = cf32(x)
        out = np.empty((len(x) // 1024 + 1, 1024), np.float32)
        n = self.lib.ref_spectrum(decimation, kavg, _p(x), len(x), _p(out), len(out))
        return out[:n]

    def fast_qpsk(self, iq_u8, omega, freq=0.0, pll_adjustment=1.0, allow_drift=0, meas_decimation=0, return_tables=False):
        iq = np.ascontiguousarray(iq_u8, np.uint8)
        n = len(iq) // 2
        L = self.lib
        L.ref_fast_qpsk.restype = C.c_long
        L.ref_fast_qpsk.argtypes = [c_f, c_f, c_f, C.c_int, C.c_ulong, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                    C.POINTER(C.c_long), C.c_void_p, C.c_long, C.POINTER(C.c_long), C.POINTER(c_f), C.POINTER(C.c_uint),
                                    C.POINTER(C.c_long), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        out = np.empty(n + 256, np.uint8)
        fo = np.empty(n // 64 + 16, np.float32)
        co = np.empty((n // 64 + 16, 2), np.uint8)
        nf, nc = C.c_long(), C.c_long()
        mu, ph, fw = c_f(), C.c_uint(), C.c_long()
        pa, pr = np.empty(65536, np.uint16), np.empty(65536, np.uint8)
        re, sc = np.empty((256, 256, 2), np.uint8), np.empty((65536, 2), np.uint8)
        k = L.ref_fast_qpsk(omega, freq, pll_adjustment, allow_drift, meas_decimation, _p(iq), n, _p(out), len(out), _p(fo), len(fo),
                            C.byref(nf), _p(co), len(co), C.byref(nc), C.byref(mu), C.byref(ph), C.byref(fw), _p(pa), _p(pr), _p(re), _p(sc))
        res = dict(sym=out[:k].copy(), freq=fo[:nf.value].copy(), cstln=co[:nc.value].copy(), mu=mu.value, phase=ph.value, freqw=fw.value)
        if return_tables:
            res.update(polar_a=pa, polar_r=pr, rect=re, sincos=sc)
        return res

    def randomizer(self, ts):
        ts = np.ascontiguousarray(ts, np.uint8).reshape(-1, 188)
        out = np.empty_like(ts)
        self.lib.ref_randomizer.restype = C.c_long
        self.lib.ref_randomizer.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        n = self.lib.ref_randomizer(_p(ts), len(ts), _p(out))
        return out[:n]

    def rs_encoder(self, ts):
        ts = np.ascontiguousarray(ts, np.uint8).reshape(-1, 188)
        out = np.empty((len(ts), 204), np.uint8)
        self.lib.ref_rs_encoder.restype = C.c_long
        self.lib.ref_rs_encoder.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        n = self.lib.ref_rs_encoder(_p(ts), len(ts), _p(out))
        return out[:n]

    def interleaver(self, packets):
        pk = np.ascontiguousarray(packets, np.uint8).reshape(-1, 204)
        out = np.empty(len(pk) * 204, np.uint8)
        self.lib.ref_interleaver.restype = C.c_long
        self.lib.ref_interleaver.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        n = self.lib.ref_interleaver(_p(pk), len(pk), _p(out), len(out))
        return out[:n].copy()

    def dvb_convol(self, data, rate=0, bps=2):
        data = np.ascontiguousarray(data, np.uint8)
        out = np.empty(len(data) * 16 + 64, np.uint8)
        self.lib.ref_dvb_convol.restype = C.c_long
        self.lib.ref_dvb_convol.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        n = self.lib.ref_dvb_convol(rate, bps, _p(data), len(data), _p(out), len(out))
        return out[:n].copy()

    def cstln_transmitter(self, sym, cstln=1, rate=0):
        sym = np.ascontiguousarray(sym, np.uint8)
        out = np.empty(len(sym), np.complex64)
        self.lib.ref_cstln_transmitter.restype = C.c_long
        self.lib.ref_cstln_transmitter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
        n = self.lib.ref_cstln_transmitter(cstln, rate, _p(sym), len(sym), _p(out))
        return out[:n]

    def simple_agc(self, x, out_rms=1.0, bw=0.001):
        x = cf32(x)
        out = np.empty_like(x)
        est = c_f()
        self.lib.ref_simple_agc.restype = C.c_long
        self.lib.ref_simple_agc.argtypes = [c_f, c_f, C.c_void_p, C.c_long, C.c_void_p, C.POINTER(c_f)]
        n = self.lib.ref_simple_agc(out_rms, bw, _p(x), len(x), _p(out), C.byref(est))
        return out[:n].copy(), est.value

    def wgn(self, n, stddev=1.0, seed=None):
        out = np.empty(n, np.complex64)
        self.lib.ref_wgn.restype = C.c_long
        self.lib.ref_wgn.argtypes = [C.c_int, C.c_long, c_f, C.c_void_p, C.c_long]
        k = self.lib.ref_wgn(0 if seed is None else 1, seed or 0, stddev, _p(out), n)
        return out[:k]

    def drand48_after(self, skip, seed=None):
        self.lib.ref_drand48_after.restype = C.c_double
        self.lib.ref_drand48_after.argtypes = [C.c_int, C.c_long, C.c_long]
        return self.lib.ref_drand48_after(0 if seed is None else 1, seed or 0, skip)

    def logf(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        self.lib.ref_logf.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        self.lib.ref_logf(_p(x), len(x), _p(y))
        return y

    def logf_mismatches(self, oracle, lo, hi):
        """Scan every float with bits in [lo, hi): count of inputs where libm's logf and the oracle's restatement differ."""
        first = C.c_uint32()
        fn = C.cast(oracle.lib.lo_logf, C.c_void_p)
        self.lib.ref_logf_mismatches.restype = C.c_long
        self.lib.ref_logf_mismatches.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        return self.lib.ref_logf_mismatches(fn, lo, hi, C.byref(first)), first.value

    def adder(self, a, b):
        a = cf32(a); b = cf32(b)
        n = min(len(a), len(b))
        out = np.empty(n, np.complex64)
        self.lib.ref_adder.restype = C.c_long
        self.lib.ref_adder.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        k = self.lib.ref_adder(_p(a), _p(b), n, _p(out))
        return out[:k]

    def cconv_f32_u8(self, x):
        x = cf32(x)
        out = np.empty((len(x), 2), np.uint8)
        self.lib.ref_cconv_f32_u8.restype = C.c_long
        self.lib.ref_cconv_f32_u8.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        k = self.lib.ref_cconv_f32_u8(_p(x), len(x), _p(out))
        return out[:k]

    def cconv_f32_s16(self, x):
        x = cf32(x)
        out = np.empty((len(x), 2), np.int16)
        self.lib.ref_cconv_f32_s16.restype = C.c_long
        self.lib.ref_cconv_f32_s16.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        k = self.lib.ref_cconv_f32_s16(_p(x), len(x), _p(out))
        return out[:k]

    def hs_deconvol(self, symbols, resync_period=32):
        sym = np.ascontiguousarray(symbols, np.uint8)
        L = self.lib
        L.ref_hs_deconvol.restype = C.c_long
        L.ref_hs_deconvol.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        out = np.empty(len(sym) // 8 + 64, np.uint8)
        k = L.ref_hs_deconvol(resync_period, _p(sym), len(sym), _p(out), len(out))
        return out[:k].copy()

    def rx(self, params, x):
        x = cf32(x)
        cap = len(x) + 256
        cost = np.empty(cap, np.int16)
        sym = np.empty(cap, np.uint8)
        mcap = len(x) // max(1, params.meas_decimation) + 8
        fr, ss, mer = (np.empty(mcap, np.float32) for _ in range(3))
        ccap = len(x) // 128 + 8
        cst = np.empty(ccap, np.complex64)
        nm, nc = C.c_long(), C.c_long(ccap)
        st = RxState()
        n = self.lib.ref_cstln_receiver(C.byref(params), _p(x), len(x), _p(cost), _p(sym), cap,
                                        _p(fr), _p(ss), _p(mer), _p(cst), mcap, C.byref(nm), C.byref(nc),
                                        C.byref(st))
        out = np.zeros(n, SOFTSYM)
        out["cost"] = cost[:n]
        out["symbol"] = sym[:n]
        return dict(sym=out, freq=fr[:nm.value], ss=ss[:nm.value], mer=mer[:nm.value],
                    cstln=cst[:nc.value], state=st)


def have_ref():
    return os.path.exists(REF_SO)

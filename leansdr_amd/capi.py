"""ctypes driver for liblsdr_hip.so (include/lsdr_hip.h).  No CPU fallback."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSDR_HIP_LIB", os.path.join(HERE, "liblsdr_hip.so"))  # override: instrumented builds (tools/)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C leansdr_amd/csrc`). leansdr_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)

c_f, c_sz, vp = C.c_float, C.c_size_t, C.c_void_p
psz = C.POINTER(c_sz)

SOFTSYM = np.dtype([("cost", "<i2"), ("symbol", "u1"), ("pad", "u1")])

(BPSK, QPSK, PSK8, APSK16, APSK32, APSK64E, QAM16, QAM64, QAM256) = range(9)
CSTLN_BITS = {BPSK: 1, QPSK: 2, PSK8: 3, APSK16: 4, APSK32: 5, APSK64E: 6, QAM16: 4, QAM64: 6, QAM256: 8}
(FEC12, FEC23, FEC46, FEC34, FEC56, FEC78, FEC45, FEC89, FEC910) = range(9)
IN_CF32, IN_CU8 = 0, 1
SYM_SOFT, SYM_HARD2 = 0, 1
FIR_EXACT, FIR_FMA, FIR_MFMA, FIR_MFMA_BLK = 0, 1, 2, 3
SAMP_NEAREST, SAMP_LINEAR, SAMP_FIR = 0, 1, 2
RX_SERIAL, RX_TILED = 0, 1
NOTCH_EXACT, NOTCH_SCAN = 0, 1


class LsdrError(RuntimeError):
    pass


class FirCfg(C.Structure):
    _fields_ = [("ncoeffs", C.c_uint), ("coeffs_host", vp), ("decim", C.c_uint),
                ("in_format", C.c_int), ("in_scale", c_f), ("arith", C.c_int)]


class NotchFirCfg(C.Structure):
    _fields_ = [("ncoeffs", C.c_uint), ("coeffs_host", vp), ("decim", C.c_uint), ("in_scale", c_f), ("nslots", C.c_int),
                ("notch_decimation", C.c_int), ("k", c_f)]


class RxCfg(C.Structure):
    _fields_ = [("sampler", C.c_int), ("ncoeffs", C.c_int), ("coeffs_host", vp), ("subsampling", C.c_int),
                ("cstln", C.c_int), ("fec", C.c_int), ("harden", C.c_int), ("omega", c_f), ("freq", c_f),
                ("pll_adjustment", c_f), ("allow_drift", C.c_int), ("meas_decimation", C.c_ulong),
                ("kest", c_f), ("mode", C.c_int), ("tile_len", C.c_uint), ("tile_warmup", C.c_uint), ("in_format", C.c_int), ("out_format", C.c_int)]


class RxState(C.Structure):
    _fields_ = [("mu", c_f), ("phase", c_f), ("freqw", c_f), ("agc_gain", c_f), ("est_insp", c_f),
                ("est_sp", c_f), ("est_ep", c_f), ("freq_tap", c_f), ("min_freqw", c_f), ("max_freqw", c_f),
                ("meas_count", C.c_ulong), ("hist", c_f * 12)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "hist"}
        d["hist"] = list(self.hist)
        return d


def _sig(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


_sig("lsdr_abi_version", C.c_int, [])
# include/lsdr_hip.h LSDR_ABI_VERSION: 2 since lsdr_rx_run_multi_async writes one `consumed` PER receiver (round 5) — a caller
# built against version 1 passed one size_t there.  A stale liblsdr_hip.so is refused at import, not discovered through memory damage.
ABI_VERSION = 2
if lib.lsdr_abi_version() != ABI_VERSION:
    raise ImportError(f"liblsdr_hip.so reports ABI version {lib.lsdr_abi_version()}, leansdr_amd.capi is written for {ABI_VERSION}: "
                      "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
_sig("lsdr_last_error", C.c_char_p, [])
_sig("lsdr_device_count", C.c_int, [])
_sig("lsdr_device_pci_bus_id", C.c_int, [C.c_int, C.c_char_p, C.c_int])


def device_pci_bus_id(device):
    buf = C.create_string_buffer(32)
    check(lib.lsdr_device_pci_bus_id(device, buf, 32))
    return buf.value.decode()
_sig("lsdr_ctx_create", C.c_int, [C.c_int, vp, C.POINTER(vp)])
_sig("lsdr_ctx_create_masked", C.c_int, [C.c_int, vp, C.c_uint, C.POINTER(vp)])
_sig("lsdr_ctx_destroy", None, [vp])
_sig("lsdr_ctx_sync", C.c_int, [vp])
_sig("lsdr_ctx_stream", vp, [vp])
_sig("lsdr_malloc", C.c_int, [vp, c_sz, C.POINTER(vp)])
_sig("lsdr_free", C.c_int, [vp, vp])
_sig("lsdr_malloc_host", C.c_int, [c_sz, C.POINTER(vp)])
PROBE_FN = C.CFUNCTYPE(C.c_int, vp, vp)
_sig("lsdr_arena_create", C.c_int, [vp, c_sz, C.POINTER(vp)])
_sig("lsdr_arena_destroy", None, [vp])
_sig("lsdr_arena_bytes", c_sz, [vp])
_sig("lsdr_arena_owns", C.c_int, [vp, vp])
_sig("lsdr_arena_place", C.c_int, [vp, c_sz, C.c_uint, C.c_uint, C.c_int, vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_float)])
_sig("lsdr_arena_time", C.c_int, [vp, vp, vp, vp, C.POINTER(C.c_float)])
_sig("lsdr_arena_release", C.c_int, [vp, vp])
_sig("lsdr_arena_probe_log", C.c_int, [vp, C.POINTER(C.c_float), C.c_uint, C.POINTER(C.c_uint)])
_sig("lsdr_ctx_set_arena", C.c_int, [vp, vp])
_sig("lsdr_free_host", C.c_int, [vp])
_sig("lsdr_memcpy_h2d", C.c_int, [vp, vp, vp, c_sz])
_sig("lsdr_memcpy_d2h", C.c_int, [vp, vp, vp, c_sz])
_sig("lsdr_memcpy_d2d", C.c_int, [vp, vp, vp, c_sz])
_sig("lsdr_memset", C.c_int, [vp, vp, C.c_int, c_sz])
_sig("lsdr_timer_start", C.c_int, [vp])
_sig("lsdr_timer_stop_ms", C.c_int, [vp, C.POINTER(c_f)])
_sig("lsdr_event_create", C.c_int, [vp, C.POINTER(vp)])
_sig("lsdr_event_destroy", None, [vp])
_sig("lsdr_event_record", C.c_int, [vp])
_sig("lsdr_event_elapsed_ms", C.c_int, [vp, vp, C.POINTER(c_f)])
_sig("lsdr_ctx_wait_event", C.c_int, [vp, vp])
_sig("lsdr_filtergen_lowpass", C.c_int, [C.c_int, c_f, c_f, vp])
_sig("lsdr_filtergen_root_raised_cosine", C.c_int, [C.c_int, c_f, c_f, vp])
_sig("lsdr_filtergen_normalize_dcgain", None, [C.c_int, vp, c_f])
_sig("lsdr_filtergen_normalize_power", None, [C.c_int, vp, c_f])
_sig("lsdr_trig16_table", None, [vp])
_sig("lsdr_cstln_lut_build", C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_int)])
_sig("lsdr_cconverter_u8_run", C.c_int, [vp, vp, c_sz, vp])
_sig("lsdr_scaler_run", C.c_int, [vp, c_f, vp, c_sz, vp])
_sig("lsdr_decimator_run", C.c_int, [vp, C.c_uint, vp, c_sz, vp, c_sz, psz])
_sig("lsdr_auto_notch_create", C.c_int, [vp, C.c_int, c_f, C.POINTER(vp)])
_sig("lsdr_auto_notch_destroy", None, [vp])
_sig("lsdr_auto_notch_set", C.c_int, [vp, C.c_int, c_f])
_sig("lsdr_auto_notch_slot_bin", C.c_int, [vp, C.c_int])
_sig("lsdr_auto_notch_set_mode", C.c_int, [vp, C.c_int])
_sig("lsdr_auto_notch_stats", C.c_int, [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)])
_sig("lsdr_auto_notch_scan_time", C.c_int, [vp, C.c_int, C.POINTER(c_f), C.POINTER(C.c_uint)])
if hasattr(lib, "lsdr_auto_notch_debug_poison"):       # the measure build only (make -C leansdr_amd/csrc measure)
    _sig("lsdr_auto_notch_debug_poison", C.c_int, [vp])
_sig("lsdr_auto_notch_set_overlap", C.c_int, [vp, C.c_int])
_sig("lsdr_auto_notch_check", C.c_int, [vp, C.POINTER(C.c_uint)])
_sig("lsdr_auto_notch_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_notch_fir_create", C.c_int, [vp, C.POINTER(NotchFirCfg), C.POINTER(vp)])
_sig("lsdr_notch_fir_destroy", None, [vp])
_sig("lsdr_notch_fir_set", C.c_int, [vp, C.c_int, c_f])
_sig("lsdr_notch_fir_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_notch_fir_slot_bin", C.c_int, [vp])
_sig("lsdr_notch_fir_set_overlap", C.c_int, [vp, C.c_int])
_sig("lsdr_notch_fir_set_freq", C.c_int, [vp, c_f])
_sig("lsdr_notch_fir_track", C.c_int, [vp, c_f, c_f, c_f, C.POINTER(C.c_int)])
_sig("lsdr_notch_fir_current_freq", c_f, [vp])
_sig("lsdr_notch_fir_time", C.c_int, [vp, C.c_int, C.POINTER(c_f), C.POINTER(C.c_uint)])
_sig("lsdr_cfft_host", C.c_int, [C.c_int, vp, C.c_int])
_sig("lsdr_cnr_fft_create", C.c_int, [vp, c_f, C.c_int, C.POINTER(vp)])
_sig("lsdr_cnr_fft_destroy", None, [vp])
_sig("lsdr_cnr_fft_set", C.c_int, [vp, C.c_int, c_f])
_sig("lsdr_cnr_fft_run", C.c_int, [vp, c_f, c_f, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_mpeg_sync_set_resync_period", C.c_int, [vp, C.c_int])
_sig("lsdr_fastqpsk_create", C.c_int, [vp, c_f, c_f, c_f, C.c_int, C.c_ulong, C.POINTER(vp)])
_sig("lsdr_fastqpsk_destroy", None, [vp])
_sig("lsdr_fastqpsk_get_state", C.c_int, [vp, C.POINTER(c_f), C.POINTER(C.c_uint), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)])
_sig("lsdr_fastqpsk_set_tiled", C.c_int, [vp, C.c_int, C.c_uint, C.c_uint])
_sig("lsdr_fastqpsk_tiled_stats", C.c_int, [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)])
_sig("lsdr_fastqpsk_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz, vp, c_sz, psz, vp, c_sz, psz])
_sig("lsdr_hsdeconv_create", C.c_int, [vp, C.c_int, C.POINTER(vp)])
_sig("lsdr_hsdeconv_destroy", None, [vp])
_sig("lsdr_hsdeconv_locked", C.c_int, [vp])
_sig("lsdr_hsdeconv_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_cfft_run", C.c_int, [vp, C.c_int, C.c_int, vp, vp])
_sig("lsdr_randomizer_create", C.c_int, [vp, C.POINTER(vp)])
_sig("lsdr_randomizer_destroy", None, [vp])
_sig("lsdr_randomizer_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_rs_encoder_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_interleaver_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_convol_create", C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)])
_sig("lsdr_convol_destroy", None, [vp])
_sig("lsdr_convol_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_cstln_transmitter_run", C.c_int, [vp, C.c_int, C.c_int, vp, c_sz, vp])
_sig("lsdr_fir_resampler_create", C.c_int, [vp, C.c_uint, vp, C.c_uint, C.POINTER(vp)])
_sig("lsdr_fir_resampler_destroy", None, [vp])
_sig("lsdr_fir_resampler_set_freq", C.c_int, [vp, c_f])
_sig("lsdr_fir_resampler_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_simple_agc_create", C.c_int, [vp, c_f, c_f, C.POINTER(vp)])
_sig("lsdr_simple_agc_destroy", None, [vp])
_sig("lsdr_simple_agc_set", C.c_int, [vp, c_f, c_f])
_sig("lsdr_simple_agc_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_wgn_create", C.c_int, [vp, C.c_int, C.c_long, C.POINTER(vp)])
_sig("lsdr_wgn_destroy", None, [vp])
_sig("lsdr_wgn_get_state", C.c_int, [vp, C.POINTER(C.c_ulonglong)])
_sig("lsdr_wgn_set_state", C.c_int, [vp, C.c_ulonglong])
_sig("lsdr_wgn_run", C.c_int, [vp, c_f, vp, vp, c_sz])
_sig("lsdr_adder_run", C.c_int, [vp, vp, vp, c_sz, vp])
_sig("lsdr_cconverter_f32_u8_run", C.c_int, [vp, vp, c_sz, vp])
_sig("lsdr_cconverter_f32_s16_run", C.c_int, [vp, vp, c_sz, vp])
_sig("lsdr_drifter_create", C.c_int, [vp, C.POINTER(vp)])
_sig("lsdr_drifter_destroy", None, [vp])
_sig("lsdr_drifter_set_component", C.c_int, [vp, C.c_int, c_f, c_f])
_sig("lsdr_drifter_get_phases", C.c_int, [vp, C.c_longlong * 3])
_sig("lsdr_drifter_set_phases", C.c_int, [vp, C.c_longlong * 3])
_sig("lsdr_drifter_run", C.c_int, [vp, vp, c_sz, vp, c_sz])
_sig("lsdr_rotator_create", C.c_int, [vp, c_f, C.POINTER(vp)])
_sig("lsdr_rotator_destroy", None, [vp])
_sig("lsdr_rotator_run", C.c_int, [vp, vp, c_sz, vp])
_sig("lsdr_spectrum_create", C.c_int, [vp, C.POINTER(vp)])
_sig("lsdr_spectrum_destroy", None, [vp])
_sig("lsdr_spectrum_set", C.c_int, [vp, C.c_int, c_f])
_sig("lsdr_spectrum_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_fir_filter_create", C.c_int, [vp, C.POINTER(FirCfg), C.POINTER(vp)])
_sig("lsdr_fir_filter_destroy", None, [vp])
_sig("lsdr_fir_filter_set_freq", C.c_int, [vp, c_f])
_sig("lsdr_fir_filter_track", C.c_int, [vp, c_f, c_f, c_f, C.POINTER(C.c_int)])
_sig("lsdr_fir_filter_current_freq", c_f, [vp])
_sig("lsdr_fir_filter_get_shifted_coeffs", C.c_int, [vp, vp])
_sig("lsdr_fir_filter_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_fir_filter_run_multi", C.c_int, [vp, C.c_uint, C.POINTER(vp), c_sz, C.POINTER(vp), c_sz, psz, psz])
_sig("lsdr_rx_create", C.c_int, [vp, C.POINTER(RxCfg), C.POINTER(vp)])
_sig("lsdr_rx_destroy", None, [vp])
_sig("lsdr_rx_readahead", C.c_int, [vp])
_sig("lsdr_rx_get_state", C.c_int, [vp, C.POINTER(RxState)])
_sig("lsdr_rx_set_state", C.c_int, [vp, C.POINTER(RxState)])
_sig("lsdr_rx_tiled_stats", C.c_int, [vp] + [C.POINTER(C.c_uint)] * 4)
_sig("lsdr_deconv_create", C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)])
_sig("lsdr_deconv_destroy", None, [vp])
_sig("lsdr_deconv_next_sync", C.c_int, [vp])
_sig("lsdr_deconv_reset", C.c_int, [vp])
_sig("lsdr_mpeg_sync_reset", C.c_int, [vp])
_sig("lsdr_derandomizer_reset", C.c_int, [vp])
_sig("lsdr_rx_reset", C.c_int, [vp])
_sig("lsdr_rx_tile_time", C.c_int, [vp, C.c_int, C.POINTER(c_f), C.POINTER(C.c_uint)])
_sig("lsdr_deconv_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_deconv_run_hs2", C.c_int, [vp, vp, c_sz, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_viterbi_create", C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)])
_sig("lsdr_viterbi_destroy", None, [vp])
_sig("lsdr_viterbi_set_resync_period", C.c_int, [vp, C.c_int])
_sig("lsdr_viterbi_current_sync", C.c_int, [vp])
_sig("lsdr_viterbi_stats", C.c_int, [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)])
_sig("lsdr_viterbi_repair_stats", C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)])
_sig("lsdr_viterbi_q4_supported", C.c_int, [C.c_int, C.c_int])
_sig("lsdr_viterbi_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_mpeg_sync_create", C.c_int, [vp, C.c_int, C.POINTER(vp)])
_sig("lsdr_mpeg_sync_destroy", None, [vp])
_sig("lsdr_mpeg_sync_locked", C.c_int, [vp])
_sig("lsdr_mpeg_sync_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz, vp, C.POINTER(C.c_int), C.POINTER(C.c_ulong), C.POINTER(C.c_int)])
_sig("lsdr_deinterleaver_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_rs_decoder_run", C.c_int, [vp, vp, c_sz, vp, C.POINTER(C.c_long), C.POINTER(C.c_long)])
_sig("lsdr_derandomizer_create", C.c_int, [vp, C.POINTER(vp)])
_sig("lsdr_derandomizer_destroy", None, [vp])
_sig("lsdr_derandomizer_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_derandomizer_pattern", None, [vp])
_sig("lsdr_rs_tables", None, [vp, vp, vp])
_sig("lsdr_rx_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz, vp, vp, vp, c_sz, psz, vp, c_sz, psz])
_sig("lsdr_rx_run_async", C.c_int, [vp, vp, c_sz, vp, c_sz, psz])
_sig("lsdr_rx_run_multi_async", C.c_int, [C.POINTER(vp), C.c_uint, C.POINTER(vp), c_sz, C.POINTER(vp), c_sz, psz])
_sig("lsdr_rx_run_async_hs2", C.c_int, [vp, vp, c_sz, vp, c_sz, c_sz, psz])
_sig("lsdr_rx_wait", C.c_int, [vp, psz])
_sig("lsdr_rx_retired_freq_tap", C.c_float, [vp])
_sig("lsdr_fec_spec", C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), vp])
_sig("lsdr_cconverter_int_run", C.c_int, [vp, C.c_int, vp, c_sz, vp])
_sig("lsdr_copy_h2d_async", C.c_int, [vp, vp, vp, c_sz])
_sig("lsdr_copy_d2h_async", C.c_int, [vp, vp, vp, c_sz])
_sig("lsdr_copy_fence", C.c_int, [vp])
_sig("lsdr_copy_sync_d2h", C.c_int, [vp])
_sig("lsdr_copy_sync_all", C.c_int, [vp])
_sig("lsdr_rx_batch_create", C.c_int, [vp, C.POINTER(RxCfg), C.c_uint, C.POINTER(vp)])
_sig("lsdr_rx_batch_destroy", None, [vp])
_sig("lsdr_rx_batch_run", C.c_int, [vp, vp, c_sz, vp, c_sz, psz, psz])
_sig("lsdr_rx_batch_get_state", C.c_int, [vp, C.c_uint, C.POINTER(RxState)])
_sig("lsdr_rx_decision_mode", C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint)])
_sig("lsdr_rx_snapshot_async", C.c_int, [vp])
_sig("lsdr_rx_get_snapshot", C.c_int, [vp, C.POINTER(RxState)])
_sig("lsdr_rx_snapshot_async_slot", C.c_int, [vp, C.c_uint])
_sig("lsdr_rx_get_snapshot_slot", C.c_int, [vp, C.c_uint, C.POINTER(RxState)])


class CaptureBatchCfg(C.Structure):
    _fields_ = [("n_captures", C.c_int), ("max_samples", C.c_size_t), ("omega", C.c_float), ("fec", C.c_int), ("anf", C.c_int),
                ("tile_len", C.c_uint), ("tile_warmup", C.c_uint), ("notch_k", C.c_float), ("notch_decimation", C.c_int),
                ("unlocked_window", C.c_uint), ("aux_cus", C.c_uint)]


class CaptureResult(C.Structure):
    _fields_ = [("ts_packets", C.c_uint64), ("rs_packets", C.c_uint64), ("rs_bit_errors", C.c_uint64), ("symbols", C.c_uint64),
                ("samples", C.c_uint64), ("bytes_deconv", C.c_uint64), ("bytes_mpeg", C.c_uint64), ("first_lock_byte", C.c_uint64),
                ("next_sync_calls", C.c_uint32), ("locked", C.c_uint32), ("alignment", C.c_uint32), ("bitphase", C.c_uint32),
                ("tiles", C.c_uint32), ("seam_dup", C.c_uint32), ("seam_miss", C.c_uint32), ("seam_bad", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_sig("lsdr_capture_batch_create", C.c_int, [vp, C.POINTER(CaptureBatchCfg), C.POINTER(vp)])
_sig("lsdr_capture_batch_destroy", None, [vp])
_sig("lsdr_capture_batch_run_async", C.c_int, [vp, vp, c_sz])
_sig("lsdr_capture_batch_wait", C.c_int, [vp, vp])
_sig("lsdr_capture_batch_ts_download_async", C.c_int, [vp, vp, c_sz])
_sig("lsdr_capture_batch_ts_wait", C.c_int, [vp])
_sig("lsdr_capture_batch_ts_dev", vp, [vp, C.c_int])
_sig("lsdr_capture_batch_words_dev", vp, [vp, C.c_int])
_sig("lsdr_capture_batch_bytes_dev", vp, [vp, C.c_int])
_sig("lsdr_capture_batch_mpeg_dev", vp, [vp, C.c_int])
_sig("lsdr_capture_batch_bins", C.c_int, [vp, C.c_int, vp, C.c_uint, C.POINTER(C.c_uint)])
_sig("lsdr_capture_batch_notched", C.c_int, [vp, C.c_int, vp, c_sz])
_sig("lsdr_capture_batch_tile_time", C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint)])

#: every symbol include/lsdr_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [n for n in dir(lib) if n.startswith("lsdr_")]


def check(rc):
    if rc != 0:
        raise LsdrError(f"lsdr error {rc}: {lib.lsdr_last_error().decode()}")


def _np(a):
    return a.ctypes.data_as(vp)


# ---- host-side table design (no GPU needed) --------------------------------
def lowpass(order, fcut, renormalize=True):
    """filtergen::lowpass (+ the extra normalize_dcgain of leandvb.cc:377-378)."""
    out = np.empty(order + 1, np.float32)
    n = lib.lsdr_filtergen_lowpass(order, fcut, 1.0, _np(out))
    if renormalize:
        lib.lsdr_filtergen_normalize_dcgain(n, _np(out), 1.0)
    return out[:n]


def root_raised_cosine(order, fs, rolloff):
    out = np.empty(order + 3, np.float32)
    n = lib.lsdr_filtergen_root_raised_cosine(order, fs, rolloff, _np(out))
    return out[:n].copy()


def normalize_power(coeffs, gain):
    c = np.ascontiguousarray(coeffs, np.float32).copy()
    lib.lsdr_filtergen_normalize_power(len(c), _np(c), gain)
    return c


def trig16():
    out = np.empty(65536, np.complex64)
    lib.lsdr_trig16_table(_np(out))
    return out


def cstln_lut(predef, fec=0):
    cost = np.empty(65536, np.int16)
    sym = np.empty(65536, np.uint8)
    pe = np.empty(65536, np.int16)
    symbols = np.zeros((256, 2), np.int8)
    nrot = C.c_int()
    n = lib.lsdr_cstln_lut_build(predef, fec, _np(cost), _np(sym), _np(pe), _np(symbols), C.byref(nrot))
    if n < 0:
        check(n)
    return dict(nsymbols=n, nrotations=nrot.value, symbols=symbols[:n].copy(), cost=cost, symbol=sym,
                phase_error=pe)


# ---- device side ------------------------------------------------------------
class DevBuf:
    """A device allocation (what a device-resident pipebuf owns)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = vp()
        check(lib.lsdr_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            check(lib.lsdr_free(self.ctx.h, self.ptr))
            self.ptr = None

    def at(self, byte_offset):
        return vp(self.ptr + int(byte_offset))


class ArenaWindow:
    """A window of an Arena, with what a pipeline uses of a DevBuf; free() gives it back to the arena."""

    def __init__(self, arena, ptr, nbytes, probe_ms=None):
        self.arena, self.ptr, self.nbytes, self.probe_ms = arena, int(ptr), int(nbytes), probe_ms

    def at(self, byte_offset):
        return vp(self.ptr + int(byte_offset))

    def free(self):
        if self.ptr and self.arena.h:
            check(lib.lsdr_arena_release(self.arena.h, vp(self.ptr)))
        self.ptr = None


class Arena:
    """lsdr_arena (include/lsdr_hip.h): one large device allocation handed out in windows, the fastest ones first."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_arena_create(ctx.h, int(nbytes), C.byref(h)))
        self.h = h
        self.nbytes = int(lib.lsdr_arena_bytes(h))

    def place(self, nbytes, n_best=1, max_windows=64, from_tail=False, fill_from=None, probe=None):
        """probe(window_ptr: int) queues the launch(es) to be timed on the context's stream (None: the built-in streaming read).
        Returns the n_best windows, fastest first (.probe_ms = ms per probe call)."""
        out = (vp * n_best)()
        ms = (C.c_float * n_best)()
        err = []

        def _cb(user, window):
            try:
                probe(int(window))
                return 0
            except BaseException as e:      # never unwinds through the C caller
                err.append(e)
                return -2
        cb = PROBE_FN(_cb) if probe is not None else C.cast(None, PROBE_FN)
        rc = lib.lsdr_arena_place(self.h, int(nbytes), n_best, max_windows, int(bool(from_tail)), vp(fill_from) if fill_from else None,
                                  C.cast(cb, vp), None, out, ms)
        if err:
            raise err[0]
        check(rc)
        return [ArenaWindow(self, out[k], nbytes, float(ms[k])) for k in range(n_best)]

    def time(self, ptr, probe):
        """ms per call of probe(ptr) over any device pointer, timed like a candidate window (3 untimed calls, 6 timed)."""
        ms = C.c_float()
        err = []

        def _cb(user, window):
            try:
                probe(int(window))
                return 0
            except BaseException as e:
                err.append(e)
                return -2
        cb = PROBE_FN(_cb)
        rc = lib.lsdr_arena_time(self.h, vp(int(ptr)), C.cast(cb, vp), None, C.byref(ms))
        if err:
            raise err[0]
        check(rc)
        return float(ms.value)

    def probe_log(self):
        n = C.c_uint()
        buf = (C.c_float * 4096)()
        check(lib.lsdr_arena_probe_log(self.h, buf, 4096, C.byref(n)))
        return [float(buf[i]) for i in range(min(n.value, 4096))]

    def attach(self, on=True):
        """lsdr_ctx_set_arena: lsdr_malloc of 1 MiB or more on this context is served from the arena from now on."""
        check(lib.lsdr_ctx_set_arena(self.ctx.h, self.h if on else None))

    def close(self):
        if self.h:
            lib.lsdr_ctx_set_arena(self.ctx.h, None)
            lib.lsdr_arena_destroy(self.h)
            self.h = None


class Ctx:
    def __init__(self, device=0, stream=None, cu_mask=None):
        """cu_mask: iterable of CU indices this context's stream may use (None: the whole GPU)."""
        h = vp()
        if cu_mask is None:
            check(lib.lsdr_ctx_create(device, stream, C.byref(h)))
        else:
            cus = sorted(set(int(c) for c in cu_mask))
            words = (max(cus) // 32 + 1) if cus else 1
            m = (C.c_uint32 * words)()
            for c in cus:
                m[c // 32] |= 1 << (c % 32)
            check(lib.lsdr_ctx_create_masked(device, m, words, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(lib.lsdr_ctx_sync(self.h))

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        b = DevBuf(self, max(arr.nbytes, 1))
        check(lib.lsdr_memcpy_h2d(self.h, b.ptr, _np(arr), arr.nbytes))
        self.sync()
        return b

    def download(self, buf, dtype, count, byte_offset=0):
        out = np.empty(count, dtype)
        check(lib.lsdr_memcpy_d2h(self.h, _np(out), buf.at(byte_offset), out.nbytes))
        self.sync()
        return out

    def event(self):
        e = vp()
        check(lib.lsdr_event_create(self.h, C.byref(e)))
        return e

    @staticmethod
    def event_record(e):
        check(lib.lsdr_event_record(e))

    @staticmethod
    def event_elapsed_ms(a, b):
        ms = c_f()
        check(lib.lsdr_event_elapsed_ms(a, b, C.byref(ms)))
        return ms.value

    def wait_event(self, e):
        check(lib.lsdr_ctx_wait_event(self.h, e))

    def timer_start(self):
        check(lib.lsdr_timer_start(self.h))

    def timer_stop_ms(self):
        ms = c_f()
        check(lib.lsdr_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    # elementwise blocks on numpy arrays (test convenience)
    def cconverter_u8(self, u8):
        u8 = np.ascontiguousarray(u8, np.uint8).reshape(-1, 2)
        din = self.upload(u8)
        dout = self.alloc(len(u8) * 8)
        check(lib.lsdr_cconverter_u8_run(self.h, din.ptr, len(u8), dout.ptr))
        out = self.download(dout, np.complex64, len(u8))
        din.free(); dout.free()
        return out

    def scaler(self, scale, x):
        x = np.ascontiguousarray(x, np.complex64)
        din = self.upload(x)
        dout = self.alloc(x.nbytes)
        check(lib.lsdr_scaler_run(self.h, scale, din.ptr, len(x), dout.ptr))
        out = self.download(dout, np.complex64, len(x))
        din.free(); dout.free()
        return out

    def decimator(self, d, x):
        x = np.ascontiguousarray(x, np.complex64)
        din = self.upload(x)
        dout = self.alloc(x.nbytes)
        prod = c_sz()
        check(lib.lsdr_decimator_run(self.h, d, din.ptr, len(x), dout.ptr, len(x), C.byref(prod)))
        out = self.download(dout, np.complex64, prod.value)
        din.free(); dout.free()
        return out


class FirFilter:
    """fir_filter<cf32,float> (dsp.h:219-285) on the GPU."""

    def __init__(self, ctx, coeffs, decim=1, in_format=IN_CF32, in_scale=0.0, arith=FIR_EXACT):
        self.ctx = ctx
        self.coeffs = np.ascontiguousarray(coeffs, np.float32)
        self.decim = decim
        self.in_format = in_format
        cfg = FirCfg(len(self.coeffs), self.coeffs.ctypes.data, decim, in_format, in_scale, arith)
        h = vp()
        check(lib.lsdr_fir_filter_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_fir_filter_destroy(self.h)
            self.h = None

    def set_freq(self, f):
        check(lib.lsdr_fir_filter_set_freq(self.h, f))

    def track(self, freq_tap, tap_multiplier, freq_tol):
        s = C.c_int()
        check(lib.lsdr_fir_filter_track(self.h, freq_tap, tap_multiplier, freq_tol, C.byref(s)))
        return bool(s.value)

    @property
    def current_freq(self):
        return lib.lsdr_fir_filter_current_freq(self.h)

    def shifted_coeffs(self):
        out = np.empty(len(self.coeffs), np.complex64)
        check(lib.lsdr_fir_filter_get_shifted_coeffs(self.h, _np(out)))
        return out

    def run_dev(self, in_ptr, n_in, out_ptr, cap_out):
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_fir_filter_run(self.h, in_ptr, n_in, out_ptr, cap_out, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run_multi_dev(self, in_ptrs, n_in, out_ptrs, cap_out):
        """The same filter over several equal-length device buffers in one launch; returns (consumed, produced) per buffer."""
        n = len(in_ptrs)
        ins = (vp * n)(*[vp(p) if not isinstance(p, vp) else p for p in in_ptrs])
        outs = (vp * n)(*[vp(p) if not isinstance(p, vp) else p for p in out_ptrs])
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_fir_filter_run_multi(self.h, n, ins, n_in, outs, cap_out, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run(self, x):
        """numpy in → (numpy out, consumed)."""
        if self.in_format == IN_CU8:
            x = np.ascontiguousarray(x, np.uint8).reshape(-1, 2)
        else:
            x = np.ascontiguousarray(x, np.complex64)
        n = len(x)
        cap = max(0, (n - len(self.coeffs)) // self.decim) + 1
        din = self.ctx.upload(x)
        dout = self.ctx.alloc(cap * 8)
        cons, prod = self.run_dev(din.ptr, n, dout.ptr, cap)
        out = self.ctx.download(dout, np.complex64, prod)
        din.free(); dout.free()
        return out, cons


class CstlnReceiver:
    """cstln_receiver<f32> (sdr.h:697-938) on the GPU."""

    def __init__(self, ctx, sampler=SAMP_LINEAR, coeffs=None, subsampling=1, cstln=QPSK, fec=FEC12, harden=0,
                 omega=4.0, freq=0.0, pll_adjustment=1.0, allow_drift=0, meas_decimation=1048576, kest=0.01,
                 mode=RX_SERIAL, tile_len=0, tile_warmup=0, in_format=0, out_format=0):
        """in_format = IN_CU8: cconverter<u8,128,f32,0,1,1> fused into the receiver's loads (input = cu8 items).
        out_format = SYM_HARD2 (tiled QPSK on cu8 only): packed 2-bit decisions instead of soft symbols."""
        self.ctx = ctx
        cfg = RxCfg()
        cfg.in_format, cfg.out_format = in_format, out_format
        cfg.sampler = sampler
        if coeffs is not None:
            self.coeffs = np.ascontiguousarray(coeffs, np.float32)
            cfg.ncoeffs = len(self.coeffs)
            cfg.coeffs_host = self.coeffs.ctypes.data
        cfg.subsampling = subsampling
        cfg.cstln, cfg.fec, cfg.harden = cstln, fec, harden
        cfg.omega, cfg.freq, cfg.pll_adjustment = omega, freq, pll_adjustment
        cfg.allow_drift, cfg.meas_decimation, cfg.kest = allow_drift, meas_decimation, kest
        cfg.mode, cfg.tile_len, cfg.tile_warmup = mode, tile_len, tile_warmup
        self.cfg = cfg
        h = vp()
        check(lib.lsdr_rx_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_rx_destroy(self.h)
            self.h = None

    @property
    def readahead(self):
        return lib.lsdr_rx_readahead(self.h)

    def state(self):
        st = RxState()
        check(lib.lsdr_rx_get_state(self.h, C.byref(st)))
        return st

    def set_state(self, st):
        check(lib.lsdr_rx_set_state(self.h, C.byref(st)))

    def reset(self):
        """The loop state right after construction: the next run starts a new capture."""
        check(lib.lsdr_rx_reset(self.h))

    def tile_time(self, enable):
        """(mean ms, launches) of the k_rx_tiles launches of the runs retired since the last call; then set the switch."""
        ms, n = c_f(), C.c_uint()
        check(lib.lsdr_rx_tile_time(self.h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def decision_mode(self):
        a, d = C.c_int(), C.c_uint()
        check(lib.lsdr_rx_decision_mode(self.h, C.byref(a), C.byref(d)))
        return dict(arithmetic=bool(a.value), max_phase_error_delta=d.value)

    def snapshot_async(self, slot=0):
        """Copy the device-side loop state into one of the receiver's four pinned slots, in stream order (between queued runs)."""
        check(lib.lsdr_rx_snapshot_async_slot(self.h, slot))

    def snapshot(self, slot=0):
        st = RxState()
        check(lib.lsdr_rx_get_snapshot_slot(self.h, slot, C.byref(st)))
        return st

    def tiled_stats(self):
        v = [C.c_uint() for _ in range(4)]
        check(lib.lsdr_rx_tiled_stats(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("tiles", "dup", "miss", "bad_seams"), [x.value for x in v]))

    def run_async(self, in_ptr, n_in, out_ptr, cap_out):
        """Queue one tiled run on the receiver's stream; returns the samples it will consume."""
        cons = c_sz()
        check(lib.lsdr_rx_run_async(self.h, in_ptr, n_in, out_ptr, cap_out, C.byref(cons)))
        return cons.value

    @staticmethod
    def run_multi_async(rxs, in_ptrs, n_in, out_ptrs, cap_out):
        """lsdr_rx_run_multi_async: one tiled run of every receiver in `rxs` (equally long inputs) with shared launches; returns the
        LIST of samples each receiver will consume.  Each receiver is retired with its own wait()."""
        n = len(rxs)
        hs = (vp * n)(*[r.h for r in rxs])
        ins = (vp * n)(*[p if isinstance(p, vp) else vp(p) for p in in_ptrs])
        outs = (vp * n)(*[p if isinstance(p, vp) else vp(p) for p in out_ptrs])
        cons = (c_sz * n)()
        check(lib.lsdr_rx_run_multi_async(hs, n, ins, n_in, outs, cap_out, cons))
        return [int(v) for v in cons]

    def run_async_hs2(self, in_ptr, n_in, out_ptr, out_sym_offset, cap_out):
        """Queue one tiled run of a SYM_HARD2 receiver writing its packed symbols from symbol out_sym_offset of `out_ptr`."""
        cons = c_sz()
        check(lib.lsdr_rx_run_async_hs2(self.h, in_ptr, n_in, out_ptr, out_sym_offset, cap_out, C.byref(cons)))
        return cons.value

    def wait(self):
        """Retire the oldest queued run; returns its symbol count."""
        prod = c_sz()
        check(lib.lsdr_rx_wait(self.h, C.byref(prod)))
        return prod.value

    @property
    def retired_freq_tap(self):
        """freq_tap after the most recently retired queued run (cycles per sample)."""
        return lib.lsdr_rx_retired_freq_tap(self.h)

    def run_dev(self, in_ptr, n_in, out_ptr, cap_out, meas=True):
        cons, prod, nm, nc = c_sz(), c_sz(), c_sz(), c_sz()
        if meas:
            mcap = n_in // max(1, self.cfg.meas_decimation) + 8
            ccap = n_in // 128 + 8
            fr, ss, mer = (np.empty(mcap, np.float32) for _ in range(3))
            cst = np.empty(ccap, np.complex64)
            check(lib.lsdr_rx_run(self.h, in_ptr, n_in, out_ptr, cap_out, C.byref(cons), C.byref(prod),
                                  _np(fr), _np(ss), _np(mer), mcap, C.byref(nm), _np(cst), ccap, C.byref(nc)))
            return dict(consumed=cons.value, produced=prod.value, freq=fr[:nm.value], ss=ss[:nm.value],
                        mer=mer[:nm.value], cstln=cst[:nc.value])
        check(lib.lsdr_rx_run(self.h, in_ptr, n_in, out_ptr, cap_out, C.byref(cons), C.byref(prod),
                              None, None, None, 0, None, None, 0, None))
        return dict(consumed=cons.value, produced=prod.value)

    def run(self, x, meas=True):
        """x: complex64 samples, or (in_format = IN_CU8) a uint8 array of interleaved re, im."""
        if self.cfg.in_format == IN_CU8:
            x = np.ascontiguousarray(x, np.uint8).reshape(-1, 2)
        else:
            x = np.ascontiguousarray(x, np.complex64)
        cap = len(x) + 256
        din = self.ctx.upload(x)
        dout = self.ctx.alloc(cap * 4)
        r = self.run_dev(din.ptr, len(x), dout.ptr, cap, meas)
        r["sym"] = self.ctx.download(dout, SOFTSYM, r["produced"])
        r["state"] = self.state()
        din.free(); dout.free()
        return r


class RxBatch:
    """Exact cstln_receiver<f32>, one GPU lane per independent capture (lsdr_rx_batch_*)."""

    def __init__(self, ctx, n_streams, sampler=SAMP_LINEAR, cstln=QPSK, fec=FEC12, omega=4.0, freq=0.0, pll_adjustment=1.0,
                 allow_drift=0, meas_decimation=1048576, kest=0.01, in_format=0):
        self.ctx, self.n = ctx, n_streams
        cfg = RxCfg()
        cfg.in_format = in_format
        cfg.sampler, cfg.cstln, cfg.fec = sampler, cstln, fec
        cfg.omega, cfg.freq, cfg.pll_adjustment = omega, freq, pll_adjustment
        cfg.allow_drift, cfg.meas_decimation, cfg.kest = allow_drift, meas_decimation, kest
        cfg.subsampling = 1
        h = vp()
        check(lib.lsdr_rx_batch_create(ctx.h, C.byref(cfg), n_streams, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_rx_batch_destroy(self.h)
            self.h = None

    def run_dev(self, in_ptrs, n_in, out_ptrs, cap_out):
        ins = (vp * self.n)(*[p if isinstance(p, vp) else vp(p) for p in in_ptrs])
        outs = (vp * self.n)(*[p if isinstance(p, vp) else vp(p) for p in out_ptrs])
        cons = c_sz()
        prod = (c_sz * self.n)()
        check(lib.lsdr_rx_batch_run(self.h, ins, n_in, outs, cap_out, C.byref(cons), prod))
        return cons.value, list(prod)

    def state(self, stream):
        st = RxState()
        check(lib.lsdr_rx_batch_get_state(self.h, stream, C.byref(st)))
        return st


class CaptureBatch:
    """lsdr_capture_batch: B independent cu8 captures, each from its first sample to TS (leandvb's default `--u8` graph per capture),
    in shared launches with the counts on the device."""

    def __init__(self, ctx, n_captures, max_samples, omega, fec=FEC12, anf=1, tile_len=0, tile_warmup=0, notch_k=0.0, notch_decimation=0,
                 unlocked_window=0, aux_cus=0):
        self.ctx, self.n = ctx, int(n_captures)
        cfg = CaptureBatchCfg()
        cfg.n_captures, cfg.max_samples, cfg.omega, cfg.fec, cfg.anf = self.n, int(max_samples), omega, fec, anf
        cfg.tile_len, cfg.tile_warmup, cfg.notch_k, cfg.notch_decimation = tile_len, tile_warmup, notch_k, notch_decimation
        cfg.unlocked_window = unlocked_window
        cfg.aux_cus = aux_cus
        self.unlocked_window = unlocked_window or 8192
        h = vp()
        check(lib.lsdr_capture_batch_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self._res = (CaptureResult * self.n)()

    def close(self):
        if self.h:
            lib.lsdr_capture_batch_destroy(self.h)
            self.h = None

    def run_async(self, iq_ptrs, n_samples):
        ins = (vp * self.n)(*[p if isinstance(p, vp) else vp(p) for p in iq_ptrs])
        check(lib.lsdr_capture_batch_run_async(self.h, ins, int(n_samples)))

    def wait(self, results=True):
        check(lib.lsdr_capture_batch_wait(self.h, self._res if results else None))
        return [r.as_dict() for r in self._res] if results else None

    def ts_download_async(self, host_ptrs, cap_bytes):
        outs = (vp * self.n)(*[p if isinstance(p, vp) else vp(p) for p in host_ptrs])
        check(lib.lsdr_capture_batch_ts_download_async(self.h, outs, int(cap_bytes)))

    def ts_wait(self):
        check(lib.lsdr_capture_batch_ts_wait(self.h))

    def decode(self, iq_ptrs, n_samples):
        """One batch, synchronously: (results, [TS bytes per capture])."""
        self.run_async(iq_ptrs, n_samples)
        res = self.wait()
        out = []
        for i, r in enumerate(res):
            nb = r["ts_packets"] * 188
            a = np.empty(nb, np.uint8)
            if nb:
                check(lib.lsdr_memcpy_d2h(self.ctx.h, _np(a), lib.lsdr_capture_batch_ts_dev(self.h, i), nb))
            out.append(a)
        self.ctx.sync()
        return res, [a.tobytes() for a in out]

    def words(self, i, nsym):
        """The packed decisions of capture i after a run, unpacked (host)."""
        nw = (int(nsym) + 15) // 16
        w = np.empty(nw, np.uint32)
        if nw:
            check(lib.lsdr_memcpy_d2h(self.ctx.h, _np(w), lib.lsdr_capture_batch_words_dev(self.h, i), nw * 4))
            self.ctx.sync()
        return hs2_unpack(w, int(nsym))

    def stage_bytes(self, i, which, n):
        fn = lib.lsdr_capture_batch_bytes_dev if which == "deconv" else lib.lsdr_capture_batch_mpeg_dev
        a = np.empty(int(n), np.uint8)
        if n:
            check(lib.lsdr_memcpy_d2h(self.ctx.h, _np(a), fn(self.h, i), int(n)))
            self.ctx.sync()
        return a

    def bins(self, i, cap=4096):
        b = (C.c_int * cap)()
        n = C.c_uint()
        check(lib.lsdr_capture_batch_bins(self.h, i, b, cap, C.byref(n)))
        return list(b[:min(n.value, cap)])

    def notched(self, i, n):
        """The notched stream of capture i as the tiles of the last run saw it (test hook), n complex64 items."""
        d = self.ctx.alloc(int(n) * 8)
        check(lib.lsdr_capture_batch_notched(self.h, i, d.ptr, int(n)))
        out = self.ctx.download(d, np.complex64, int(n))
        d.free()
        return out

    def tile_time(self, enable):
        ms, n = C.c_float(), C.c_uint()
        check(lib.lsdr_capture_batch_tile_time(self.h, 1 if enable else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def hs2_pack(symbols, offset=0):
    """Hard symbols (0..3) → the packed "hs2" stream (16 per uint32, MSB first), starting at symbol `offset` of word 0."""
    s = np.concatenate([np.zeros(offset, np.uint8), np.asarray(symbols, np.uint8) & 3])
    s = np.concatenate([s, np.zeros((-len(s)) % 16, np.uint8)]).reshape(-1, 16).astype(np.uint32)
    sh = (30 - 2 * np.arange(16)).astype(np.uint32)
    return np.bitwise_or.reduce(s << sh, axis=1).astype(np.uint32)


def hs2_unpack(words, n, offset=0):
    w = np.asarray(words, np.uint32)
    sh = (30 - 2 * np.arange(16)).astype(np.uint32)
    return ((w[:, None] >> sh) & 3).astype(np.uint8).reshape(-1)[offset:offset + n]


# ---- FEC tail -----------------------------------------------------------------------
def derandomizer_pattern():
    out = np.empty(1504, np.uint8)
    lib.lsdr_derandomizer_pattern(_np(out))
    return out


def rs_tables():
    e, l, g = np.empty(512, np.uint8), np.empty(256, np.uint8), np.empty(17, np.uint8)
    lib.lsdr_rs_tables(_np(e), _np(l), _np(g))
    return e, l, g


class Deconv:
    """deconvol_sync<u8,0> (dvb.h:122-513) on the GPU."""

    def __init__(self, ctx, rate=FEC12, fastlock=0):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_deconv_create(ctx.h, rate, fastlock, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_deconv_destroy(self.h)
            self.h = None

    def next_sync(self):
        check(lib.lsdr_deconv_next_sync(self.h))

    def reset(self):
        check(lib.lsdr_deconv_reset(self.h))

    def run_dev(self, in_ptr, n_in, out_ptr, cap):
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_deconv_run(self.h, in_ptr, n_in, out_ptr, cap, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run_dev_hs2(self, words_ptr, sym_offset, n_in, out_ptr, cap):
        """The same on packed hard symbols (SYM_HARD2): symbols [sym_offset, sym_offset + n_in) of the stream at words_ptr."""
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_deconv_run_hs2(self.h, words_ptr, sym_offset, n_in, out_ptr, cap, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run_stream(self, sym, pipe=4096, room=8192):
        """Whole stream through repeated run() calls on windows of `pipe` symbols (like the reference pipes)."""
        sym = np.ascontiguousarray(sym, SOFTSYM)
        din = self.ctx.upload(sym)
        dout = self.ctx.alloc(len(sym) + 64)
        pos, nout = 0, 0
        while True:
            avail = min(pipe, len(sym) - pos)
            cons, prod = self.run_dev(din.at(pos * 4), avail, dout.at(nout), room)
            if not cons and not prod:
                break
            pos += cons
            nout += prod
        out = self.ctx.download(dout, np.uint8, nout)
        din.free(); dout.free()
        return out


class Viterbi:
    """viterbi_sync (dvb.h:1173-1416) on the GPU."""

    def __init__(self, ctx, cstln=QPSK, rate=FEC12, resync_period=0):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_viterbi_create(ctx.h, cstln, rate, C.byref(h)))
        self.h = h
        if resync_period:
            check(lib.lsdr_viterbi_set_resync_period(h, resync_period))

    def close(self):
        if self.h:
            lib.lsdr_viterbi_destroy(self.h)
            self.h = None

    @property
    def current_sync(self):
        return lib.lsdr_viterbi_current_sync(self.h)

    def stats(self):
        t, b = C.c_uint(), C.c_uint()
        check(lib.lsdr_viterbi_stats(self.h, C.byref(t), C.byref(b)))
        return dict(tiles=t.value, bad_seams=b.value)

    def repair_stats(self):
        d, h = C.c_ulonglong(), C.c_ulonglong()
        check(lib.lsdr_viterbi_repair_stats(self.h, C.byref(d), C.byref(h)))
        return dict(device_repaired=d.value, host_rounds=h.value)

    def run_dev(self, in_ptr, n_in, out_ptr, cap):
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_viterbi_run(self.h, in_ptr, n_in, out_ptr, cap, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run_stream(self, sym, pipe=None):
        """Whole stream; repeated run() calls (a call stops early when the alignment switches)."""
        sym = np.ascontiguousarray(sym, SOFTSYM)
        din = self.ctx.upload(sym)
        dout = self.ctx.alloc(len(sym) + 64)
        pos, nout = 0, 0
        while True:
            avail = len(sym) - pos if pipe is None else min(pipe, len(sym) - pos)
            cons, prod = self.run_dev(din.at(pos * 4), avail, dout.at(nout), len(sym) + 64 - nout)
            if not cons and not prod:
                break
            pos += cons
            nout += prod
        out = self.ctx.download(dout, np.uint8, nout)
        din.free(); dout.free()
        return out, pos


class MpegSync:
    """mpeg_sync<u8,0> (dvb.h:712-891) on the GPU."""

    def __init__(self, ctx, fastlock=0):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_mpeg_sync_create(ctx.h, fastlock, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_mpeg_sync_destroy(self.h)
            self.h = None

    @property
    def locked(self):
        return bool(lib.lsdr_mpeg_sync_locked(self.h))

    def set_resync_period(self, period):
        check(lib.lsdr_mpeg_sync_set_resync_period(self.h, period))

    def reset(self):
        check(lib.lsdr_mpeg_sync_reset(self.h))

    def run_dev(self, in_ptr, n_in, out_ptr, cap):
        cons, prod = c_sz(), c_sz()
        ev = (C.c_int * 8)()
        nev, lt, cns = C.c_int(), C.c_ulong(), C.c_int()
        check(lib.lsdr_mpeg_sync_run(self.h, in_ptr, n_in, out_ptr, cap, C.byref(cons), C.byref(prod), ev, C.byref(nev),
                                     C.byref(lt), C.byref(cns)))
        return cons.value, prod.value, list(ev[:nev.value]), lt.value, cns.value

    def run_stream(self, data):
        data = np.ascontiguousarray(data, np.uint8)
        din = self.ctx.upload(data)
        dout = self.ctx.alloc(len(data) + 4096)
        pos, nout, events = 0, 0, []
        while True:
            cons, prod, ev, lt, cns = self.run_dev(din.at(pos), len(data) - pos, dout.at(nout), len(data) + 4096 - nout)
            events += ev
            if not cons and not prod:
                break
            pos += cons
            nout += prod
        out = self.ctx.download(dout, np.uint8, nout)
        din.free(); dout.free()
        return out, events


def deinterleaver(ctx, data):
    data = np.ascontiguousarray(data, np.uint8)
    din = ctx.upload(data)
    cap = len(data) // 204 + 1
    dout = ctx.alloc(cap * 204)
    cons, prod = c_sz(), c_sz()
    check(lib.lsdr_deinterleaver_run(ctx.h, din.ptr, len(data), dout.ptr, cap, C.byref(cons), C.byref(prod)))
    out = ctx.download(dout, np.uint8, prod.value * 204).reshape(-1, 204)
    din.free(); dout.free()
    return out, cons.value


def rs_decoder(ctx, packets):
    packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 204)
    din = ctx.upload(packets)
    dout = ctx.alloc(max(1, len(packets)) * 188)
    b, e = C.c_long(), C.c_long()
    check(lib.lsdr_rs_decoder_run(ctx.h, din.ptr, len(packets), dout.ptr, C.byref(b), C.byref(e)))
    out = ctx.download(dout, np.uint8, len(packets) * 188).reshape(-1, 188)
    din.free(); dout.free()
    return out, b.value, e.value


class Derandomizer:
    def __init__(self, ctx):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_derandomizer_create(ctx.h, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_derandomizer_destroy(self.h)
            self.h = None

    def run(self, packets):
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 188)
        din = self.ctx.upload(packets)
        dout = self.ctx.alloc(max(1, len(packets)) * 188)
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_derandomizer_run(self.h, din.ptr, len(packets), dout.ptr, len(packets), C.byref(cons), C.byref(prod)))
        out = self.ctx.download(dout, np.uint8, prod.value * 188).reshape(-1, 188)
        din.free(); dout.free()
        return out


# ---- auto_notch / cnr_fft / cfft ------------------------------------------------------
def cfft_dev(ctx, x, reverse=False):
    """cfft_engine<float> on the GPU (k_cfft): upload one block, transform, spectrum back on the host."""
    x = np.ascontiguousarray(x, np.complex64)
    din = ctx.upload(x)
    out = np.empty_like(x)
    check(lib.lsdr_cfft_run(ctx.h, len(x), int(reverse), din.ptr, _np(out)))
    din.free()
    return out


def cfft_host(x, reverse=False):
    x = np.ascontiguousarray(x, np.complex64).copy()
    check(lib.lsdr_cfft_host(len(x), _np(x), int(reverse)))
    return x


class AutoNotch:
    """auto_notch<f32> (sdr.h:46-154) on the GPU."""

    def __init__(self, ctx, nslots=1, setpoint=0.0, decimation=1024 * 4096, k=0.002, mode=0):
        """mode: NOTCH_EXACT (0, bit-exact verified tiles) or NOTCH_SCAN (1, single-pass scan, device-side detect; tolerance)."""
        self.ctx, self.nslots = ctx, nslots
        h = vp()
        check(lib.lsdr_auto_notch_create(ctx.h, nslots, setpoint, C.byref(h)))
        self.h = h
        check(lib.lsdr_auto_notch_set(h, decimation, k))
        if mode:
            check(lib.lsdr_auto_notch_set_mode(h, mode))

    def close(self):
        if self.h:
            lib.lsdr_auto_notch_destroy(self.h)
            self.h = None

    def bins(self):
        return [lib.lsdr_auto_notch_slot_bin(self.h, s) for s in range(self.nslots)]

    def scan_time(self, enable=True):
        """(mean ms, launches) of the k_notch_scan launches recorded since the previous call; then switches recording."""
        ms, n = c_f(), C.c_uint()
        check(lib.lsdr_auto_notch_scan_time(self.h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stats(self):
        t, b = C.c_uint(), C.c_uint()
        check(lib.lsdr_auto_notch_stats(self.h, C.byref(t), C.byref(b)))
        return dict(tiles=t.value, bad_seams=b.value)

    def run_dev(self, in_ptr, n, out_ptr, cap):
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_auto_notch_run(self.h, in_ptr, n, out_ptr, cap, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        din = self.ctx.upload(x)
        dout = self.ctx.alloc(max(8, x.nbytes))
        cons, prod = self.run_dev(din.ptr, len(x), dout.ptr, len(x))
        out = self.ctx.download(dout, np.complex64, prod)
        din.free(); dout.free()
        return out


class NotchFir:
    """auto_notch<f32>(1 slot) fused into fir_filter<cf32,float> (leandvb's default graph, leandvb.cc:296-301): lsdr_notch_fir_*.
    Tolerance mode; LsdrError (LSDR_E_UNSUPPORTED) for geometries without the fused kernel — use AutoNotch + FirFilter then."""

    def __init__(self, ctx, coeffs, decim, in_scale=0.0, nslots=1, decimation=1024 * 4096, k=0.002):
        self.ctx = ctx
        self.coeffs = np.ascontiguousarray(coeffs, np.float32)
        self.decim = decim
        cfg = NotchFirCfg(len(self.coeffs), self.coeffs.ctypes.data, decim, in_scale, nslots, decimation, k)
        h = vp()
        check(lib.lsdr_notch_fir_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_notch_fir_destroy(self.h)
            self.h = None

    def bin(self):
        return lib.lsdr_notch_fir_slot_bin(self.h)

    def set_freq(self, freq):
        """fir_filter::set_freq (dsp.h:271-280): shifted taps from the next run on."""
        check(lib.lsdr_notch_fir_set_freq(self.h, freq))

    def track(self, freq_tap, tap_multiplier, freq_tol):
        """fir_filter::run's tracking step (dsp.h:236-244); returns whether the taps were re-shifted."""
        did = C.c_int()
        check(lib.lsdr_notch_fir_track(self.h, freq_tap, tap_multiplier, freq_tol, C.byref(did)))
        return bool(did.value)

    @property
    def current_freq(self):
        return float(lib.lsdr_notch_fir_current_freq(self.h))

    def set_overlap(self, on=True):
        """Detect chain and filter pass of run k+1 on the block's own streams next to run k's tail (inputs must be complete at call time)."""
        check(lib.lsdr_notch_fir_set_overlap(self.h, int(on)))

    def pass_time(self, enable=True):
        """(mean ms, launches) of the filter pass (k_fir_mfma_stream, per-interval taps) since the previous call; then switches recording."""
        ms, n = c_f(), C.c_uint()
        check(lib.lsdr_notch_fir_time(self.h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def run_dev(self, in_ptr, n_in, out_ptr, cap_out):
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_notch_fir_run(self.h, in_ptr, n_in, out_ptr, cap_out, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run(self, x, step=None):
        """The whole stream `x` through the block in runs of at most `step` samples (None: one run); returns (outputs, consumed)."""
        x = np.ascontiguousarray(x, np.complex64)
        din = self.ctx.upload(x) if len(x) else None
        cap = len(x) // self.decim + 16
        dout = self.ctx.alloc(cap * 8)
        pos = nout = 0
        while din is not None:
            avail = len(x) - pos if step is None else min(len(x) - pos, step)
            cons, prod = self.run_dev(din.at(pos * 8), avail, dout.at(nout * 8), cap - nout)
            if not prod:
                if step is None or avail == len(x) - pos:
                    break
                step *= 2          # a run needs a whole 4096-block beyond the filter's history: offer more
                continue
            pos += cons; nout += prod
        out = self.ctx.download(dout, np.complex64, nout)
        if din is not None:
            din.free()
        dout.free()
        return out, pos


class FastQpsk:
    """fast_qpsk_receiver<u8> (sdr.h:946-1189), the --hs receiver."""

    def __init__(self, ctx, omega, freq=0.0, pll_adjustment=1.0, allow_drift=0, meas_decimation=0, tiled=False, tile_len=0,
                 tile_warmup=0):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_fastqpsk_create(ctx.h, omega, freq, pll_adjustment, allow_drift, meas_decimation, C.byref(h)))
        self.h = h
        if tiled:
            check(lib.lsdr_fastqpsk_set_tiled(h, 1, tile_len, tile_warmup))

    def set_tiled(self, enable, tile_len=0, tile_warmup=0):
        check(lib.lsdr_fastqpsk_set_tiled(self.h, int(enable), tile_len, tile_warmup))

    def tiled_stats(self):
        t, d, m, b = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
        check(lib.lsdr_fastqpsk_tiled_stats(self.h, C.byref(t), C.byref(d), C.byref(m), C.byref(b)))
        return dict(tiles=t.value, dup=d.value, miss=m.value, bad_seams=b.value)

    def close(self):
        if self.h:
            lib.lsdr_fastqpsk_destroy(self.h)
            self.h = None

    def state(self):
        mu, ph, fw, mn, mx = c_f(), C.c_uint(), C.c_longlong(), C.c_longlong(), C.c_longlong()
        check(lib.lsdr_fastqpsk_get_state(self.h, C.byref(mu), C.byref(ph), C.byref(fw), C.byref(mn), C.byref(mx)))
        return dict(mu=mu.value, phase=ph.value, freqw=fw.value, min_freqw=mn.value, max_freqw=mx.value)

    def run_dev(self, in_ptr, n_in, out_ptr, cap):
        """Device pointers (cu8 in, one hard symbol per byte out); no FREQ/constellation reports.  → (consumed, produced)."""
        cons, prod, nf, nc = c_sz(), c_sz(), c_sz(), c_sz()
        check(lib.lsdr_fastqpsk_run(self.h, in_ptr, n_in, out_ptr, cap, C.byref(cons), C.byref(prod), None, 0, C.byref(nf), None, 0,
                                    C.byref(nc)))
        return cons.value, prod.value

    def run(self, iq_u8, meas=True):
        """Upload interleaved u8 I/Q, run once, download.  Returns dict(sym, consumed, freq, cstln)."""
        iq = np.ascontiguousarray(iq_u8, np.uint8)
        n = len(iq) // 2
        din = self.ctx.upload(iq)
        dout = self.ctx.alloc(n + 256)
        fo = np.empty(n // 64 + 16, np.float32)
        co = np.empty((n // 64 + 16, 2), np.uint8)
        cons, prod, nf, nc = c_sz(), c_sz(), c_sz(), c_sz()
        check(lib.lsdr_fastqpsk_run(self.h, din.ptr, n, dout.ptr, n + 256, C.byref(cons), C.byref(prod),
                                    _np(fo) if meas else None, len(fo), C.byref(nf), _np(co) if meas else None, len(co), C.byref(nc)))
        sym = self.ctx.download(dout, np.uint8, prod.value)
        din.free(); dout.free()
        return dict(sym=sym, consumed=cons.value, freq=fo[:nf.value].copy(), cstln=co[:nc.value].copy())


class HsDeconv:
    """dvb_deconvol_sync<u8> (dvb.h:612-707), the --hs deconvolver."""

    def __init__(self, ctx, resync_period=32):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_hsdeconv_create(ctx.h, resync_period, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_hsdeconv_destroy(self.h)
            self.h = None

    @property
    def locked(self):
        return lib.lsdr_hsdeconv_locked(self.h)

    def run_dev(self, in_ptr, n_in, out_ptr, cap):
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_hsdeconv_run(self.h, in_ptr, n_in, out_ptr, cap, C.byref(cons), C.byref(prod)))
        return cons.value, prod.value

    def run_stream(self, symbols, pipe=None, room=None):
        sym = np.ascontiguousarray(symbols, np.uint8)
        din = self.ctx.upload(sym)
        dout = self.ctx.alloc(len(sym) // 8 + 64)
        pos = nout = 0
        while True:
            avail = len(sym) - pos if pipe is None else min(pipe, len(sym) - pos)
            cap = len(sym) // 8 + 64 - nout if room is None else min(room, len(sym) // 8 + 64 - nout)
            cons, prod = c_sz(), c_sz()
            check(lib.lsdr_hsdeconv_run(self.h, din.at(pos), avail, dout.at(nout), cap, C.byref(cons), C.byref(prod)))
            if not prod.value:
                break
            pos += cons.value
            nout += prod.value
        out = self.ctx.download(dout, np.uint8, nout)
        din.free(); dout.free()
        return out


class Rotator:
    """rotator<f32> (sdr.h:1226-1259)."""

    def __init__(self, ctx, freq):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_rotator_create(ctx.h, freq, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_rotator_destroy(self.h)
            self.h = None

    def run(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        din = self.ctx.upload(x)
        dout = self.ctx.alloc(max(8, len(x) * 8))
        check(lib.lsdr_rotator_run(self.h, din.ptr, len(x), dout.ptr))
        out = self.ctx.download(dout, np.complex64, len(x))
        din.free(); dout.free()
        return out


class Spectrum:
    """spectrum<f32> (sdr.h:1347-1404)."""

    def __init__(self, ctx, decimation=1048576, kavg=0.1):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_spectrum_create(ctx.h, C.byref(h)))
        self.h = h
        check(lib.lsdr_spectrum_set(h, decimation, kavg))

    def close(self):
        if self.h:
            lib.lsdr_spectrum_destroy(self.h)
            self.h = None

    def run(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        din = self.ctx.upload(x)
        out = np.empty((len(x) // 1024 + 1, 1024), np.float32)
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_spectrum_run(self.h, din.ptr, len(x), _np(out), len(out), C.byref(cons), C.byref(prod)))
        din.free()
        return out[:prod.value].copy(), cons.value

    def run_dev(self, in_ptr, n_in):
        out = np.empty((n_in // 1024 // 64 + 4, 1024), np.float32)
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_spectrum_run(self.h, in_ptr, n_in, _np(out), len(out), C.byref(cons), C.byref(prod)))
        return out[:prod.value].copy(), cons.value


class CnrFft:
    """cnr_fft<f32> (sdr.h:1273-1345)."""

    def __init__(self, ctx, bandwidth, nfft=4096, decimation=1048576, kavg=0.1):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_cnr_fft_create(ctx.h, bandwidth, nfft, C.byref(h)))
        self.h = h
        check(lib.lsdr_cnr_fft_set(h, decimation, kavg))

    def close(self):
        if self.h:
            lib.lsdr_cnr_fft_destroy(self.h)
            self.h = None

    def run(self, x, freq_tap=0.0, tap_multiplier=1.0):
        x = np.ascontiguousarray(x, np.complex64)
        din = self.ctx.upload(x)
        out = np.empty(len(x) // 64 + 8, np.float32)
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_cnr_fft_run(self.h, freq_tap, tap_multiplier, din.ptr, len(x), _np(out), len(out), C.byref(cons), C.byref(prod)))
        din.free()
        return out[:prod.value].copy(), cons.value

    def run_dev(self, in_ptr, n_in, freq_tap=0.0, tap_multiplier=1.0):
        """One run() over a device buffer: (CNR values appended by this call, samples consumed)."""
        out = np.empty(n_in // 64 + 8, np.float32)
        cons, prod = c_sz(), c_sz()
        check(lib.lsdr_cnr_fft_run(self.h, freq_tap, tap_multiplier, in_ptr, n_in, _np(out), len(out), C.byref(cons), C.byref(prod)))
        return out[:prod.value].copy(), cons.value


# ---- channel simulator (leanchansim.cc:34-190) -------------------------------------------
class Wgn:
    """wgn_c<f32> (dsp.h:164-190) on glibc's drand48/logf; seed=None: the state of a process that never seeds."""

    def __init__(self, ctx, seed=None):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_wgn_create(ctx.h, 0 if seed is None else 1, seed or 0, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.lsdr_wgn_destroy(self.h)
            self.h = None

    @property
    def state(self):
        x = C.c_ulonglong()
        check(lib.lsdr_wgn_get_state(self.h, C.byref(x)))
        return x.value

    def run(self, n, stddev=1.0, add=None):
        dout = self.ctx.alloc(max(8, n * 8))
        dadd = self.ctx.upload(np.ascontiguousarray(add, np.complex64)) if add is not None else None
        check(lib.lsdr_wgn_run(self.h, stddev, dadd.ptr if dadd else None, dout.ptr, n))
        out = self.ctx.download(dout, np.complex64, n)
        dout.free()
        if dadd:
            dadd.free()
        return out


def adder(ctx, a, b):
    a = np.ascontiguousarray(a, np.complex64); b = np.ascontiguousarray(b, np.complex64)
    n = min(len(a), len(b))
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(max(8, n * 8))
    check(lib.lsdr_adder_run(ctx.h, da.ptr, db.ptr, n, dout.ptr))
    out = ctx.download(dout, np.complex64, n)
    da.free(); db.free(); dout.free()
    return out


def cconv_f32_u8(ctx, x):
    x = np.ascontiguousarray(x, np.complex64)
    din, dout = ctx.upload(x), ctx.alloc(max(8, len(x) * 2))
    check(lib.lsdr_cconverter_f32_u8_run(ctx.h, din.ptr, len(x), dout.ptr))
    out = ctx.download(dout, np.uint8, len(x) * 2).reshape(-1, 2)
    din.free(); dout.free()
    return out


def cconv_f32_s16(ctx, x):
    x = np.ascontiguousarray(x, np.complex64)
    din, dout = ctx.upload(x), ctx.alloc(max(8, len(x) * 4))
    check(lib.lsdr_cconverter_f32_s16_run(ctx.h, din.ptr, len(x), dout.ptr))
    out = ctx.download(dout, np.int16, len(x) * 2).reshape(-1, 2)
    din.free(); dout.free()
    return out


class Drifter:
    """drifter<float> (leanchansim.cc:34-88)."""

    def __init__(self, ctx, amp=(0, 0, 0), freq=(0, 0, 0)):
        self.ctx = ctx
        h = vp()
        check(lib.lsdr_drifter_create(ctx.h, C.byref(h)))
        self.h = h
        for i in range(3):
            check(lib.lsdr_drifter_set_component(h, i, amp[i], freq[i]))

    def close(self):
        if self.h:
            lib.lsdr_drifter_destroy(self.h)
            self.h = None

    @property
    def phases(self):
        a = (C.c_longlong * 3)()
        check(lib.lsdr_drifter_get_phases(self.h, a))
        return tuple(a)

    @phases.setter
    def phases(self, v):
        check(lib.lsdr_drifter_set_phases(self.h, (C.c_longlong * 3)(*v)))

    def run(self, x, chunk=4096):
        x = np.ascontiguousarray(x, np.complex64)
        din, dout = self.ctx.upload(x), self.ctx.alloc(max(8, len(x) * 8))
        check(lib.lsdr_drifter_run(self.h, din.ptr, len(x), dout.ptr, chunk))
        out = self.ctx.download(dout, np.complex64, len(x))
        din.free(); dout.free()
        return out


# ---- transmit chain (leandvbtx.cc:79-175) ------------------------------------------------
class TxChain:
    """randomizer → rs_encoder → interleaver → dvb_convol → cstln_transmitter → fir_resampler(RRC) → decimator [→ simple_agc]
    on one context, device-resident between the blocks.  `run(ts)` feeds a batch of TS packets and returns the baseband
    produced so far (state is carried: interleaver window, convolutional history, resampler history, AGC estimate)."""

    def __init__(self, ctx, interp=2, decim=1, amp=1.0, rolloff=0.35, rrc_rej=10.0, agc=False, cstln=QPSK, rate=FEC12):
        self.ctx, self.interp, self.decim, self.cstln, self.rate = ctx, interp, decim, cstln, rate
        self.rand = vp(); check(lib.lsdr_randomizer_create(ctx.h, C.byref(self.rand)))
        bps = CSTLN_BITS[cstln]
        conv_rate = FEC46 if (rate == FEC23 and bps in (2, 6)) else rate      # leandvbtx.cc:117-121
        self.conv = vp(); check(lib.lsdr_convol_create(ctx.h, conv_rate, bps, C.byref(self.conv)))
        order = int(interp * rrc_rej)
        co = root_raised_cosine(order, float(np.float32(1.0) / np.float32(interp)), rolloff)
        co = normalize_power(co, float(np.float32(amp) / np.float32(75.0)))
        self.coeffs = co
        self.res = vp(); check(lib.lsdr_fir_resampler_create(ctx.h, len(co), _np(co), interp, C.byref(self.res)))
        self.agc = None
        if agc:
            self.agc = vp()
            check(lib.lsdr_simple_agc_create(ctx.h, float(np.float32(amp) / np.sqrt(np.float32(np.float32(interp) / decim))),
                                             float(np.float32(0.001 * decim / interp)), C.byref(self.agc)))
        self.pk_hold = np.zeros((0, 204), np.uint8)      # interleaver window (host-side carry of unconsumed packets)
        self.il_hold = np.zeros(0, np.uint8)              # interleaved bytes short of a convolutional group
        self.iq_hold = np.zeros(0, np.complex64)          # resampler input not yet consumed
        self.dec_hold = np.zeros(0, np.complex64)         # decimator / AGC remainders
        self.agc_hold = np.zeros(0, np.complex64)

    def close(self):
        lib.lsdr_randomizer_destroy(self.rand); lib.lsdr_convol_destroy(self.conv); lib.lsdr_fir_resampler_destroy(self.res)
        if self.agc:
            lib.lsdr_simple_agc_destroy(self.agc)

    def _run2(self, fn, h, din, n_in, cap_items, out_dtype, item=1):
        """One *_run call; `cap_items` / produced are in the block's output items of `item` elements of out_dtype."""
        dout = self.ctx.alloc(max(16, cap_items * item * np.dtype(out_dtype).itemsize))
        cons, prod = c_sz(), c_sz()
        if h is None:
            check(fn(self.ctx.h, din.ptr, n_in, dout.ptr, cap_items, C.byref(cons), C.byref(prod)))
        else:
            check(fn(h, din.ptr, n_in, dout.ptr, cap_items, C.byref(cons), C.byref(prod)))
        out = self.ctx.download(dout, out_dtype, prod.value * item)
        dout.free()
        return out, cons.value

    def run(self, ts):
        ctx = self.ctx
        ts = np.ascontiguousarray(ts, np.uint8).reshape(-1, 188)
        d = ctx.upload(ts)
        r, _ = self._run2(lib.lsdr_randomizer_run, self.rand, d, len(ts), len(ts), np.uint8, 188); d.free()
        r = r.reshape(-1, 188)
        d = ctx.upload(r)
        pk, _ = self._run2(lib.lsdr_rs_encoder_run, None, d, len(r), len(r), np.uint8, 204); d.free()
        pk = np.concatenate([self.pk_hold, pk.reshape(-1, 204)])
        if len(pk) < 12:
            self.pk_hold = pk
            return np.zeros(0, np.complex64)
        d = ctx.upload(pk)
        il, cons = self._run2(lib.lsdr_interleaver_run, None, d, len(pk), len(pk) * 204, np.uint8); d.free()
        self.pk_hold = pk[cons:]
        il = np.concatenate([self.il_hold, il])                 # dvb_convol consumes whole groups of bits_in bytes
        d = ctx.upload(il)
        sym, cons = self._run2(lib.lsdr_convol_run, self.conv, d, len(il), len(il) * 16 + 64, np.uint8); d.free()
        self.il_hold = il[cons:]
        d = ctx.upload(sym)
        dq = ctx.alloc(max(16, len(sym) * 8))
        check(lib.lsdr_cstln_transmitter_run(ctx.h, self.cstln, self.rate, d.ptr, len(sym), dq.ptr)); d.free()
        iq = np.concatenate([self.iq_hold, ctx.download(dq, np.complex64, len(sym))]); dq.free()
        d = ctx.upload(iq)
        y, cons = self._run2(lib.lsdr_fir_resampler_run, self.res, d, len(iq), len(iq) * self.interp, np.complex64); d.free()
        self.iq_hold = iq[cons:]
        y = np.concatenate([self.dec_hold, y])
        nd = len(y) // self.decim
        self.dec_hold = y[nd * self.decim:]
        y = np.ascontiguousarray(y[:nd * self.decim:self.decim]) if self.decim > 1 else y
        if self.agc:
            y = np.concatenate([self.agc_hold, y])
            d = ctx.upload(y)
            z, cons = self._run2(lib.lsdr_simple_agc_run, self.agc, d, len(y), len(y), np.complex64); d.free()
            self.agc_hold = y[cons:]
            y = z
        return y

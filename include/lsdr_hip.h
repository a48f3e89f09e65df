/* include/lsdr_hip.h — C ABI of the MI355X-native leandvb hot path.
 *
 * This is the drop-in boundary: the reference has no FFI layer,
 * a "block" is a C++ `runnable` whose run() moves items between pipebufs
 * (framework.h:124-131).  The host keeps that scheduler/pipebuf surface
 * (leansdr_amd/host headers, same class names and constructor signatures) and each
 * GPU-backed block's run() calls exactly one of the `*_run` entry points below.
 * Every entry point names the reference interface it replaces.
 *
 * Conventions
 *  - Plain C types only.  No torch / HIP types: a stream is passed as void*.
 *  - All `in`/`out` data pointers are DEVICE pointers (HBM) unless the name
 *    ends in `_host`.  lsdr_malloc/lsdr_memcpy_* exist so that a host without
 *    any HIP binding can own device pipebufs.
 *  - Every function returns 0 on success, <0 on error (LSDR_E_*); the message
 *    is available from lsdr_last_error().  The C++ shim maps !=0 to the
 *    reference's fail() (framework.h:33).  "Not enough input / no room" is not
 *    an error: *produced == 0, exactly like a reference run() that returns
 *    without progress (SURVEY §8b "Error convention").
 *  - One lsdr_ctx = one device + one HIP stream; all calls on a ctx (and on the
 *    blocks created on it) must come from one thread at a time (the scheduler
 *    thread, README.coding.md:29).  Different contexts may be driven from
 *    different threads concurrently (per-context tables, counters and staging;
 *    lsdr_last_error() is per thread): bench_more.py runs the front end and the
 *    FEC tail of one capture that way.
 *  - Work is enqueued on the ctx stream.  Functions whose outputs have a
 *    data-dependent size (cstln_receiver, the FEC tail) synchronise before
 *    returning so that the scheduler's progress hash (framework.h:96-113)
 *    never sees "no progress" while work is in flight.
 *  - Item layouts are the reference's: cf32 {float re,im} 8 B,
 *    cu8 {u8 re,im} 2 B, softsymbol {int16 cost; u8 symbol; u8 pad} 4 B,
 *    rspacket 204 B, tspacket 188 B.
 */
#ifndef LSDR_HIP_H
#define LSDR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSDR_ABI_VERSION 2

enum { LSDR_OK = 0, LSDR_E_HIP = -1, LSDR_E_ARG = -2, LSDR_E_NOMEM = -3, LSDR_E_UNSUPPORTED = -4 };

typedef struct { float re, im; } lsdr_cf32;
typedef struct { uint8_t re, im; } lsdr_cu8;
typedef struct { int16_t cost; uint8_t symbol; uint8_t pad; } lsdr_softsymbol; /* sdr.h:287-290 */

/* ------------------------------------------------------------------ context */
typedef struct lsdr_ctx lsdr_ctx;
int lsdr_abi_version(void);
/* PCI address of a device ("0000:c1:00.0", at least 16 bytes): lets the host pin a capture's threads to the GPU's NUMA node */
int lsdr_device_pci_bus_id(int device, char *buf, int len);
const char *lsdr_last_error(void);
/* stream == NULL: the ctx creates (and owns) a non-blocking HIP stream.
 *
 * PROCESS-WIDE SIDE EFFECT: the first context created in a process changes the C library's allocator policy for the WHOLE
 * process — mallopt(M_MMAP_THRESHOLD, 32 MiB) and mallopt(M_TRIM_THRESHOLD, 2 GiB), once, thread-safe (glibc only).  Reason: when
 * a host block of hundreds of KB is freed, glibc unmaps it; the amdgpu MMU notifier then suspends and restores every GPU queue of
 * the process, and the next submission waits 20-25 ms (measured: viterbi_sync 25 ms per call instead of 4.5).  Consequence for the
 * host application: freed host memory up to 32 MiB per block stays in the process heap instead of going back to the OS.  Set
 * LSDR_KEEP_MALLOC=1 in the environment before the first context to leave the allocator alone (and accept the stalls, or keep
 * your own large host buffers alive while GPU work is queued). */
int lsdr_ctx_create(int device, void *hip_stream, lsdr_ctx **ctx);
/* A context whose stream may only use the compute units set in cu_mask (bit i of word i/32 = CU i): lets two blocks
 * that would disturb each other (the HBM-streaming fir_filter and the latency-bound receiver tiles) own disjoint parts of
 * the chip.  Kernels size their grids for the masked CU count. */
int lsdr_ctx_create_masked(int device, const uint32_t *cu_mask, unsigned mask_words, lsdr_ctx **ctx);
void lsdr_ctx_destroy(lsdr_ctx *ctx);
int lsdr_ctx_sync(lsdr_ctx *ctx);
void *lsdr_ctx_stream(lsdr_ctx *ctx);
int lsdr_device_count(void);
/* Device pipebuf storage (replaces `new T[size]` of pipebuf, framework.h:141-143). */
int lsdr_malloc(lsdr_ctx *ctx, size_t bytes, void **dev_ptr);
int lsdr_free(lsdr_ctx *ctx, void *dev_ptr);
int lsdr_malloc_host(size_t bytes, void **pinned_ptr);   /* pinned staging for file_reader/writer */
int lsdr_free_host(void *pinned_ptr);
/* ---- placed stream buffers (arena.hip) ----
 * WHERE a resident buffer lands in HBM decides how fast a streaming kernel reads it (the same fir_filter launch: 0.37 ms over one 2 GiB
 * buffer, 0.42 ms over the next, reproducibly per buffer; it goes with the physical backing, and the windows of ONE large allocation are of
 * the fast kind far more often than separate allocations).  An arena is one device allocation handed out in 2 MiB-aligned windows, the
 * FASTEST free ones first: replaces, for the large stream buffers of a graph, the `new T[size]` of pipebuf (framework.h:141-143) that has no
 * such concern on a CPU.
 *   lsdr_arena_place   the n_best fastest of up to max_windows free candidate windows of `bytes` (from the arena's start, or from its end
 *                      downwards with from_tail).  "Fastest" under `probe` — it queues, on the context's stream, the launch whose speed
 *                      matters over the candidate it is given; the library times 6 calls after 3 untimed ones with events — or, with a null
 *                      probe, under a built-in streaming read of the window.  fill_from (device pointer, may be null): every candidate is
 *                      first filled with `bytes` bytes from there (a probe that reads samples needs samples).  The search ends early once
 *                      the n-th best candidate is 8 % under the median of at least five.  out[n_best], ms[n_best] (may be null): the
 *                      windows and their probe times, fastest first.  The arena REMEMBERS every probed window (time relative to its call's
 *                      median): later calls try free positions inside stretches known to be fast first, never-probed ones next, known-slow
 *                      ones last (the slow kind comes in stretches of GiBs; small buffers fit into a fast large window that was not taken).
 *   lsdr_arena_probe_log  the probe time of every candidate the last lsdr_arena_place tried, in the order tried.
 *   lsdr_ctx_set_arena    from now on lsdr_malloc(ctx, ≥ 1 MiB) is served from the arena (built-in probe, ≤ 12 candidates; an ordinary
 *                      allocation once the arena is full) and lsdr_free gives such windows back: a graph built on the host framework
 *                      gets placed device pipes unchanged.  Null detaches.  The arena must outlive what was allocated from it.
 * One thread per arena, like every object of a context. */
typedef struct lsdr_arena lsdr_arena;
typedef int (*lsdr_probe_fn)(void *user, void *candidate_window);
int lsdr_arena_create(lsdr_ctx *ctx, size_t bytes, lsdr_arena **arena);   /* LSDR_E_NOMEM: no such piece of device memory */
void lsdr_arena_destroy(lsdr_arena *arena);
size_t lsdr_arena_bytes(const lsdr_arena *arena);
int lsdr_arena_owns(const lsdr_arena *arena, const void *dev_ptr);
int lsdr_arena_place(lsdr_arena *arena, size_t bytes, unsigned n_best, unsigned max_windows, int from_tail, const void *fill_from,
                     lsdr_probe_fn probe, void *user, void **out, float *ms);
int lsdr_arena_time(lsdr_arena *arena, void *dev_ptr, lsdr_probe_fn probe, void *user, float *ms);   /* the probe over any buffer, timed like a candidate */
int lsdr_arena_release(lsdr_arena *arena, void *window);
int lsdr_arena_probe_log(const lsdr_arena *arena, float *ms, unsigned cap, unsigned *n);
int lsdr_ctx_set_arena(lsdr_ctx *ctx, lsdr_arena *arena);
int lsdr_memcpy_h2d(lsdr_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes); /* async on ctx stream */
int lsdr_memcpy_d2h(lsdr_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* async on ctx stream */
int lsdr_memcpy_d2d(lsdr_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);  /* pipebuf::pack() memmove, framework.h:153-159 (overlap-safe) */
int lsdr_memset(lsdr_ctx *ctx, void *dst_dev, int value, size_t bytes);
/* Copy engine of a context — what a pipebuf with ends on both sides of PCIe uses (host framework.h; replaces the plain
 * `new T[size]` + in-place access of framework.h:133-183 for the file_reader → first GPU block and last GPU block →
 * file_writer edges of leandvb.cc:205-256,593).  Two side streams, one per direction; `src_host`/`dst_host` should be
 * pinned (lsdr_malloc_host).
 *   lsdr_copy_h2d_async  upload on the upload stream; returns at once.
 *   lsdr_copy_fence      later work on the compute stream waits (on the GPU) for every upload enqueued so far.
 *   lsdr_copy_d2h_async  download on the download stream, ordered after everything already on the compute stream.
 *   lsdr_copy_sync_d2h   the host waits for the downloads.
 *   lsdr_copy_sync_all   the host waits for uploads, compute stream and downloads (pipe compaction). */
int lsdr_copy_h2d_async(lsdr_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int lsdr_copy_d2h_async(lsdr_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int lsdr_copy_fence(lsdr_ctx *ctx);
int lsdr_copy_sync_d2h(lsdr_ctx *ctx);
int lsdr_copy_sync_all(lsdr_ctx *ctx);
/* Stream timing for bench.py (HIP events recorded on the ctx stream). */
int lsdr_timer_start(lsdr_ctx *ctx);
int lsdr_timer_stop_ms(lsdr_ctx *ctx, float *ms); /* synchronises on the stop event */
/* Free-standing HIP events on the ctx stream (per-kernel timing inside a timed region). */
typedef struct lsdr_event lsdr_event;
int lsdr_event_create(lsdr_ctx *ctx, lsdr_event **ev);
void lsdr_event_destroy(lsdr_event *ev);
int lsdr_event_record(lsdr_event *ev);
int lsdr_event_elapsed_ms(lsdr_event *start, lsdr_event *stop, float *ms); /* synchronises on `stop` */
/* Make all later work on `ctx` wait for `ev` (recorded on another ctx's stream): the edge between
 * two blocks that run on different HIP streams (e.g. fir_filter of batch k+1 ‖ cstln_receiver of batch k). */
int lsdr_ctx_wait_event(lsdr_ctx *ctx, lsdr_event *ev);

/* ----------------------------------------------------- host-side table design
 * Replaces filtergen.h (coefficients are *inputs* to the kernels and must be
 * numerically identical to the reference's), math.h trig16 and the
 * cstln_lut<256> constructor.  Pure host code using the same libm. */
int lsdr_filtergen_lowpass(int order, float Fcut, float gain, float *coeffs_host);               /* filtergen.h:45-62; returns ncoeffs */
int lsdr_filtergen_root_raised_cosine(int order, float Fs, float rolloff, float *coeffs_host);   /* filtergen.h:68-92; returns ncoeffs */
void lsdr_filtergen_normalize_dcgain(int n, float *coeffs_host, float gain);                      /* filtergen.h:34-40 */
void lsdr_filtergen_normalize_power(int n, float *coeffs_host, float gain);                       /* filtergen.h:26-32 */
void lsdr_trig16_table(lsdr_cf32 *lut_host /*[65536]*/);                                         /* math.h:95-106 */
enum { LSDR_BPSK, LSDR_QPSK, LSDR_PSK8, LSDR_APSK16, LSDR_APSK32, LSDR_APSK64E,
       LSDR_QAM16, LSDR_QAM64, LSDR_QAM256 };                                                    /* sdr.h:318-324 */
enum { LSDR_FEC12, LSDR_FEC23, LSDR_FEC46, LSDR_FEC34, LSDR_FEC56, LSDR_FEC78,
       LSDR_FEC45, LSDR_FEC89, LSDR_FEC910 };                                                    /* dvb.h:37-41 */
/* make_dvbs2_constellation (dvb.h:45-81) + cstln_lut<256> ctor (sdr.h:326-560).
 * Arrays are indexed [(u8)I*256 + (u8)Q]; symbols is [nsymbols][2] (re,im). */
int lsdr_cstln_lut_build(int predef, int fec, int16_t *cost_host, uint8_t *symbol_host,
                         int16_t *phase_error_host, int8_t *symbols_host, int *nrotations); /* returns nsymbols */

/* ------------------------------------------------------ elementwise blocks */
/* cconverter<u8,128,f32,0,1,1>::run, dsp.h:40-50 */
int lsdr_cconverter_u8_run(lsdr_ctx *ctx, const lsdr_cu8 *in, size_t n, lsdr_cf32 *out);
/* cconverter<s8,0,f32,0,1,1>, <u16,32768,f32,0,1,1>, <s16,0,f32,0,1,1>::run (dsp.h:40-50; leandvb --s8 / --u16 / --s16,
 * leandvb.cc:218-248): `in` is n complex items of the given integer format. */
enum { LSDR_IN_CS8 = 2, LSDR_IN_CU16 = 3, LSDR_IN_CS16 = 4 };   /* (LSDR_IN_CF32 = 0, LSDR_IN_CU8 = 1: fir_filter's fused formats) */
int lsdr_cconverter_int_run(lsdr_ctx *ctx, int in_format, const void *in, size_t n, lsdr_cf32 *out);
/* scaler<float,cf32,cf32>::run, dsp.h:149-156 */
int lsdr_scaler_run(lsdr_ctx *ctx, float scale, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out);
/* decimator<cf32>::run, generic.h:256-262; *produced = min(n/d, cap), consumes produced*d */
int lsdr_decimator_run(lsdr_ctx *ctx, unsigned d, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out,
                       size_t cap, size_t *produced);
/* rotator<f32> (sdr.h:1226-1259): out[i] = in[i]·(cos, sin)(2π·i·ifreq/65536), ifreq = (int)(freq·65536); the 16-bit table
 * index is carried across calls.  The table is built on the host with libm like the reference's constructor. */
typedef struct lsdr_rotator lsdr_rotator;
int lsdr_rotator_create(lsdr_ctx *ctx, float freq, lsdr_rotator **r);
void lsdr_rotator_destroy(lsdr_rotator *r);
int lsdr_rotator_run(lsdr_rotator *r, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out);

/* -------------------------------------------------------------- auto_notch / cnr_fft / cfft
 * auto_notch<f32>, sdr.h:46-154: every `decimation` samples the `nslots` strongest bins of a 4096-point
 * spectrum are re-detected; every sample is cleaned by one first-order complex tracker per slot.
 * The per-sample recurrence runs on the GPU (time tiles with warm-up; every seam is verified bit for
 * bit against the previous tile's end state, unverified spans are redone sequentially → bit-exact);
 * detect() is a once-per-millions-of-samples control step: the block is fetched to the host and the
 * reference's FFT / hypotf / cosf / sinf sequence is evaluated there (exact libm). */
typedef struct lsdr_auto_notch lsdr_auto_notch;
int lsdr_auto_notch_create(lsdr_ctx *ctx, int nslots, float agc_rms_setpoint, lsdr_auto_notch **a);
void lsdr_auto_notch_destroy(lsdr_auto_notch *a);
int lsdr_auto_notch_set(lsdr_auto_notch *a, int decimation, float k);      /* public members, sdr.h:49-50 */
int lsdr_auto_notch_slot_bin(const lsdr_auto_notch *a, int slot);
/* LSDR_NOTCH_EXACT (default): the reference's sequential arithmetic in verified time tiles — bit-exact.
 * LSDR_NOTCH_SCAN: throughput mode — the estimator recurrence as a single-pass scan (one wavefront per 1024 samples,
 * decoupled look-back for the carry), detect() entirely on the device, no host synchronisation per run; tolerance-tested
 * (tests/test_gpu_notch.py), needs 1-4 slots and agc_rms_setpoint == 0 (leandvb's configuration, leandvb.cc:300-301). */
enum { LSDR_NOTCH_EXACT = 0, LSDR_NOTCH_SCAN = 1 };
int lsdr_auto_notch_set_mode(lsdr_auto_notch *a, int mode);
int lsdr_auto_notch_stats(const lsdr_auto_notch *a, unsigned *tiles, unsigned *bad_seams);
/* Measurement hook (bench_more.py's roofline of the scan kernel): HIP events on the block's stream around the k_notch_scan
 * launch of every LSDR_NOTCH_SCAN run while enabled.  Each call returns the mean over the (up to 16 most recent) launches
 * recorded since the previous call — after waiting for the stream — and then sets the switch. */
/* LSDR_NOTCH_SCAN, opt-in: a run's detect chain (FFTs of the detect points of its INPUT -> peaks -> phasor tables) on a side stream,
 * overlapping the previous run's scan kernel.  The caller promises that an input buffer is complete when lsdr_auto_notch_run is
 * called with it (the side stream does not wait for earlier work queued on the context).  Same results. */
int lsdr_auto_notch_set_overlap(lsdr_auto_notch *n, int on);
/* LSDR_NOTCH_SCAN's cross-workgroup look-back spins are bounded: *aborted_run = 0 while all were served, else the number of the
 * first run in which one gave up (the next lsdr_auto_notch_run then fails instead of continuing from garbage) */
int lsdr_auto_notch_check(lsdr_auto_notch *n, unsigned *aborted_run);
#ifdef LSDR_MEASURE
/* Measurement build only (make -C leansdr_amd/csrc measure; NOT in liblsdr_hip.so): garbage into the scan mode's hand-off buffers
 * (totals and flags); a correct hand-off never reads it (tools/notch_poison_stress.py) */
int lsdr_auto_notch_debug_poison(lsdr_auto_notch *n);
#endif
int lsdr_auto_notch_scan_time(lsdr_auto_notch *a, int enable, float *avg_ms, unsigned *launches);
/* run(), sdr.h:64-75: whole 4096-sample blocks; *consumed == *produced.  Synchronous. */
int lsdr_auto_notch_run(lsdr_auto_notch *a, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out,
                        size_t *consumed, size_t *produced);

/* ---- notch_fir: auto_notch (one slot) FUSED into fir_filter — leandvb's default graph (`--anf 1`, leandvb.cc:103,296-301:
 * auto_notch<f32>(…, 1, 0) feeding fir_filter<cf32,float>) without the notched stream ever existing.  The notch of sdr.h:119-138 is, per
 * detect interval, the LTI filter (1 − k − p·z⁻¹)/(1 − p·z⁻¹), p = (1−k)·exp(j2π·bin/4096); composed with the decimating FIR it is ONE
 * complex-tap decimating FIR over the raw samples (matrix pipe, 8 B per sample) plus a first-order recurrence at the decimated rate;
 * detect() (FFT of the detect block, first maximum, sdr.h:76-118) runs on the device; outputs around a bin CHANGE are computed directly.
 * TOLERANCE MODE (float32 with exact phases, not the reference's rounding sequence): same bins as the reference; every output within
 * 2e-5 of the stream's full scale of fir_filter(auto_notch(x)) in the reference's arithmetic for bins below 2048 and within 1e-3 for
 * bins 2048…4095 (measured up to 3.8e-4), and within 1e-5 of the float64 restatement of the same filter for EVERY bin — the bounds
 * tests/test_gpu_notch_fir.py asserts.  (The larger figure is the reference's own doing: its phasor table is cosf/sinf of an angle
 * ROUNDED TO FLOAT, 2π·bin·i/4096 up to 25 736 rad for the negative-frequency bins, i.e. ±1e-3 rad of table noise that an exact-phase
 * formulation does not reproduce.)
 * Exists for nslots == 1, decim == 30, ncoeffs ≤ 330, cf32 input, agc set point 0; anything else: LSDR_E_UNSUPPORTED at create —
 * use the two blocks.  `in` is the RAW stream at fir_filter's read position (the block keeps the 32 samples in front of it).
 * One run = auto_notch::run over the whole 4096-sample blocks present, then fir_filter::run over what the notch released:
 * *produced = (A − F − ncoeffs)/decim with A the notch's frontier (a multiple of 4096 of the stream), *consumed = *produced·decim.
 * Queued on the context's stream.  cap_out must hold at least 150 outputs for the block to make progress. */
typedef struct lsdr_notch_fir lsdr_notch_fir;
typedef struct {
  unsigned ncoeffs; const float *coeffs_host;   /* fir_filter's prototype taps (dsp.h:221-231) */
  unsigned decim;
  float in_scale;                               /* a fused scaler in front (0 or 1: none); rides on the taps */
  int nslots;                                   /* auto_notch's slots: 1 */
  int notch_decimation;                         /* auto_notch::decimation (0: 1024·4096, sdr.h:56) */
  float k;                                      /* auto_notch::k (0: 0.002) */
} lsdr_notch_fir_cfg;
int lsdr_notch_fir_create(lsdr_ctx *ctx, const lsdr_notch_fir_cfg *cfg, lsdr_notch_fir **h);
void lsdr_notch_fir_destroy(lsdr_notch_fir *h);
int lsdr_notch_fir_set(lsdr_notch_fir *h, int decimation, float k);
/* fir_filter's carrier tracking (dsp.h:236-244,271-280) on the fused block: the taps are re-shifted (host libm, as lsdr_fir_filter_set_freq)
 * and take effect with the next run; the decimated-rate recurrence is re-anchored under the new taps. */
int lsdr_notch_fir_set_freq(lsdr_notch_fir *h, float freq);
int lsdr_notch_fir_track(lsdr_notch_fir *h, float freq_tap, float tap_multiplier, float freq_tol, int *shifted);
float lsdr_notch_fir_current_freq(const lsdr_notch_fir *h);
int lsdr_notch_fir_run(lsdr_notch_fir *h, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out, size_t *consumed, size_t *produced);
int lsdr_notch_fir_slot_bin(lsdr_notch_fir *h);   /* waits for the stream; −1: nothing detected yet */
/* Opt-in: run k+1's detect chain and filter pass on ONE stream of the block's own, next to run k's tail (first output, fix-ups, recurrence,
 * state) on the context's stream; the output is complete in the order of the context's stream, as always.  The own stream does NOT wait
 * for earlier work queued on the context: the caller promises that an input buffer is complete when lsdr_notch_fir_run is called with
 * it and stays untouched until that run has completed on the context's stream.  Same results, bit for bit. */
int lsdr_notch_fir_set_overlap(lsdr_notch_fir *h, int on);
/* measurement hook: HIP events around the filter pass of every run while enabled (as lsdr_auto_notch_scan_time) */
int lsdr_notch_fir_time(lsdr_notch_fir *h, int enable, float *avg_ms, unsigned *launches);
/* cfft_engine<float>::inplace (dsp.h:56-116).  lsdr_cfft_run: the transform of one device block on the GPU (one workgroup,
 * in LDS, bit-identical butterflies) with the spectrum delivered to the host — what auto_notch::detect, cnr_fft and spectrum
 * use.  lsdr_cfft_host: the same arithmetic on host data (utility / cross-check). */
int lsdr_cfft_run(lsdr_ctx *ctx, int n, int reverse, const lsdr_cf32 *in_dev, lsdr_cf32 *out_host);
int lsdr_cfft_host(int n, lsdr_cf32 *data, int reverse);
/* cnr_fft<f32>, sdr.h:1273-1345: consumes whole nfft blocks; every `decimation` samples one block is
 * fetched and a CNR value (dB) is appended to cnr_out_host.  Synchronous only when a block is fetched. */
typedef struct lsdr_cnr_fft lsdr_cnr_fft;
int lsdr_cnr_fft_create(lsdr_ctx *ctx, float bandwidth, int nfft, lsdr_cnr_fft **c);
void lsdr_cnr_fft_destroy(lsdr_cnr_fft *c);
int lsdr_cnr_fft_set(lsdr_cnr_fft *c, int decimation, float kavg);         /* public members, sdr.h:1287-1288 */
int lsdr_cnr_fft_run(lsdr_cnr_fft *c, float freq_tap, float tap_multiplier, const lsdr_cf32 *in, size_t n_in,
                     float *cnr_out_host, size_t cap_out, size_t *consumed, size_t *produced);

/* spectrum<f32>, sdr.h:1347-1404 (nfft = 1024): consumes whole 1024-sample blocks; every `decimation` samples one
 * block is fetched, FFT'd on the host (cfft_engine), averaged (kavg) and written as one row of 1024 dB values,
 * fftshifted, to spectrum_out_host[cap_rows][1024].  Replaces spectrum::run / do_spectrum (sdr.h:1361-1396). */
typedef struct lsdr_spectrum lsdr_spectrum;
int lsdr_spectrum_create(lsdr_ctx *ctx, lsdr_spectrum **s);
void lsdr_spectrum_destroy(lsdr_spectrum *s);
int lsdr_spectrum_set(lsdr_spectrum *s, int decimation, float kavg);      /* public members, sdr.h:1358-1359 */
int lsdr_spectrum_run(lsdr_spectrum *s, const lsdr_cf32 *in, size_t n_in, float *spectrum_out_host, size_t cap_rows,
                      size_t *consumed, size_t *produced_rows);

/* -------------------------------------------------------------- fir_filter
 * fir_filter<cf32,float>, dsp.h:219-285 (decimating FIR, real prototype taps
 * frequency-shifted to complex).  The optional input stage fuses the block in
 * front of it in the leandvb graph so that its output never round-trips
 * through HBM: cconverter (u8 input, leandvb.cc:215) or scaler (f32 input,
 * leandvb.cc:255-256); results are bit-identical to running the two blocks
 * separately. */
enum { LSDR_IN_CF32 = 0, LSDR_IN_CU8 = 1 };
enum {
  LSDR_FIR_EXACT = 0, /* reference arithmetic: i-ascending accumulation, no FMA contraction → bit-exact */
  LSDR_FIR_FMA = 1,   /* same order, fused multiply-add (≤ 1 ulp/tap differences; tolerance-tested) */
  LSDR_FIR_MFMA = 2,  /* LSDR_FIR_FMA's arithmetic, bit for bit, on the f32 matrix pipe (v_mfma_f32_16x16x4_f32 is an exact
                       * k-ordered fmaf chain): the taps as a banded Toeplitz block, sixteen outputs per row group.  cf32 input
                       * at the decimations with a compile-time kernel; anything else runs LSDR_FIR_FMA's kernels (same bits). */
  LSDR_FIR_MFMA_BLK = 3 /* block-polyphase form on the matrix pipe (a dense product): the taps in blocks of `decim`, each block an
                       * fmaf chain from zero in tap order (complex taps, set_freq != 0: four taps at a time — their re-part products, then
                       * their im-part products), the block sums added in block order, in_scale multiplied into the taps (one
                       * rounding per tap) instead of the samples — its own stated arithmetic (oracle lo_fir_filter_blk), same
                       * error bound as LSDR_FIR_FMA.  cf32 input, every decimation 1 … 64 (whatever
                       * Fs / (4·Fm) leandvb.cc:353-378 computes), ncoeffs ≤ 16·decim; refused (LSDR_E_ARG) otherwise. */
};
typedef struct {
  unsigned ncoeffs;          /* fir_filter ctor _ncoeffs */
  const float *coeffs_host;  /* fir_filter ctor _coeffs (host) */
  unsigned decim;            /* fir_filter ctor _decim */
  int in_format;             /* LSDR_IN_* */
  float in_scale;            /* 0 = no fused scaler; else samples are multiplied by it first */
  int arith;                 /* LSDR_FIR_* */
} lsdr_fir_filter_cfg;
typedef struct lsdr_fir_filter lsdr_fir_filter;
/* Precondition of the matrix-pipe arithmetics (LSDR_FIR_MFMA, LSDR_FIR_MFMA_BLK): FINITE input samples.  Their K padding and the zero band of
 * the Toeplitz form multiply padding taps of 0.0 with real samples (fma(0, x, acc) = acc needs a finite x): one Inf or NaN sample reaches up to
 * 15·decim more outputs than in LSDR_FIR_EXACT / LSDR_FIR_FMA, where it stays inside its own tap window.  The same holds for what lies up to
 * 4 GiB behind the first tile of a stream longer than 4 GiB (read through a clamped buffer resource instead of zeros: finite, hence harmless). */
int lsdr_fir_filter_create(lsdr_ctx *ctx, const lsdr_fir_filter_cfg *cfg, lsdr_fir_filter **f);
void lsdr_fir_filter_destroy(lsdr_fir_filter *f);
/* set_freq(f), dsp.h:271-280 (host libm cosf/sinf, then upload). */
int lsdr_fir_filter_set_freq(lsdr_fir_filter *f, float freq);
/* The freq_tap prologue of run(), dsp.h:236-244: new=tap*mult; if |current-new|>tol → set_freq(new).
 * *shifted (may be NULL) reports whether a re-shift happened. */
int lsdr_fir_filter_track(lsdr_fir_filter *f, float freq_tap, float tap_multiplier, float freq_tol, int *shifted);
float lsdr_fir_filter_current_freq(const lsdr_fir_filter *f);
int lsdr_fir_filter_get_shifted_coeffs(const lsdr_fir_filter *f, lsdr_cf32 *shifted_host);
/* run() body, dsp.h:233-262.  in: n_in items of in_format.  Needs n_in >= ncoeffs;
 * *produced = min((n_in-ncoeffs)/decim, cap_out); *consumed = *produced*decim.
 * Output m = Σ_i sc[i]·in[ncoeffs + m·decim − i].  Asynchronous on the ctx stream. */
int lsdr_fir_filter_run(lsdr_fir_filter *f, const void *in, size_t n_in, lsdr_cf32 *out, size_t cap_out,
                        size_t *consumed, size_t *produced);
/* The same filter over n_streams (≤ 8) equal-length, independent buffers in ONE launch (no reference counterpart: the
 * reference has one fir_filter object per stream; this is the batched form of n_streams identical run() calls — same
 * consumed/produced for each, one prologue and one tail of the persistent kernel instead of n_streams). */
int lsdr_fir_filter_run_multi(lsdr_fir_filter *f, unsigned n_streams, const void *const *ins, size_t n_in,
                              lsdr_cf32 *const *outs, size_t cap_out, size_t *consumed, size_t *produced);

/* ---------------------------------------------------------- cstln_receiver
 * cstln_receiver<f32> + sampler_interface<f32>, sdr.h:589-938. */
enum { LSDR_SAMP_NEAREST = 0, LSDR_SAMP_LINEAR = 1, LSDR_SAMP_FIR = 2 }; /* sdr.h:600-689 */
enum { LSDR_SYM_SOFT = 0, LSDR_SYM_HARD2 = 1 };
enum {
  LSDR_RX_SERIAL = 0, /* one sequential pass, the reference's exact arithmetic → bit-exact soft symbols */
  LSDR_RX_TILED = 1   /* time-tiled with warm-up overlap (throughput mode; tolerance-tested) */
};
typedef struct {
  int sampler;               /* LSDR_SAMP_* */
  int ncoeffs;               /* fir_sampler(_ncoeffs,_coeffs,_subsampling), sdr.h:637-643 */
  const float *coeffs_host;
  int subsampling;
  int cstln, fec;            /* demod.cstln = make_dvbs2_constellation(cstln, fec), leandvb.cc:476 */
  int harden;                /* cstln->harden(), leandvb.cc:477-481 */
  float omega;               /* set_omega(Fs/Fm), leandvb.cc:482 */
  float freq;                /* set_freq(Ftune/Fs), leandvb.cc:483-487 (0: untouched) */
  float pll_adjustment;      /* public member; /=6 with --viterbi, leandvb.cc:498-501 */
  int allow_drift;           /* set_allow_drift() */
  unsigned long meas_decimation; /* public member, leandvb.cc:502 */
  float kest;                /* public member, default 0.01 */
  int mode;                  /* LSDR_RX_* */
  unsigned tile_len;         /* LSDR_RX_TILED: samples per tile (multiple of 128); 0 = default (twice the warm-up) */
  unsigned tile_warmup;      /* LSDR_RX_TILED: warm-up samples before each tile (multiple of 128); 0 = default (≈ 64 symbols) */
  int in_format;             /* LSDR_IN_CF32 (0, the reference's block) or LSDR_IN_CU8: the cconverter<u8,128,f32,0,1,1> in front
                              * of the receiver in the `leandvb --u8` graph (leandvb.cc:211-217, dsp.h:40-50) fused into the
                              * receiver's loads — `in` then points to lsdr_cu8 items and the converted cf32 stream never exists
                              * in HBM; results are bit-identical to running the two blocks separately */
  int out_format;            /* LSDR_SYM_SOFT (0): lsdr_softsymbol items.  LSDR_SYM_HARD2: LSDR_RX_TILED on QPSK only — the decisions
                              * alone, packed 16 per uint32 word, MSB first (symbol k of a run in word k/16, bits 31-2(k%16)..30-2(k%16)):
                              * what deconvol_sync, the next block of the default leandvb chain, reads of a softsymbol (`symbol & 3`,
                              * dvb.h:369-417); `out` then points to uint32 words and counts stay in symbols.  Needs cu8 input and the
                              * nearest or linear sampler. */
} lsdr_rx_cfg;
typedef struct {             /* the receiver's loop state, sdr.h:923-935 */
  float mu, phase, freqw, agc_gain, est_insp, est_sp, est_ep, freq_tap;
  float min_freqw, max_freqw;
  unsigned long meas_count;
  float hist[12];            /* hist[k] = {p.re,p.im,c.re,c.im}, k = 0..2 */
} lsdr_rx_state;
typedef struct lsdr_rx lsdr_rx;
int lsdr_rx_create(lsdr_ctx *ctx, const lsdr_rx_cfg *cfg, lsdr_rx **r);
void lsdr_rx_destroy(lsdr_rx *r);
int lsdr_rx_readahead(const lsdr_rx *r);                 /* sampler->readahead() */
int lsdr_rx_get_state(lsdr_rx *r, lsdr_rx_state *st);    /* includes freq_tap (sdr.h:918-921) */
int lsdr_rx_set_state(lsdr_rx *r, const lsdr_rx_state *st);
/* Back to the loop state lsdr_rx_create left (= a freshly constructed cstln_receiver, sdr.h:709-736 + leandvb.cc:476-487): the next
 * run starts a new capture.  Stream-ordered, no host wait; no queued runs may be outstanding. */
int lsdr_rx_reset(lsdr_rx *r);
/* LSDR_RX_TILED diagnostics of the last run: tiles, seams where a duplicated / lost symbol was
 * repaired, seams whose timing or carrier-phase mismatch exceeded the lock criterion. */
int lsdr_rx_tiled_stats(const lsdr_rx *r, unsigned *tiles, unsigned *dup, unsigned *miss, unsigned *bad_seams);
/* run(), sdr.h:772-916, over one buffer: consumes whole chunks of 128 samples while
 * n_in-pos >= 128+readahead and cap_out-produced >= 128 (and the measurement
 * outputs have room).  freq/ss/mer (meas_cap floats each, HOST pointers, may be
 * NULL) receive one value per meas_decimation samples; cstln_out_host (may be
 * NULL) one cf32 per chunk that produced a symbol.  Synchronous. */
int lsdr_rx_run(lsdr_rx *r, const void *in /* n_in items of cfg.in_format */, size_t n_in, lsdr_softsymbol *out, size_t cap_out,
                size_t *consumed, size_t *produced,
                float *freq_out_host, float *ss_out_host, float *mer_out_host, size_t meas_cap, size_t *n_meas,
                lsdr_cf32 *cstln_out_host, size_t cstln_cap, size_t *n_cstln);
/* Queued variant of lsdr_rx_run for LSDR_RX_TILED (no measurement outputs): puts the run on the context's stream and
 * returns at once with `consumed` (a pure function of the sizes); lsdr_rx_wait() retires the oldest queued run and
 * yields its symbol count.  Up to 8 runs may be queued; the loop state is carried on the device from run to run, so
 * the host never sits between two runs.  lsdr_rx_run / _set_state refuse to mix with outstanding queued runs. */
int lsdr_rx_run_async(lsdr_rx *rx, const void *in, size_t n_in, lsdr_softsymbol *out, size_t cap_out, size_t *consumed);
int lsdr_rx_wait(lsdr_rx *rx, size_t *produced);
/* lsdr_rx_run_async for n_rx independent captures at once (one receiver per capture — the reference has one cstln_receiver
 * object per stream, sdr.h:697-938, leandvb.cc:163): equally long inputs ins[i] → outs[i]; `consumed` is an ARRAY of n_rx
 * entries, consumed[i] = the samples receiver i takes (they differ only when the receivers are configured differently).  Every
 * run is planned before anything is queued: on an argument error no receiver has been queued.  Receivers
 * on ONE context with the same configuration share their launches (four per batch instead of four per capture); any other
 * combination is queued receiver by receiver.  Same results as n_rx separate lsdr_rx_run_async calls, bit for bit; each
 * receiver is retired with its own lsdr_rx_wait. */
int lsdr_rx_run_multi_async(lsdr_rx *const *rxs, unsigned n_rx, const void *const *ins, size_t n_in, lsdr_softsymbol *const *outs,
                            size_t cap_out, size_t *consumed);
/* The queued run of a LSDR_SYM_HARD2 receiver, writing its packed symbols from symbol position out_sym_offset of the stream
 * that starts at out_words[0] (symbols before it — a caller's unconsumed remainder of the previous run — are preserved, so
 * consecutive runs form one packed stream without a bit-shifting copy).  cap_out counts symbols after the offset. */
int lsdr_rx_run_async_hs2(lsdr_rx *rx, const void *in, size_t n_in, uint32_t *out_words, size_t out_sym_offset, size_t cap_out,
                          size_t *consumed);
/* freq_tap (sdr.h:919-921, cycles per sample) as of the end of the most recently retired queued run: lets the caller keep
 * fir_filter tracking the carrier (lsdr_fir_filter_track, dsp.h:236-244) from runs that have already completed while later
 * ones are still queued — the feedback of leandvb.cc:506-510 with a latency of the queue depth instead of a host wait. */
float lsdr_rx_retired_freq_tap(const lsdr_rx *rx);
/* Measurement hook (bench.py's roofline of the tile kernel): HIP events on the receiver's stream around the k_rx_tiles launch of
 * every queued run while enabled; each call returns the mean over the runs retired since the previous call, then sets the switch. */
int lsdr_rx_tile_time(lsdr_rx *rx, int enable, float *avg_ms, unsigned *launches);
/* Loop-state snapshot between queued runs: lsdr_rx_snapshot_async() puts a copy of the device-side loop state (the fields
 * of cstln_receiver<f32>, sdr.h:923-935) into a pinned slot in stream order, i.e. the state the NEXT queued run starts
 * from; lsdr_rx_get_snapshot() waits for the stream and returns it.  Lets a caller (bench.py's verification) replay one
 * queued run on a checker from exactly the state the device used, without putting the host between two runs.  Four slots
 * (`_slot` forms; the plain forms use slot 0), so that a batch in the middle of a long queue and the last one can both be kept. */
/* Exact receiver, one GPU LANE per independent capture: n_streams captures with the same parameters (BASELINE config 4's
 * shape), each with its own loop state, every lane running cstln_receiver<f32>::run's exact arithmetic (sdr.h:772-916) →
 * bit-exact soft symbols and state per capture, 64 captures per wavefront.  `in_dev` / `out_dev` are HOST arrays of
 * n_streams DEVICE pointers; every capture gets n_in samples and cap_out symbol slots; all consume the same *consumed. */
typedef struct lsdr_rx_batch lsdr_rx_batch;
int lsdr_rx_batch_create(lsdr_ctx *ctx, const lsdr_rx_cfg *cfg, unsigned n_streams, lsdr_rx_batch **b);
void lsdr_rx_batch_destroy(lsdr_rx_batch *b);
int lsdr_rx_batch_run(lsdr_rx_batch *b, const void *const *in_dev, size_t n_in, lsdr_softsymbol *const *out_dev, size_t cap_out,
                      size_t *consumed, size_t *produced /*[n_streams]*/);
int lsdr_rx_batch_get_state(lsdr_rx_batch *b, unsigned stream, lsdr_rx_state *st);
/* LSDR_RX_TILED, QPSK: whether the tolerance tiles take their decisions by arithmetic instead of the constellation-table
 * gather (only when lsdr_rx_create verified the arithmetic against all 65536 table entries: symbol/cost/point identical,
 * |phase_error difference| ≤ 2 table units — reported in *max_phase_error_delta). */
int lsdr_rx_decision_mode(const lsdr_rx *rx, int *arithmetic, unsigned *max_phase_error_delta);
int lsdr_rx_snapshot_async(lsdr_rx *rx);
int lsdr_rx_get_snapshot(lsdr_rx *rx, lsdr_rx_state *st);
int lsdr_rx_snapshot_async_slot(lsdr_rx *rx, unsigned slot);
int lsdr_rx_get_snapshot_slot(lsdr_rx *rx, unsigned slot, lsdr_rx_state *st);

/* ================================================================== DVB-S FEC tail
 * Item layouts: bytes are u8; RS packets 204 B (rspacket<u8>), TS packets 188 B (tspacket). */

/* ---- deconvol_sync<u8,0> via make_deconvol_sync_simple, dvb.h:122-513 (algebraic deconvolution).
 * `rate` is LSDR_FEC12/23/46/34/56/78.  fastlock (dvb.h:428-454) is not implemented on the device:
 * create fails with LSDR_E_UNSUPPORTED when requested. */
typedef struct lsdr_deconv lsdr_deconv;
int lsdr_deconv_create(lsdr_ctx *ctx, int rate, int fastlock, lsdr_deconv **d);
void lsdr_deconv_destroy(lsdr_deconv *d);
int lsdr_deconv_next_sync(lsdr_deconv *d);                       /* deconvol_sync::next_sync, dvb.h:185-193 */
int lsdr_deconv_reset(lsdr_deconv *d);                           /* the state of a freshly constructed block (a new stream begins); stream-ordered */
/* One run() call (run_decoding, dvb.h:419-470): skips `skip` symbols, needs >= 64 symbols of margin,
 * produces n = min(maxrd, cap_out) bytes when n >= 32.  Asynchronous (sizes are data-independent). */
int lsdr_deconv_run(lsdr_deconv *d, const lsdr_softsymbol *in, size_t n_in, uint8_t *out, size_t cap_out,
                    size_t *consumed, size_t *produced);

/* The same block on the packed hard symbols of a LSDR_SYM_HARD2 receiver (deconvol_sync reads `symbol & 3` only, dvb.h:373,396):
 * symbols [sym_offset, sym_offset + n_in) of the packed stream that starts at in_words[0]; same arithmetic, same outputs as
 * lsdr_deconv_run on the unpacked symbols.  No fastlock. */
int lsdr_deconv_run_hs2(lsdr_deconv *d, const uint32_t *in_words, size_t sym_offset, size_t n_in, uint8_t *out, size_t cap_out,
                        size_t *consumed, size_t *produced);

/* ---- viterbi_sync, dvb.h:1173-1416 (+ viterbi_dec / trellis / bitpath, viterbi.h).  Soft-decision Viterbi
 * with the reference's partial-metric update and its alignment search (conjugation x rotation x shift).
 * One run() call: all 128-block chunks that fit.  Bit-exact: the stream is decoded in tiles that start
 * from zero metrics a few chunks early; every seam is verified against the previous tile's end state
 * (all 64 path metrics and path registers) and anything unverified is re-decoded sequentially.
 * Synchronous. */
typedef struct lsdr_viterbi lsdr_viterbi;
int lsdr_viterbi_create(lsdr_ctx *ctx, int cstln, int rate, lsdr_viterbi **v);
void lsdr_viterbi_destroy(lsdr_viterbi *v);
int lsdr_viterbi_set_resync_period(lsdr_viterbi *v, int period);   /* public member resync_period (dvb.h:1232) */
int lsdr_viterbi_current_sync(const lsdr_viterbi *v);
/* diagnostics of the last run: tiles decoded, seams that failed verification (re-decoded serially) */
int lsdr_viterbi_stats(const lsdr_viterbi *v, unsigned *tiles, unsigned *bad_seams);
/* since create: seams re-decoded by the repair round that runs on the device behind the main launch (no host round trip), and the
 * launch -> readback rounds the host still had to add (a repaired tile whose end state changed fails the NEXT seam; anything that
 * does not settle is re-decoded sequentially: the output is exact either way) */
int lsdr_viterbi_repair_stats(const lsdr_viterbi *v, unsigned long long *device_repaired, unsigned long long *host_rounds);
/* host only (no GPU, no context): 1 if the trellis of (constellation, code rate) — built as trellis::init_convolutional does,
 * viterbi.h:59-92 — has the structure the four-lanes-per-tile kernel relies on (long inputs of QPSK 1/2 and 8PSK 2/3 then
 * use it; same bytes as every other kernel), 0 if not, -1 if viterbi_sync does not support the combination (dvb.h:1234-1331). */
int lsdr_viterbi_q4_supported(int cstln, int rate);
int lsdr_viterbi_run(lsdr_viterbi *v, const lsdr_softsymbol *in, size_t n_in, uint8_t *out, size_t cap_out,
                     size_t *consumed, size_t *produced);

/* ---- mpeg_sync<u8,0>, dvb.h:712-891: bit alignment, polarity and 0x47/0xB8 sync search, lock tracking.
 * One run() call.  *call_next_sync = 1 when the reference would call deconv->next_sync() (dvb.h:771-779);
 * state_events (host, may be NULL) receives the values written to the lock-state pipe (0/1), at most 2.
 * *locktime receives the running count of packets since lock (the last value written to the locktime
 * pipe; one value per produced packet, consecutive).  Synchronous. */
typedef struct lsdr_mpeg_sync lsdr_mpeg_sync;
int lsdr_mpeg_sync_create(lsdr_ctx *ctx, int fastlock, lsdr_mpeg_sync **m);
void lsdr_mpeg_sync_destroy(lsdr_mpeg_sync *m);
int lsdr_mpeg_sync_run(lsdr_mpeg_sync *m, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out,
                       size_t *consumed, size_t *produced, int *state_events_host, int *n_state_events,
                       unsigned long *locktime, int *call_next_sync);
int lsdr_mpeg_sync_locked(const lsdr_mpeg_sync *m);
int lsdr_mpeg_sync_reset(lsdr_mpeg_sync *m);                     /* a freshly constructed block (options kept); stream-ordered */
int lsdr_mpeg_sync_set_resync_period(lsdr_mpeg_sync *m, int period);   /* public member resync_period (dvb.h:717), used by --hs */

/* ---- deinterleaver<u8>::run, dvb.h:932-944 (Forney I=12, M=17 as a gather).  Needs 2448 bytes per
 * packet window; *produced packets of 204 B, *consumed = 204 * produced.  Asynchronous. */
int lsdr_deinterleaver_run(lsdr_ctx *ctx, const uint8_t *in, size_t n_in, uint8_t *out_packets, size_t cap_packets,
                           size_t *consumed, size_t *produced);

/* ---- rs_decoder<u8,0>::run + rs_engine, dvb.h:998-1053, rs.h:84-272.  in: n packets of 204 B (corrected in
 * place like the reference), out: n packets of 188 B (sync byte ^0x55 when uncorrectable).  `bits`/`errs`
 * are the values of the bitcount/errcount pipes for this call.  Synchronous (counters). */
int lsdr_rs_decoder_run(lsdr_ctx *ctx, uint8_t *in_packets, size_t n_packets, uint8_t *out_packets,
                        long *bits, long *errs);

/* ---- derandomizer, dvb.h:1107-1163: PRBS removal, resync on the inverted sync byte, drops packets whose
 * restored sync byte is not 0x47.  Synchronous (output count is data dependent). */
typedef struct lsdr_derandomizer lsdr_derandomizer;
int lsdr_derandomizer_create(lsdr_ctx *ctx, lsdr_derandomizer **d);
void lsdr_derandomizer_destroy(lsdr_derandomizer *d);
int lsdr_derandomizer_reset(lsdr_derandomizer *d);               /* a new stream begins (PRBS position 0) */
int lsdr_derandomizer_run(lsdr_derandomizer *d, const uint8_t *in_packets, size_t n_packets, uint8_t *out_packets,
                          size_t cap_packets, size_t *consumed, size_t *produced);
/* host-side tables for tests: PRBS pattern (dvb.h:1116-1129), GF(256) exp/log and RS generator (rs.h:47-105) */
void lsdr_derandomizer_pattern(uint8_t *pattern1504_host);
void lsdr_rs_tables(uint8_t *exp512_host, uint8_t *log256_host, uint8_t *G17_host);

/* -------------------------------------------------------------- `--hs` path (leandvb.cc:727-969)
 * fast_qpsk_receiver<u8> (sdr.h:946-1189): cu8 samples → hard QPSK symbols {0,1,2,3}; optional FREQ measurements
 * (one per meas_decimation samples) and one sampled constellation point per chunk, both to HOST buffers.
 * _create = ctor + set_omega(omega) + set_freq(freq) + the public members; _run = run() over one buffer (needs
 * 129 samples and room for 128 symbols).  Bit-exact incl. the carried loop state. */
typedef struct lsdr_fastqpsk lsdr_fastqpsk;
int lsdr_fastqpsk_create(lsdr_ctx *ctx, float omega, float freq, float pll_adjustment, int allow_drift,
                         unsigned long meas_decimation, lsdr_fastqpsk **r);
void lsdr_fastqpsk_destroy(lsdr_fastqpsk *r);
int lsdr_fastqpsk_get_state(const lsdr_fastqpsk *r, float *mu, unsigned *phase, long long *freqw, long long *min_freqw,
                            long long *max_freqw);
/* throughput mode (time-tiled like LSDR_RX_TILED; not bit-exact at the symbol level, no FREQ/constellation reports);
 * tile_len / tile_warmup in samples (multiples of 128), 0 = defaults (≈ 400 symbols of warm-up, tiles twice that) */
int lsdr_fastqpsk_set_tiled(lsdr_fastqpsk *r, int enable, unsigned tile_len, unsigned tile_warmup);
int lsdr_fastqpsk_tiled_stats(const lsdr_fastqpsk *r, unsigned *tiles, unsigned *dup, unsigned *miss, unsigned *bad_seams);
int lsdr_fastqpsk_run(lsdr_fastqpsk *r, const lsdr_cu8 *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *consumed,
                      size_t *produced, float *freq_out_host, size_t freq_cap, size_t *n_freq, lsdr_cu8 *cstln_out_host,
                      size_t cstln_cap, size_t *n_cstln);
/* dvb_deconvol_sync<u8> (dvb.h:612-707) on deconvol_poly2 (convolutional.h:80-192): 512 hard symbols → 64 bytes per
 * chunk, the four alignments re-scored every resync_period chunks. */
typedef struct lsdr_hsdeconv lsdr_hsdeconv;
int lsdr_hsdeconv_create(lsdr_ctx *ctx, int resync_period, lsdr_hsdeconv **d);
void lsdr_hsdeconv_destroy(lsdr_hsdeconv *d);
int lsdr_hsdeconv_locked(const lsdr_hsdeconv *d);
int lsdr_hsdeconv_run(lsdr_hsdeconv *d, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *consumed,
                      size_t *produced);

/* -------------------------------------------------------------- transmit chain (leandvbtx.cc:79-175)
 * Device pointers throughout; every *_run is the body of the reference block's run() over one buffer. */
typedef struct lsdr_randomizer lsdr_randomizer;          /* randomizer, dvb.h:1063-1102 (8-packet pattern position carried) */
int lsdr_randomizer_create(lsdr_ctx *ctx, lsdr_randomizer **r);
void lsdr_randomizer_destroy(lsdr_randomizer *r);
int lsdr_randomizer_run(lsdr_randomizer *r, const uint8_t *in_packets, size_t n_packets, uint8_t *out_packets, size_t cap_packets,
                        size_t *consumed, size_t *produced);
/* rs_encoder, dvb.h:957-980 + rs_engine::encode, rs.h:141-167: 188-byte packets → 204-byte packets */
int lsdr_rs_encoder_run(lsdr_ctx *ctx, const uint8_t *in_packets, size_t n_packets, uint8_t *out_packets, size_t cap_packets,
                        size_t *consumed, size_t *produced);
/* interleaver, dvb.h:899-921: needs 12 packets, consumes n−11 */
int lsdr_interleaver_run(lsdr_ctx *ctx, const uint8_t *in_packets, size_t n_packets, uint8_t *out_bytes, size_t cap_bytes,
                         size_t *consumed_packets, size_t *produced_bytes);
/* fec_specs[rate] (dvb.h:553-565): bits entering / leaving the convolutional coder and its generator polynomials. */
int lsdr_fec_spec(int rate, int *bits_in, int *bits_out, uint16_t polys_host[8]);
typedef struct lsdr_convol lsdr_convol;                  /* dvb_convol, dvb.h:567-604 (16-bit history carried) */
int lsdr_convol_create(lsdr_ctx *ctx, int rate, int bits_per_symbol, lsdr_convol **v);
void lsdr_convol_destroy(lsdr_convol *v);
int lsdr_convol_run(lsdr_convol *v, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *consumed, size_t *produced);
/* cstln_transmitter<f32,0>, sdr.h:1196-1222, with make_dvbs2_constellation(cstln, rate) */
int lsdr_cstln_transmitter_run(lsdr_ctx *ctx, int cstln, int rate, const uint8_t *sym, size_t n, lsdr_cf32 *out);
typedef struct lsdr_fir_resampler lsdr_fir_resampler;    /* fir_resampler<cf32,float>, dsp.h:290-364 (interpolator, decim = 1) */
int lsdr_fir_resampler_create(lsdr_ctx *ctx, unsigned ncoeffs, const float *coeffs_host, unsigned interp, lsdr_fir_resampler **f);
void lsdr_fir_resampler_destroy(lsdr_fir_resampler *f);
int lsdr_fir_resampler_set_freq(lsdr_fir_resampler *f, float freq);
int lsdr_fir_resampler_run(lsdr_fir_resampler *f, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out, size_t *consumed,
                           size_t *produced);
typedef struct lsdr_simple_agc lsdr_simple_agc;          /* simple_agc<f32>, sdr.h:238-274 (power estimate carried) */
int lsdr_simple_agc_create(lsdr_ctx *ctx, float out_rms, float bw, lsdr_simple_agc **a);
void lsdr_simple_agc_destroy(lsdr_simple_agc *a);
int lsdr_simple_agc_set(lsdr_simple_agc *a, float out_rms, float bw);
int lsdr_simple_agc_run(lsdr_simple_agc *a, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out, size_t *consumed,
                        size_t *produced);

/* ---- channel simulator of leanchansim (leanchansim.cc:34-190): noise, adder, LO drift, f32 → u8 ------------------ */
typedef struct lsdr_wgn lsdr_wgn;                          /* wgn_c<f32>, dsp.h:164-190, on glibc's drand48()/logf() */
/* seeded = 0: the drand48 state of a process that never seeds (leanchansim --deterministic); else srand48(seed)
 * (leanchansim.cc:146-147 seeds with the pid). */
int lsdr_wgn_create(lsdr_ctx *ctx, int seeded, long seed, lsdr_wgn **w);
void lsdr_wgn_destroy(lsdr_wgn *w);
int lsdr_wgn_get_state(lsdr_wgn *w, unsigned long long *x48);
int lsdr_wgn_set_state(lsdr_wgn *w, unsigned long long x48);
/* the next n samples of wgn_c::run (stddev is its public member); add != NULL: out = add + noise (adder fused) */
int lsdr_wgn_run(lsdr_wgn *w, float stddev, const lsdr_cf32 *add, lsdr_cf32 *out, size_t n);
/* adder<cf32>::run, dsp.h:125-134 */
int lsdr_adder_run(lsdr_ctx *ctx, const lsdr_cf32 *a, const lsdr_cf32 *b, size_t n, lsdr_cf32 *out);
/* cconverter<f32,0,u8,128,1,1>::run, dsp.h:40-50 (x86 float → int32 → u8 truncation) */
int lsdr_cconverter_f32_u8_run(lsdr_ctx *ctx, const lsdr_cf32 *in, size_t n, lsdr_cu8 *out);
/* cconverter<f32,0,int16_t,0,32768,1>::run (leandvbtx --s16, leandvbtx.cc:179): out = interleaved (re, im) int16 */
int lsdr_cconverter_f32_s16_run(lsdr_ctx *ctx, const lsdr_cf32 *in, size_t n, int16_t *out);
typedef struct lsdr_drifter lsdr_drifter;                  /* drifter<float>, leanchansim.cc:34-88 */
int lsdr_drifter_create(lsdr_ctx *ctx, lsdr_drifter **d);
void lsdr_drifter_destroy(lsdr_drifter *d);
int lsdr_drifter_set_component(lsdr_drifter *d, int i, float amp, float freq);   /* public member drifts[i] */
int lsdr_drifter_get_phases(lsdr_drifter *d, long long a[3]);
int lsdr_drifter_set_phases(lsdr_drifter *d, const long long a[3]);
/* n samples as consecutive run() calls of `chunk` samples (0: one call).  run() restarts its phase accumulator at 0, so
 * the reference's output depends on how its 4096-sample pipes cut the stream; pass 4096 to reproduce leanchansim. */
int lsdr_drifter_run(lsdr_drifter *d, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out, size_t chunk);

/* ------------------------------------------------------------ capture batch
 * BASELINE configs 1 and 4: B independent cu8 captures, each decoded from its FIRST SAMPLE to transport-stream packets by the blocks of
 * leandvb's default `--u8` graph, freshly constructed per capture (leandvb.cc:157-600: one scheduler per capture):
 *     cconverter<u8,128,f32,0,1,1> → auto_notch<f32>(anf slots, setpoint 0) → cstln_receiver<f32>(linear_sampler, QPSK) →
 *     deconvol_sync<u8,0> → mpeg_sync<u8,0> → deinterleaver<u8> → rs_decoder<u8,0> → derandomizer          (leandvb.cc:211-217,
 *     296-301, 425-510, 521-596; dsp.h:40-50, sdr.h:46-154, 697-938, dvb.h:122-513, 712-891, 926-948, 985-1058, 1107-1163)
 * All captures of a batch share every launch (blockIdx.y = capture) and every data-dependent count stays on the device — the symbols the
 * receiver produced, where mpeg_sync locked, the packets the derandomizer kept: the host reads one result record per capture and batch.
 *  * front end: the time-tiled receiver of LSDR_RX_TILED with packed decisions (tolerance mode, same seam reconciliation), restated for
 *    VALU issue (leansdr_amd/csrc/rxb_device.h), with the one-slot notch of sdr.h:119-138 INSIDE the tile's sample walk:
 *    S[n] = p·S[n−1] + k·x[n], out = x − S, p = (1−k)·exp(j2π·bin/4096) — float32 with exact phases (lsdr_notch_fir's tolerance class);
 *    detect() (sdr.h:76-118) is the reference's FFT bit for bit, at the reference's detect points, on the device.  auto_notch moves
 *    whole 4096-sample blocks: with anf = 1 a capture's last n mod 4096 samples are not demodulated (as in the reference).
 *  * FEC tail: the bit-exact blocks of this ABI, driven on the device the way a scheduler with `unlocked_window`-byte pipes drives them while
 *    mpeg_sync is unlocked (next_sync() included), then in one call each (leansdr_amd/csrc/tail_device.h).
 * anf ∈ {0, 1}; omega = Fs/Fm ∈ [1, 8]; tile_len a multiple of 2048 (0: 4096), tile_warmup a multiple of 128 (0: 512).
 * One batch in flight per object: run_async → wait → [ts_download_async → ts_wait]; the NEXT run_async may be queued while the
 * download is in flight (its last kernel waits for it).  Use two objects on two contexts to overlap one batch's tail with the other's
 * front end. */
typedef struct lsdr_capture_batch lsdr_capture_batch;
typedef struct {
  int n_captures;            /* B */
  size_t max_samples;        /* per capture */
  float omega;               /* samples per symbol, cstln_receiver::set_omega(Fs/Fm), leandvb.cc:482 */
  int fec;                   /* LSDR_FEC12 … (deconvol_sync's rate, leandvb.cc:521-531) */
  int anf;                   /* auto_notch slots: 1 = leandvb's default (leandvb.cc:103), 0 = `--anf 0` */
  unsigned tile_len, tile_warmup;
  float notch_k;             /* auto_notch::k (0: 0.002, sdr.h:56) */
  int notch_decimation;      /* auto_notch::decimation in samples (0: 1024·4096, sdr.h:56) */
  unsigned unlocked_window;  /* bytes deconvol_sync hands mpeg_sync per call while mpeg_sync is not locked = the byte pipe between them
                              * (0: 8192 = leandvb's BUF_BYTES at its default --buf-factor 4, leandvb.cc:194); decides how far the
                              * deconvolver runs ahead of a next_sync() */
  unsigned aux_cus;          /* 0: every kernel on the context's stream.  N (a multiple of 8, < the device's CUs): the object runs on two
                              * streams of its own with disjoint compute-unit masks — the receiver's TILES (vector-issue-bound, they fill
                              * every wave slot they can get for milliseconds) on all CUs but N, everything else (the notch's detect chain
                              * and estimator pre-pass, seam pass, compaction, the whole FEC tail: memory-bound kernels) on N/8 CUs of
                              * every XCD, handed over by events.  Several objects made with the same N share the two partitions: one
                              * batch's tail then runs beside another's tiles instead of waiting for their wave slots. */
} lsdr_capture_batch_cfg;
typedef struct {
  uint64_t ts_packets;       /* packets in the capture's TS buffer */
  uint64_t rs_packets, rs_bit_errors;   /* rs_decoder: packets decoded, bits corrected (dvb.h:1007-1040) */
  uint64_t symbols;          /* decisions the receiver produced */
  uint64_t samples;          /* samples the receiver consumed */
  uint64_t bytes_deconv, bytes_mpeg, first_lock_byte;
  uint32_t next_sync_calls, locked, alignment, bitphase;     /* deconvol_sync::next_sync() calls; mpeg_sync locked at the end; alignment in force */
  uint32_t tiles, seam_dup, seam_miss, seam_bad;             /* receiver seams: duplicates dropped, symbols re-inserted, unreconciled */
} lsdr_capture_result;
int lsdr_capture_batch_create(lsdr_ctx *ctx, const lsdr_capture_batch_cfg *cfg, lsdr_capture_batch **b);
void lsdr_capture_batch_destroy(lsdr_capture_batch *b);
/* iq_dev: HOST array of B DEVICE pointers to lsdr_cu8 items (16-byte aligned), n_samples each.  Queues everything; returns at once. */
int lsdr_capture_batch_run_async(lsdr_capture_batch *b, const lsdr_cu8 *const *iq_dev, size_t n_samples);
/* waits for the batch's kernels; results[B] (may be NULL) */
int lsdr_capture_batch_wait(lsdr_capture_batch *b, lsdr_capture_result *results);
/* after wait: copies every capture's TS (ts_packets·188 bytes) to ts_host[i] (pinned memory recommended) on a side stream */
int lsdr_capture_batch_ts_download_async(lsdr_capture_batch *b, uint8_t *const *ts_host, size_t cap_bytes);
int lsdr_capture_batch_ts_wait(lsdr_capture_batch *b);
const uint8_t *lsdr_capture_batch_ts_dev(const lsdr_capture_batch *b, int i);          /* device buffer of capture i's TS */
const uint32_t *lsdr_capture_batch_words_dev(const lsdr_capture_batch *b, int i);     /* its packed decisions (16 per word, MSB first) */
const uint8_t *lsdr_capture_batch_bytes_dev(const lsdr_capture_batch *b, int i);      /* deconvol_sync's output, mpeg_sync's output */
const uint8_t *lsdr_capture_batch_mpeg_dev(const lsdr_capture_batch *b, int i);
/* the notch's detected bins of capture i after a run (tests): bins[0 … *n) */
int lsdr_capture_batch_bins(lsdr_capture_batch *b, int i, int *bins, unsigned cap, unsigned *n);
/* tests: the NOTCHED stream of capture i exactly as the tiles of the last run saw it — the same start state per tile (from the estimator
 * pre-pass), the same interval switches, the same recurrence — written to device memory as n cf32 items (anf = 1 only; synchronous) */
int lsdr_capture_batch_notched(lsdr_capture_batch *b, int i, lsdr_cf32 *out_dev, size_t n);
/* HIP events around the tile kernel of every run while enabled: mean duration since the previous call, then sets the switch */
int lsdr_capture_batch_tile_time(lsdr_capture_batch *b, int enable, float *avg_ms, unsigned *launches);

#ifdef __cplusplus
}
#endif
#endif /* LSDR_HIP_H */

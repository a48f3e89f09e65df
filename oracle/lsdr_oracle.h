/* oracle/lsdr_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the leandvb IQ hot path of pabr/leansdr
 * (SURVEY.md §8a), written from the reference's behaviour, every function
 * citing the reference file:line it follows (paths relative to
 * /root/reference/src/leansdr unless stated).  Compiled -O3 -ffp-contract=off
 * so that float arithmetic is evaluated exactly like the reference's x86-64
 * SSE2 build (no FMA, no reassociation).
 *
 * PARITY PINNING: this oracle is pinned against the real reference compiled
 * here (oracle/_ref/libleansdr_ref.so, built by oracle/Makefile from
 * oracle/ref_harness.cc + the reference headers) by tests/test_oracle_vs_ref.py,
 * and against the golden vectors under tests/golden/ (generated from the real
 * reference by oracle/make_golden.py) by tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (leansdr_amd/) never links or imports it.
 */
#ifndef LSDR_ORACLE_H
#define LSDR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } lo_cf32;
typedef struct { uint8_t re, im; } lo_cu8;
/* sdr.h:287-290; 4 bytes, pad byte written as 0 by the oracle */
typedef struct { int16_t cost; uint8_t symbol; uint8_t pad; } lo_softsymbol;

/* ---- tables ------------------------------------------------------------ */
void lo_trig16(lo_cf32 *lut /*[65536]*/);                 /* math.h:95-106 */
unsigned lo_trig16_index(float a);                        /* math.h:108-110 */

enum { LO_BPSK, LO_QPSK, LO_PSK8, LO_APSK16, LO_APSK32, LO_APSK64E,
       LO_QAM16, LO_QAM64, LO_QAM256 };                   /* sdr.h:318-324 */
/* code_rate, dvb.h:33-37 */
enum { LO_FEC12, LO_FEC23, LO_FEC46, LO_FEC34, LO_FEC56, LO_FEC78,
       LO_FEC45, LO_FEC89, LO_FEC910 };

typedef struct {
  int nsymbols, nrotations;
  int8_t symbols[256][2];
  int16_t cost[65536];          /* index [(u8)I][(u8)Q] = I8*256+Q8 */
  uint8_t symbol[65536];
  int16_t phase_error[65536];
} lo_cstln_lut;
/* sdr.h:326-468 + make_lut_from_symbols sdr.h:529-560 */
int lo_cstln_lut_init(lo_cstln_lut *c, int predef, float g1, float g2, float g3);
/* dvb.h:45-81 */
int lo_make_dvbs2_constellation(lo_cstln_lut *c, int predef, int fec);
void lo_cstln_harden(lo_cstln_lut *c);                    /* sdr.h:564-571 */
unsigned lo_cstln_lookup_index(float I, float Q);         /* sdr.h:470-482 */

int lo_lowpass(int order, float Fcut, float *coeffs, float gain); /* filtergen.h:45-62 */
int lo_root_raised_cosine(int order, float Fs, float rolloff, float *coeffs); /* filtergen.h:68-92 */
void lo_normalize_dcgain(int n, float *c, float gain);    /* filtergen.h:34-40 */
void lo_normalize_power(int n, float *c, float gain);     /* filtergen.h:26-32 */

/* ---- streaming blocks --------------------------------------------------- */
void lo_cconverter_u8(const lo_cu8 *in, size_t n, lo_cf32 *out);   /* dsp.h:40-50 */
void lo_scaler(float scale, const lo_cf32 *in, size_t n, lo_cf32 *out); /* dsp.h:149-156 */
size_t lo_decimator(unsigned d, const lo_cf32 *in, size_t n, lo_cf32 *out, size_t cap); /* generic.h:256-262 */

/* fir_filter<cf32,float>: dsp.h:219-285 */
void lo_fir_shift_coeffs(unsigned ncoeffs, const float *coeffs, float freq,
                         lo_cf32 *shifted);                /* set_freq dsp.h:271-280 */
/* run() over one buffer, dsp.h:233-262.  *consumed = count*decim. returns count. */
size_t lo_fir_filter(unsigned ncoeffs, const lo_cf32 *shifted, unsigned decim,
                     const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                     size_t *consumed);
/* the same loop with fused multiply-adds: the stated arithmetic of LSDR_FIR_FMA / LSDR_FIR_MFMA */
size_t lo_fir_filter_fma(unsigned ncoeffs, const lo_cf32 *shifted, unsigned decim,
                     const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                     size_t *consumed);
/* blocks of `decim` taps: an fmaf chain per block, block sums added in order — the stated arithmetic of LSDR_FIR_MFMA_BLK
 * (complex taps: a chain takes its taps four at a time, the re-part products of a group before its im-part products) */
size_t lo_fir_filter_blk(unsigned ncoeffs, const lo_cf32 *shifted, unsigned decim,
                     const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                     size_t *consumed);
/* fir_resampler<cf32,float> (interpolator): dsp.h:290-364 */
void lo_fir_resampler_shift_coeffs(unsigned ncoeffs, const float *coeffs, float freq,
                                   lo_cf32 *shifted);      /* dsp.h:351-360 */
size_t lo_fir_resampler(unsigned ncoeffs, const lo_cf32 *shifted, int interp,
                        const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                        size_t *consumed);

/* cfft_engine<float>::inplace, dsp.h:56-116 (n power of 2) */
void lo_cfft(int n, lo_cf32 *data, int reverse);

/* auto_notch<f32>: sdr.h:46-154 */
typedef struct lo_auto_notch lo_auto_notch;
lo_auto_notch *lo_auto_notch_new(int nslots, int decimation, float k, float agc_rms_setpoint);
void lo_auto_notch_free(lo_auto_notch *a);
/* processes floor(n/4096) blocks; returns samples consumed (= produced) */
size_t lo_auto_notch_run(lo_auto_notch *a, const lo_cf32 *in, size_t n, lo_cf32 *out);
int lo_auto_notch_slot_bin(const lo_auto_notch *a, int slot);

/* cnr_fft<f32>: sdr.h:1273-1345 */
typedef struct lo_cnr_fft lo_cnr_fft;
lo_cnr_fft *lo_cnr_fft_new(float bandwidth, int nfft, int decimation);
void lo_cnr_fft_free(lo_cnr_fft *c);
size_t lo_cnr_fft_run(lo_cnr_fft *c, float freq_tap, float tap_multiplier,
                      const lo_cf32 *in, size_t n, float *out, size_t cap);

/* rotator<f32>: sdr.h:1226-1259 (state: 16-bit table index) */
typedef struct lo_rotator lo_rotator;
lo_rotator *lo_rotator_new(float freq);
void lo_rotator_free(lo_rotator *r);
void lo_rotator_run(lo_rotator *r, const lo_cf32 *in, size_t n, lo_cf32 *out);

/* spectrum<f32>: sdr.h:1347-1404 (nfft = 1024) */
typedef struct lo_spectrum lo_spectrum;
lo_spectrum *lo_spectrum_new(int decimation, float kavg);
void lo_spectrum_free(lo_spectrum *c);
size_t lo_spectrum_run(lo_spectrum *c, const lo_cf32 *in, size_t n, float *out /*[cap][1024]*/, size_t cap);

/* ---- cstln_receiver<f32>: sdr.h:697-938 ---------------------------------- */
enum { LO_SAMP_NEAREST, LO_SAMP_LINEAR, LO_SAMP_FIR };
typedef struct {
  int sampler;            /* LO_SAMP_* (sdr.h:600-689) */
  int ncoeffs;            /* fir_sampler prototype */
  const float *coeffs;
  int subsampling;
  int cstln, fec;         /* make_dvbs2_constellation(cstln, fec) */
  float omega;            /* set_omega(), sdr.h:738-743 */
  float freq;             /* set_freq(), sdr.h:745-749 (0 = untouched) */
  float pll_adjustment;
  int allow_drift;
  unsigned long meas_decimation;
  float kest;
} lo_rx_params;

typedef struct {
  float mu, phase, freqw, agc_gain, est_insp, est_sp, est_ep, freq_tap;
  float min_freqw, max_freqw;
  unsigned long meas_count;
  float hist[12];         /* hist[k] = {p.re,p.im,c.re,c.im}, k=0..2 (sdr.h:923-926) */
} lo_rx_state;

typedef struct lo_rx lo_rx;
lo_rx *lo_rx_new(const lo_rx_params *p);
void lo_rx_free(lo_rx *r);
/* run() over one buffer (sdr.h:772-916): consumes whole chunks of 128 while
 * n_in-pos >= 128+readahead and cap-produced >= 128.  Measurement outputs are
 * appended to freq/ss/mer (one float per meas_decimation samples) and
 * cstln_out (one cf32 per chunk that produced >=1 symbol); any may be NULL. */
size_t lo_rx_run(lo_rx *r, const lo_cf32 *in, size_t n_in,
                 lo_softsymbol *out, size_t cap, size_t *consumed,
                 float *freq_out, float *ss_out, float *mer_out, size_t meas_cap, size_t *n_meas,
                 lo_cf32 *cstln_out, size_t cstln_cap, size_t *n_cstln);
void lo_rx_get_state(const lo_rx *r, lo_rx_state *st);
void lo_rx_set_state(lo_rx *r, const lo_rx_state *st);
int lo_rx_readahead(const lo_rx *r);

/* ---- FEC tail (lsdr_oracle_fec.c): dvb.h, viterbi.h, rs.h ------------------------ */
typedef struct lo_deconv lo_deconv;                       /* deconvol_sync<u8,0>, dvb.h:122-476 */
lo_deconv *lo_deconv_new(int rate, int fastlock);         /* make_deconvol_sync_simple, dvb.h:480-513 */
void lo_deconv_free(lo_deconv *d);
void lo_deconv_info(const lo_deconv *d, uint64_t *deconv, uint64_t *deconv2, int *period, int *weight);
void lo_deconv_next_sync(lo_deconv *d);                   /* dvb.h:185-193 */
int lo_deconv_locked(const lo_deconv *d);
/* one run() call (dvb.h:419-470) */
size_t lo_deconv_run(lo_deconv *d, const lo_softsymbol *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed);

typedef struct lo_viterbi lo_viterbi;                     /* viterbi_sync, dvb.h:1173-1416 */
lo_viterbi *lo_viterbi_new(int cstln, int rate);
void lo_viterbi_free(lo_viterbi *s);
void lo_viterbi_set_resync_period(lo_viterbi *s, int p);
int lo_viterbi_current_sync(const lo_viterbi *s);
int lo_viterbi_nsyncs(const lo_viterbi *s);
void lo_viterbi_map(const lo_viterbi *s, int sync, uint8_t *map256);
size_t lo_viterbi_run(lo_viterbi *s, const lo_softsymbol *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed);

typedef struct lo_mpeg_sync lo_mpeg_sync;                 /* mpeg_sync<u8,0>, dvb.h:712-891 */
lo_mpeg_sync *lo_mpeg_sync_new(int fastlock);
void lo_mpeg_sync_free(lo_mpeg_sync *m);
int lo_mpeg_sync_locked(const lo_mpeg_sync *m);
void lo_mpeg_sync_set_resync_period(lo_mpeg_sync *m, int p);   /* public member, dvb.h:717 */
size_t lo_mpeg_sync_run(lo_mpeg_sync *m, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed,
                        int *state_out, size_t state_cap, size_t *n_state,
                        unsigned long *locktime_out, size_t lt_cap, size_t *n_lt, int *call_next_sync);

size_t lo_deinterleaver(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_packets, size_t *consumed); /* dvb.h:926-948 */
void lo_rs_tables(uint8_t *exp512, uint8_t *log256, uint8_t *G17);                   /* rs.h:47-105 */
void lo_rs_encode(uint8_t *msg204);                                                     /* rs.h:141-167 */
size_t lo_rs_decoder(uint8_t *in, size_t npackets, uint8_t *out, long *bits, long *errs); /* dvb.h:998-1053, rs.h:116-270 */
void lo_derandomizer_pattern(uint8_t *pattern1504);                                     /* dvb.h:1116-1129 */
typedef struct lo_derandomizer lo_derandomizer;
lo_derandomizer *lo_derandomizer_new(void);
void lo_derandomizer_free(lo_derandomizer *d);
size_t lo_derandomizer_run(lo_derandomizer *d, const uint8_t *in, size_t npackets, uint8_t *out); /* dvb.h:1130-1157 */
/* symbols → TS packets, the tail of leandvb.cc:519-596 */
size_t lo_fec_chain(int cstln, int rate, int viterbi, int fastlock, const lo_softsymbol *sym, size_t n,
                    uint8_t *ts_out, size_t cap_packets, long *bits, long *errs);

/* ---- transmit chain of leandvbtx (lsdr_oracle_tx.c) ------------------------------------------ */
size_t lo_randomizer(unsigned *pos, const uint8_t *in, size_t npackets, uint8_t *out);                 /* dvb.h:1063-1102 */
size_t lo_interleaver(const uint8_t *in_packets, size_t npackets, uint8_t *out, size_t cap_bytes, size_t *consumed); /* dvb.h:899-921 */
typedef struct lo_convol lo_convol;                                                                     /* dvb.h:567-604 */
lo_convol *lo_convol_new(int rate, int bits_per_symbol);
void lo_convol_free(lo_convol *c);
size_t lo_convol_run(lo_convol *c, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed);
void lo_cstln_transmitter(const lo_cstln_lut *c, const uint8_t *sym, size_t n, lo_cf32 *out);          /* sdr.h:1196-1222 */
size_t lo_simple_agc(float *estimated, float out_rms, float bw, const lo_cf32 *in, size_t n, lo_cf32 *out); /* sdr.h:238-274 */

/* ---- channel simulator of leanchansim (lsdr_oracle_chan.c) --------------------------------- */
uint64_t lo_drand48_default(void);                        /* glibc's initial drand48 state */
uint64_t lo_srand48(long seed);
double lo_drand48(uint64_t *x);
float lo_logf(float x);                                   /* glibc 2.35 logf restated (positive normal x) */
void lo_logf_table(double *tab32, double *ln2, double *poly3);
void lo_wgn(uint64_t *state, float stddev, lo_cf32 *out, size_t n);                                  /* dsp.h:164-190 */
float lo_db_to_amp(double db);
void lo_adder(const lo_cf32 *a, const lo_cf32 *b, size_t n, lo_cf32 *out);                           /* dsp.h:118-138 */
void lo_cconv_f32_u8(const lo_cf32 *in, size_t n, lo_cu8 *out);                                      /* dsp.h:33-54 */
void lo_cconv_f32_s16(const lo_cf32 *in, size_t n, int16_t *out);                                   /* leandvbtx.cc:179 */
void lo_drifter_trig(lo_cf32 *lut65536);                                                             /* leanchansim.cc:42-46 */
void lo_drifter_run(const lo_cf32 *lut, const float amp[3], const float freq[3], long a[3], const lo_cf32 *in, size_t n,
                    lo_cf32 *out);                                                                   /* leanchansim.cc:57-80 */

/* ---- `--hs` path (lsdr_oracle_hs.c) ------------------------------------------------------ */
typedef struct lo_fastqpsk lo_fastqpsk;                   /* fast_qpsk_receiver<u8>, sdr.h:946-1189 */
lo_fastqpsk *lo_fastqpsk_new(float omega, float freq, float pll_adjustment, int allow_drift, unsigned long meas_decimation);
void lo_fastqpsk_free(lo_fastqpsk *r);
void lo_fastqpsk_set_omega(lo_fastqpsk *r, float omega);
void lo_fastqpsk_set_freq(lo_fastqpsk *r, float freq);
void lo_fastqpsk_tables(const lo_fastqpsk *r, uint16_t *polar_a, uint8_t *polar_r, uint8_t *rect, uint8_t *sincos);
void lo_fastqpsk_get_state(const lo_fastqpsk *r, float *mu, unsigned *phase, long *freqw, long *min_freqw, long *max_freqw);
size_t lo_fastqpsk_run(lo_fastqpsk *r, const lo_cu8 *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed,
                       float *freq_out, size_t freq_cap, size_t *n_freq, lo_cu8 *cstln_out, size_t cstln_cap, size_t *n_cstln);
typedef struct lo_hsdeconv lo_hsdeconv;                   /* dvb_deconvol_sync<u8>, dvb.h:612-707 */
lo_hsdeconv *lo_hsdeconv_new(int resync_period);
void lo_hsdeconv_free(lo_hsdeconv *d);
int lo_hsdeconv_locked(const lo_hsdeconv *d);
size_t lo_hsdeconv_run(lo_hsdeconv *d, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed);

#ifdef __cplusplus
}
#endif
#endif

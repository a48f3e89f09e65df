// leansdr_amd/host/leansdr/sdr.h — SDR blocks with the reference's class surface
// (sdr.h:28-34 typedefs, :287-290 softsymbol, :299-573 cstln_lut, :589-689 samplers,
// :697-938 cstln_receiver).  The receiver's run() is one C-ABI call; the constellation
// table and the samplers are descriptors of what the device executes.
#ifndef LEANSDR_AMD_SDR_H
#define LEANSDR_AMD_SDR_H

#include "leansdr/dsp.h"
#include "leansdr/math.h"

namespace leansdr {

typedef float f32;
typedef complex<u8> cu8;
typedef complex<s8> cs8;
typedef complex<u16> cu16;
typedef complex<s16> cs16;
typedef complex<f32> cf32;

typedef uint16_t u_angle;
typedef int16_t s_angle;

struct softsymbol {
  int16_t cost;
  uint8_t symbol;
};

const float cstln_amp = 75;

// Constellation descriptor.  Tables come from the C ABI builder (same values as the
// reference's constructor); lookup()/harden() keep their meaning for host-side users
// (e.g. a constellation viewer) while the receiver itself uses the device copy.
template <int R>
struct cstln_lut {
  enum predef { BPSK, QPSK, PSK8, APSK16, APSK32, APSK64E, QAM16, QAM64, QAM256 };
  struct result {
    struct softsymbol ss;
    s_angle phase_error;
  };
  complex<signed char> *symbols;
  int nsymbols;
  int nrotations;
  predef type;
  int fec;        // code rate used for the APSK radii (make_dvbs2_constellation)
  bool hardened;

  cstln_lut(predef t, int code_rate = 0) : type(t), fec(code_rate), hardened(false) {
    static_assert(R == 256, "cstln_lut<256> only");
    int16_t *cost = new int16_t[65536];
    uint8_t *sym = new uint8_t[65536];
    int16_t *pe = new int16_t[65536];
    int8_t pts[512];
    int n = lsdr_cstln_lut_build((int)t, code_rate, cost, sym, pe, pts, &nrotations);
    if (n < 0) fail("Constellation / code rate not supported");
    nsymbols = n;
    symbols = new complex<signed char>[n];
    for (int s = 0; s < n; ++s) symbols[s] = complex<signed char>(pts[2 * s], pts[2 * s + 1]);
    lut = new result[65536];
    for (int i = 0; i < 65536; ++i) {
      lut[i].ss.cost = cost[i];
      lut[i].ss.symbol = sym[i];
      lut[i].phase_error = pe[i];
    }
    delete[] cost; delete[] sym; delete[] pe;
  }
  inline result *lookup(float I, float Q) {
    while (I < -128 || I > 127 || Q < -128 || Q > 127) { I *= 0.5; Q *= 0.5; }
    return &lut[(unsigned)(u8)(s8)I * 256 + (u8)(s8)Q];
  }
  inline result *lookup(int I, int Q) { return &lut[(unsigned)(u8)I * 256 + (u8)Q]; }
  void harden() {
    hardened = true;
    for (int i = 0; i < 65536; ++i) {
      if (lut[i].ss.cost < 0) lut[i].ss.cost = -1;
      if (lut[i].ss.cost > 0) lut[i].ss.cost = 1;
    }
  }

 private:
  result *lut;
};

// Options a graph builder written for the reference has no members for come from the environment: LSDR_TILED=1 selects
// the throughput (time-tiled, tolerance-tested) mode of the receivers, LSDR_TILE_LEN / LSDR_TILE_WARMUP its geometry.
inline bool env_flag(const char *name) {
  const char *e = getenv(name);
  return e && atoi(e) != 0;
}
inline unsigned env_uint(const char *name) {
  const char *e = getenv(name);
  return e ? (unsigned)strtoul(e, NULL, 10) : 0u;
}

// constellation names by cstln_lut<256>::predef (sdr.h:575-585), printed by --fd-info
static const char *const cstln_names[] = {"BPSK", "QPSK", "8PSK", "16APSK", "32APSK", "64APSKe", "16QAM", "64QAM", "256QAM"};

// auto_notch<f32> (sdr.h:46-154): same constructor and public tunables (decimation, k).
template <typename T>
struct auto_notch;

template <>
struct auto_notch<f32> : runnable, notch_tap_point {
  int decimation;
  float k;
  auto_notch(scheduler *sch, pipebuf<cf32> &i, pipebuf<cf32> &o, int nslots, f32 agc_rms_setpoint)
      : runnable(sch, "auto_notch"), decimation(1024 * 4096), k(0.002),
        ctx(pipe_ctx(i, o, "auto_notch: pipebufs of two device contexts")), in(i), out(o, 4096), h(NULL), nslots_(nslots),
        setpoint_(agc_rms_setpoint) {
    o.fusable_producer = static_cast<notch_tap_point *>(this);      // a fir_filter reading `o` may take this block over (dsp.h, LSDR_FUSE_NOTCH)
    lsdr_check(lsdr_auto_notch_create(ctx, nslots, agc_rms_setpoint, &h), name);
    // throughput mode (single-pass scan, detect() on the device) where it applies: leandvb's configuration (no AGC set point)
    if (env_flag("LSDR_TILED") && agc_rms_setpoint == 0 && nslots >= 1 && nslots <= 4) lsdr_check(lsdr_auto_notch_set_mode(h, LSDR_NOTCH_SCAN), name);
  }
  void set_throughput_mode() { lsdr_check(lsdr_auto_notch_set_mode(h, LSDR_NOTCH_SCAN), name); }
  // notch_tap_point
  dev_reader<cf32> *raw_input() { return &in; }
  int notch_slots() const { return nslots_; }
  float notch_setpoint() const { return setpoint_; }
  int notch_decimation() const { return decimation; }
  float notch_k() const { return k; }
  void run() {
    if (fused_away) return;          // a fir_filter runs this block's work inside its own (lsdr_notch_fir)
    lsdr_check(lsdr_auto_notch_set(h, decimation, k), name);
    unsigned long room = out.writable();
    size_t consumed = 0, produced = 0;
    lsdr_check(lsdr_auto_notch_run(h, (const lsdr_cf32 *)in.rd(), in.readable(), (lsdr_cf32 *)out.wr(), room, &consumed, &produced), name);
    in.read(consumed);
    out.written(produced);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<cf32> in;
  dev_writer<cf32> out;
  lsdr_auto_notch *h;
  int nslots_;
  float setpoint_;
};

// cnr_fft<f32> (sdr.h:1273-1345): device input pipe, host float output pipe.
template <typename T>
struct cnr_fft;

template <>
struct cnr_fft<f32> : runnable {
  float bandwidth;
  float *freq_tap, tap_multiplier;
  int decimation;
  float kavg;
  cnr_fft(scheduler *sch, pipebuf<cf32> &i, pipebuf<float> &o, float bw, int nfft = 4096)
      : runnable(sch, "cnr_fft"), bandwidth(bw), freq_tap(NULL), tap_multiplier(1), decimation(1048576), kavg(0.1),
        ctx(pipe_ctx(i, "cnr_fft")), in(i), out(o), h(NULL) {
    lsdr_check(lsdr_cnr_fft_create(ctx, bw, nfft, &h), name);
  }
  void run() {
    lsdr_check(lsdr_cnr_fft_set(h, decimation, kavg), name);
    unsigned long room = out.writable();
    size_t consumed = 0, produced = 0;
    lsdr_check(lsdr_cnr_fft_run(h, freq_tap ? *freq_tap : 0.f, tap_multiplier, (const lsdr_cf32 *)in.rd(), in.readable(),
                                out.wr(), room, &consumed, &produced), name);
    in.read(consumed);
    out.written(produced);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<cf32> in;
  pipewriter<float> out;
  lsdr_cnr_fft *h;
};

// fast_qpsk_receiver<u8> (sdr.h:946-1189), the --hs receiver: cu8 device pipe in, hard-symbol device pipe out, FREQ and
// sampled-constellation reports to host pipes.
template <typename T>
struct fast_qpsk_receiver;

template <>
struct fast_qpsk_receiver<u8> : runnable {
  typedef u8 hardsymbol;
  unsigned long meas_decimation;
  float omega, min_omega, max_omega;
  signed long freqw, min_freqw, max_freqw;
  float pll_adjustment;
  bool allow_drift;
  static const unsigned int chunk_size = 128;
  bool tiled;               // addition: throughput mode (time-tiled, tolerance) instead of the exact recurrence
  unsigned tile_len, tile_warmup;

  fast_qpsk_receiver(scheduler *sch, pipebuf<cu8> &i, pipebuf<hardsymbol> &o, pipebuf<float> *freq_o = NULL,
                     pipebuf<cu8> *cstln_o = NULL)
      : runnable(sch, "Fast QPSK receiver"), meas_decimation(1048576), pll_adjustment(1.0), allow_drift(false),
        tiled(env_flag("LSDR_TILED")), tile_len(env_uint("LSDR_TILE_LEN")), tile_warmup(env_uint("LSDR_TILE_WARMUP")),
        ctx(pipe_ctx(i, o, "fast_qpsk_receiver: in/out of two device contexts")), in(i), out(o, chunk_size),
        h(NULL), freq0(0) {
    set_omega(1);
    set_freq(0);
    freq_out = opt_writer(freq_o);
    cstln_out = opt_writer(cstln_o);
  }
  void set_omega(float o, float tol = 10e-6) {
    omega = o;
    min_omega = omega * (1 - tol);
    max_omega = omega * (1 + tol);
    update_freq_limits();
  }
  void set_freq(float f) {
    freq0 = f;
    freqw = f * 65536;
    update_freq_limits();
  }
  void update_freq_limits() {   // ±SR/8 (sdr.h:987-992)
    min_freqw = freqw - 65536 / max_omega / 8;
    max_freqw = freqw + 65536 / max_omega / 8;
  }
  void run() {
    if (!h) {
      lsdr_check(lsdr_fastqpsk_create(ctx, omega, freq0, pll_adjustment, allow_drift, meas_decimation, &h), name);
      acquired = 0;
    }
    // throughput mode: the exact recurrence acquires on the head of the stream (first 64 Ki samples), then tiles track
    lsdr_check(lsdr_fastqpsk_set_tiled(h, tiled && acquired >= 65536, tile_len, tile_warmup), name);
    unsigned long max_meas = chunk_size / meas_decimation + 1;
    unsigned long room = out.writable();
    unsigned long freq_room = freq_out ? freq_out->writable() : 0, cstln_room = cstln_out ? cstln_out->writable() : 0;
    if (in.readable() < chunk_size + 1 || room < chunk_size) return;       // sdr.h:1010-1013
    if (freq_out && freq_room < max_meas) return;
    if (cstln_out && cstln_room < max_meas) return;
    size_t consumed = 0, produced = 0, nf = 0, nc = 0;
    unsigned long avail = in.readable();
    if (tiled && acquired < 65536 && avail > 65536 - acquired + 1) avail = 65536 - acquired + 1;   // acquisition stays short
    lsdr_check(lsdr_fastqpsk_run(h, (const lsdr_cu8 *)in.rd(), avail, out.wr(), room, &consumed, &produced,
                                 freq_out ? freq_out->wr() : NULL, freq_room, &nf,
                                 cstln_out ? (lsdr_cu8 *)cstln_out->wr() : NULL, cstln_room, &nc), name);
    in.read(consumed);
    out.written(produced);
    acquired += consumed;
    if (freq_out) freq_out->written(nf);
    if (cstln_out) cstln_out->written(nc);
    long long fw = 0;
    lsdr_check(lsdr_fastqpsk_get_state(h, NULL, NULL, &fw, NULL, NULL), name);
    freqw = (signed long)fw;
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<cu8> in;
  dev_writer<hardsymbol> out;
  pipewriter<float> *freq_out;
  pipewriter<cu8> *cstln_out;
  lsdr_fastqpsk *h;
  float freq0;
  unsigned long acquired;
};

// rotator<f32> (sdr.h:1226-1259): frequency shifter on device pipes.
template <typename T>
struct rotator;

template <>
struct rotator<f32> : runnable {
  rotator(scheduler *sch, pipebuf<cf32> &i, pipebuf<cf32> &o, float freq)
      : runnable(sch, "rotator"), ctx(pipe_ctx(i, o, "rotator: in/out of two device contexts")), in(i), out(o),
        h(NULL) {
    lsdr_check(lsdr_rotator_create(ctx, freq, &h), name);
  }
  void run() {
    unsigned long room = out.writable();
    unsigned long count = min(in.readable(), room);
    if (!count) return;
    lsdr_check(lsdr_rotator_run(h, (const lsdr_cf32 *)in.rd(), count, (lsdr_cf32 *)out.wr()), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<cf32> in;
  dev_writer<cf32> out;
  lsdr_rotator *h;
};

// spectrum<f32> (sdr.h:1347-1404): device input pipe, host output pipe of float[1024] rows.
template <typename T>
struct spectrum;

template <>
struct spectrum<f32> : runnable, passive_tap {
  static const int nfft = 1024;
  int decimation;
  float kavg;
  spectrum(scheduler *sch, pipebuf<cf32> &i, pipebuf<float[nfft]> &o)
      : runnable(sch, "spectrum"), decimation(1048576), kavg(0.1), ctx(pipe_ctx(i, "spectrum")), in(i), out(o), h(NULL), opipe(&o) {
    lsdr_check(lsdr_spectrum_create(ctx, &h), name);
    i.set_reader_owner(in.id, static_cast<passive_tap *>(this));
  }
  bool tap_output_used() { return opipe->n_readers() > 0; }
  void run() {
    if (tap_detached) return;        // nobody reads the rows, and a fir_filter fused with the auto_notch in front (dsp.h): the stream is not there
    lsdr_check(lsdr_spectrum_set(h, decimation, kavg), name);
    unsigned long room = out.writable();
    size_t consumed = 0, produced = 0;
    lsdr_check(lsdr_spectrum_run(h, (const lsdr_cf32 *)in.rd(), in.readable(), (float *)out.wr(), room, &consumed, &produced), name);
    in.read(consumed);
    out.written(produced);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<cf32> in;
  pipewriter<float[nfft]> out;
  lsdr_spectrum *h;
  pipebuf<float[nfft]> *opipe;
};

// Samplers: descriptors consumed by cstln_receiver (the interpolation itself runs on the GPU).
template <typename T>
struct sampler_interface {
  virtual ~sampler_interface() {}
  virtual int kind() = 0;          // LSDR_SAMP_*
  virtual int readahead() { return 0; }
  virtual int ncoeffs() { return 0; }
  virtual float *coeffs() { return NULL; }
  virtual int subsampling() { return 1; }
};
template <typename T>
struct nearest_sampler : sampler_interface<T> {
  int kind() { return LSDR_SAMP_NEAREST; }
  int readahead() { return 0; }
};
template <typename T>
struct linear_sampler : sampler_interface<T> {
  int kind() { return LSDR_SAMP_LINEAR; }
  int readahead() { return 1; }
};
template <typename T, typename Tc>
struct fir_sampler : sampler_interface<T> {
  fir_sampler(int n, Tc *c, int sub = 1) : n_(n), c_(c), sub_(sub) {}
  int kind() { return LSDR_SAMP_FIR; }
  int readahead() { return n_ - 1; }
  int ncoeffs() { return n_; }
  float *coeffs() { return c_; }
  int subsampling() { return sub_; }

 private:
  int n_;
  Tc *c_;
  int sub_;
};

// cstln_receiver<f32>: constructor and public members as sdr.h:697-753,918.  The device
// handle is created lazily at the first run(), after the graph builder has set cstln,
// omega, pll_adjustment, meas_decimation … exactly as leandvb.cc:463-502 does.
template <typename T>
struct cstln_receiver;

template <>
struct cstln_receiver<f32> : runnable {
  sampler_interface<f32> *sampler;
  cstln_lut<256> *cstln;
  unsigned long meas_decimation;
  float omega, min_omega, max_omega;
  float freqw, min_freqw, max_freqw;
  float pll_adjustment;
  bool allow_drift;
  static const unsigned int chunk_size = 128;
  float kest;
  float freq_tap;
  int mode;                 // addition: LSDR_RX_SERIAL (exact, default) or LSDR_RX_TILED
  unsigned tile_len, tile_warmup;

  cstln_receiver(scheduler *sch, sampler_interface<f32> *s, pipebuf<cf32> &i, pipebuf<softsymbol> &o,
                 pipebuf<float> *freq_o = NULL, pipebuf<float> *ss_o = NULL, pipebuf<float> *mer_o = NULL,
                 pipebuf<cf32> *cstln_o = NULL)
      : runnable(sch, "Constellation receiver"), sampler(s), cstln(NULL), meas_decimation(1048576), pll_adjustment(1.0),
        allow_drift(false), kest(0.01), freq_tap(0), mode(env_flag("LSDR_TILED") ? LSDR_RX_TILED : LSDR_RX_SERIAL),
        tile_len(env_uint("LSDR_TILE_LEN")), tile_warmup(env_uint("LSDR_TILE_WARMUP")),
        ctx(pipe_ctx(i, o, "cstln_receiver: in/out of two device contexts")), in(i),
        out(o, chunk_size), h(NULL), freq0(0) {
    set_omega(1);
    set_freq(0);
    freq_out = opt_writer(freq_o);
    ss_out = opt_writer(ss_o);
    mer_out = opt_writer(mer_o);
    cstln_out = opt_writer(cstln_o);
  }
  void set_omega(float o, float tol = 10e-6) {
    omega = o;
    min_omega = omega * (1 - tol);
    max_omega = omega * (1 + tol);
    update_freq_limits();
  }
  void set_freq(float f) {
    freq0 = f;
    freqw = f * 65536;
    update_freq_limits();
    freq_tap = freqw / 65536;
  }
  void set_allow_drift(bool d) { allow_drift = d; }
  void update_freq_limits() {
    int n = 4;
    if (cstln) switch (cstln->nsymbols) {
        case 2: n = 2; break;
        case 4: n = 4; break;
        case 8: n = 8; break;
        case 16: n = 12; break;
        case 32: n = 16; break;
        default: n = 4; break;
      }
    min_freqw = freqw - 65536 / max_omega / n / 2;
    max_freqw = freqw + 65536 / max_omega / n / 2;
  }

  void run() {
    if (!cstln) fail("constellation not set");
    if (!h) create();
    unsigned long max_meas = chunk_size / meas_decimation + 1;
    unsigned long room = out.writable();
    unsigned long meas_room = ~0ul;
    if (freq_out) meas_room = min(meas_room, freq_out->writable());
    if (ss_out) meas_room = min(meas_room, ss_out->writable());
    if (mer_out) meas_room = min(meas_room, mer_out->writable());
    if (meas_room > 65536) meas_room = 65536;  // scratch size
    unsigned long cstln_room = cstln_out ? cstln_out->writable() : 0;
    if (in.readable() < chunk_size + sampler->readahead() || room < chunk_size) return;
    if ((freq_out || ss_out || mer_out) && meas_room < max_meas) return;
    if (cstln_out && cstln_room < max_meas) return;
    bool meas = freq_out || ss_out || mer_out;
    size_t consumed = 0, produced = 0, n_meas = 0, n_cstln = 0;
    lsdr_check(lsdr_rx_run(h, (const lsdr_cf32 *)in.rd(), in.readable(), (lsdr_softsymbol *)out.wr(), room, &consumed,
                           &produced, meas ? tmp_meas(0, meas_room) : NULL, meas ? tmp_meas(1, meas_room) : NULL,
                           meas ? tmp_meas(2, meas_room) : NULL, meas ? meas_room : 0, &n_meas,
                           cstln_out ? (lsdr_cf32 *)cstln_out->wr() : NULL, cstln_room, &n_cstln),
               name);
    in.read(consumed);
    out.written(produced);
    for (size_t k = 0; k < n_meas; ++k) {
      if (freq_out) freq_out->write(scratch[0][k]);
      if (ss_out) ss_out->write(scratch[1][k]);
      if (mer_out) mer_out->write(scratch[2][k]);
    }
    if (cstln_out) cstln_out->written(n_cstln);
    lsdr_rx_state st;
    lsdr_check(lsdr_rx_get_state(h, &st), name);
    freqw = st.freqw;
    freq_tap = st.freq_tap;  // read by fir_filter / cnr_fft through their freq_tap pointers
  }

 private:
  void create() {
    lsdr_rx_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.sampler = sampler->kind();
    cfg.ncoeffs = sampler->ncoeffs();
    cfg.coeffs_host = sampler->coeffs();
    cfg.subsampling = sampler->subsampling();
    cfg.cstln = (int)cstln->type;
    cfg.fec = cstln->fec;
    cfg.harden = cstln->hardened;
    cfg.omega = omega;
    cfg.freq = freq0;
    cfg.pll_adjustment = pll_adjustment;
    cfg.allow_drift = allow_drift;
    cfg.meas_decimation = meas_decimation;
    cfg.kest = kest;
    cfg.mode = mode;
    cfg.tile_len = tile_len;
    cfg.tile_warmup = tile_warmup;
    lsdr_check(lsdr_rx_create(ctx, &cfg, &h), name);
  }
  float *tmp_meas(int which, unsigned long n) {
    if (n > 65536) n = 65536;
    if (!scratch[which]) scratch[which] = new float[65536];
    return scratch[which];
  }
  lsdr_ctx *ctx;
  dev_reader<cf32> in;
  dev_writer<softsymbol> out;
  pipewriter<float> *freq_out, *ss_out, *mer_out;
  pipewriter<cf32> *cstln_out;
  lsdr_rx *h;
  float freq0;
  float *scratch[3] = {NULL, NULL, NULL};
};

// cstln_transmitter<f32,0> (sdr.h:1196-1222) and simple_agc<f32> (sdr.h:238-274) on device pipebufs.
template <typename Tout, int Zout>
struct cstln_transmitter;

template <>
struct cstln_transmitter<f32, 0> : runnable {
  cstln_lut<256> *cstln;
  cstln_transmitter(scheduler *sch, pipebuf<u8> &i, pipebuf<cf32> &o)
      : runnable(sch, "cstln_transmitter"), cstln(NULL),
        ctx(pipe_ctx(i, o, "cstln_transmitter: pipebufs of two device contexts")), in(i), out(o) {}
  void run() {
    if (!cstln) fail("constellation not set");
    unsigned long room = out.writable();
    unsigned long count = min(in.readable(), room);
    if (!count) return;
    lsdr_check(lsdr_cstln_transmitter_run(ctx, (int)cstln->type, cstln->fec, in.rd(), count, (lsdr_cf32 *)out.wr()), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<u8> in;
  dev_writer<cf32> out;
};

template <typename T>
struct simple_agc;

template <>
struct simple_agc<f32> : runnable {
  float out_rms, bw;
  simple_agc(scheduler *sch, pipebuf<cf32> &i, pipebuf<cf32> &o)
      : runnable(sch, "AGC"), out_rms(1), bw(0.001), ctx(pipe_ctx(i, o, "simple_agc: pipebufs of two device contexts")),
        in(i), out(o), h(NULL) {}
  void run() {
    if (!h) lsdr_check(lsdr_simple_agc_create(ctx, out_rms, bw, &h), name);
    lsdr_check(lsdr_simple_agc_set(h, out_rms, bw), name);
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_simple_agc_run(h, (const lsdr_cf32 *)in.rd(), in.readable(), (lsdr_cf32 *)out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<cf32> in;
  dev_writer<cf32> out;
  lsdr_simple_agc *h;
};

}  // namespace leansdr
#endif

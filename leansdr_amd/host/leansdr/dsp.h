// leansdr_amd/host/leansdr/dsp.h — DSP blocks with the reference's class surface
// (dsp.h:33-54 cconverter, :140-160 scaler, :219-285 fir_filter) whose run() is one
// call through the C ABI into the HIP kernels.  These blocks attach to the DEVICE side
// of their pipebufs (dev_reader / dev_writer).  There is no CPU implementation here.
#ifndef LEANSDR_AMD_DSP_H
#define LEANSDR_AMD_DSP_H

#include "leansdr/framework.h"
#include "leansdr/math.h"

namespace leansdr {

// cconverter<u8,128,f32,0,1,1>: the only instantiation leandvb uses (leandvb.cc:215).
template <typename Tin, int Zin, typename Tout, int Zout, int Gn, int Gd>
struct cconverter;

template <>
struct cconverter<u8, 128, float, 0, 1, 1> : runnable {
  cconverter(scheduler *sch, pipebuf<complex<u8> > &i, pipebuf<complex<float> > &o)
      : runnable(sch, "cconverter"), ctx(pipe_ctx(i, o, "cconverter: pipebufs of two device contexts")),
        in(i), out(o) {}
  void run() {
    unsigned long count = min(in.readable(), out.writable());
    if (!count) return;
    lsdr_check(lsdr_cconverter_u8_run(ctx, (const lsdr_cu8 *)in.rd(), count, (lsdr_cf32 *)out.wr()), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<complex<u8> > in;
  dev_writer<complex<float> > out;
};

// cconverter<s8,0,…>, <u16,32768,…>, <s16,0,…> → f32: the other input formats of leandvb (leandvb.cc:218-248).
namespace detail {
template <typename Tin, int FMT>
struct int_cconverter : runnable {
  int_cconverter(scheduler *sch, pipebuf<complex<Tin> > &i, pipebuf<complex<float> > &o)
      : runnable(sch, "cconverter"), ctx(pipe_ctx(i, o, "cconverter: pipebufs of two device contexts")), in(i), out(o) {}
  void run() {
    unsigned long count = min(in.readable(), out.writable());
    if (!count) return;
    lsdr_check(lsdr_cconverter_int_run(ctx, FMT, in.rd(), count, (lsdr_cf32 *)out.wr()), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<complex<Tin> > in;
  dev_writer<complex<float> > out;
};
}  // namespace detail
template <>
struct cconverter<s8, 0, float, 0, 1, 1> : detail::int_cconverter<s8, LSDR_IN_CS8> {
  cconverter(scheduler *sch, pipebuf<complex<s8> > &i, pipebuf<complex<float> > &o) : detail::int_cconverter<s8, LSDR_IN_CS8>(sch, i, o) {}
};
template <>
struct cconverter<u16, 32768, float, 0, 1, 1> : detail::int_cconverter<u16, LSDR_IN_CU16> {
  cconverter(scheduler *sch, pipebuf<complex<u16> > &i, pipebuf<complex<float> > &o) : detail::int_cconverter<u16, LSDR_IN_CU16>(sch, i, o) {}
};
template <>
struct cconverter<s16, 0, float, 0, 1, 1> : detail::int_cconverter<s16, LSDR_IN_CS16> {
  cconverter(scheduler *sch, pipebuf<complex<s16> > &i, pipebuf<complex<float> > &o) : detail::int_cconverter<s16, LSDR_IN_CS16>(sch, i, o) {}
};

template <typename Tscale, typename Tin, typename Tout>
struct scaler;

template <>
struct scaler<float, complex<float>, complex<float> > : runnable {
  float scale;
  scaler(scheduler *sch, float s, pipebuf<complex<float> > &i, pipebuf<complex<float> > &o)
      : runnable(sch, "scaler"), scale(s), ctx(pipe_ctx(i, o, "scaler: pipebufs of two device contexts")),
        in(i), out(o) {}
  void run() {
    unsigned long count = min(in.readable(), out.writable());
    if (!count) return;
    lsdr_check(lsdr_scaler_run(ctx, scale, (const lsdr_cf32 *)in.rd(), count, (lsdr_cf32 *)out.wr()), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<complex<float> > in;
  dev_writer<complex<float> > out;
};

// fir_filter<cf32,float> — same constructor and public tracking members as dsp.h:219-285.
// `fuse_u8` / `fuse_scale` (additions) let a graph builder drop the cconverter / scaler
// block in front and have the filter read the raw stream (bit-identical result).
template <typename T, typename Tc>
struct fir_filter;

// What an auto_notch block (sdr.h) offers a fir_filter that reads its output — leandvb's default graph, leandvb.cc:296-301,353-382 — so that
// the two can run as ONE block (lsdr_notch_fir: the notched stream never exists).  Opt-in (LSDR_FUSE_NOTCH=1): a tolerance mode.
struct notch_tap_point {
  virtual ~notch_tap_point() {}
  virtual dev_reader<complex<float> > *raw_input() = 0;     // the notch's own input end: the fused block reads the raw stream there
  virtual int notch_slots() const = 0;
  virtual float notch_setpoint() const = 0;
  virtual int notch_decimation() const = 0;
  virtual float notch_k() const = 0;
  bool fused_away;                                           // set by the fir_filter that took over: the notch's run() does nothing
  notch_tap_point() : fused_away(false) {}
};
// A reader of the notched stream that only observes it (spectrum: leandvb.cc:335-343 instantiates one ALWAYS — `if (cfg.fd_spectrum)` with
// a default of −1 — and nobody reads its rows unless --fd-spectrum is given): while nobody reads its output it can be switched off, and
// the notched stream need not exist for its sake.
struct passive_tap {
  virtual ~passive_tap() {}
  virtual bool tap_output_used() = 0;
  bool tap_detached;
  passive_tap() : tap_detached(false) {}
};

template <>
struct fir_filter<complex<float>, float> : runnable {
  float *freq_tap;        // → cstln_receiver::freq_tap (leandvb.cc:506-510)
  float tap_multiplier;
  float freq_tol;

  fir_filter(scheduler *sch, int ncoeffs, float *coeffs, pipebuf<complex<float> > &i, pipebuf<complex<float> > &o,
             unsigned int decim = 1, float fuse_scale = 0)
      : runnable(sch, "fir_filter"), freq_tap(NULL), tap_multiplier(1), freq_tol(0.1),
        ctx(pipe_ctx(i, o, "fir_filter: pipebufs of two device contexts")), n(ncoeffs), d(decim), in(i), out(o), h(NULL),
        hf(NULL), notch(NULL), fused_pipe(&i), fused_ready(false) {
    // LSDR_FUSE_NOTCH=1: the pipe's writer is an auto_notch the fused block exists for (one slot, no AGC set point, decimation 30,
    // ncoeffs ≤ 330): take it over.  Anything else — and every graph without the variable — keeps the two blocks.
    // The decision is prepare()'s: only then is every reader of the notched stream known.
    const char *fe = getenv("LSDR_FUSE_NOTCH");
    if (fe && atoi(fe) != 0 && i.fusable_producer) {
      notch_tap_point *np = static_cast<notch_tap_point *>(i.fusable_producer);
      lsdr_notch_fir_cfg fc;
      fc.ncoeffs = ncoeffs; fc.coeffs_host = coeffs; fc.decim = decim; fc.in_scale = fuse_scale; fc.nslots = np->notch_slots();
      fc.notch_decimation = 0; fc.k = 0;
      if (np->notch_setpoint() == 0 && lsdr_notch_fir_create(ctx, &fc, &hf) == LSDR_OK) notch = np;
      else hf = NULL;
    }
    lsdr_fir_filter_cfg cfg;
    cfg.ncoeffs = ncoeffs; cfg.coeffs_host = coeffs; cfg.decim = decim;
    cfg.in_format = LSDR_IN_CF32; cfg.in_scale = fuse_scale; cfg.arith = LSDR_FIR_EXACT;
    // LSDR_FIR_ARITH=fma|mfma|blk: the filter's tolerance arithmetics (include/lsdr_hip.h; like LSDR_TILED an option the reference's
    // config has no member for).  A geometry the chosen arithmetic has no kernel for keeps the exact filter.
    const char *ae = getenv("LSDR_FIR_ARITH");
    if (ae && *ae) {
      const int want = !strcmp(ae, "fma") ? LSDR_FIR_FMA : !strcmp(ae, "mfma") ? LSDR_FIR_MFMA : !strcmp(ae, "blk") ? LSDR_FIR_MFMA_BLK : LSDR_FIR_EXACT;
      cfg.arith = want;
      if (want != LSDR_FIR_EXACT && lsdr_fir_filter_create(ctx, &cfg, &h) == LSDR_OK) {
        if (sch->verbose) fprintf(stderr, "fir_filter: arithmetic %s\n", ae);
        return;
      }
      h = NULL; cfg.arith = LSDR_FIR_EXACT;
      if (want != LSDR_FIR_EXACT) fprintf(stderr, "fir_filter: no %s kernel for %d taps / decimation %u: exact arithmetic\n", ae, ncoeffs, decim);
    }
    lsdr_check(lsdr_fir_filter_create(ctx, &cfg, &h), name);
  }
  // (the reference's blocks live as long as the process; a graph that is torn down gives its device objects back)
  ~fir_filter() {
    if (hf) lsdr_notch_fir_destroy(hf);
    if (h) lsdr_fir_filter_destroy(h);
  }
  void prepare() {
    if (!hf) return;
    // every other reader of the notched stream must be an observer nobody listens to; then they are switched off and the notch is ours
    bool ok = !notch->fused_away;
    for (size_t r = 0; ok && r < fused_pipe->n_readers(); ++r) {
      if ((int)r == in.id) continue;
      passive_tap *pt = static_cast<passive_tap *>(fused_pipe->reader_owner((int)r));
      ok = pt != NULL && !pt->tap_output_used();
    }
    if (!ok) { lsdr_notch_fir_destroy(hf); hf = NULL; return; }
    for (size_t r = 0; r < fused_pipe->n_readers(); ++r)
      if ((int)r != in.id) static_cast<passive_tap *>(fused_pipe->reader_owner((int)r))->tap_detached = true;
    notch->fused_away = true;
    lsdr_fir_filter_destroy(h); h = NULL;       // fused: the stand-alone filter's tables and scratch are not needed any more
    out.buf.need_room(4096 / d + 16);          // (the fused block produces a whole 4096-sample block's outputs or nothing)
    lsdr_check(lsdr_notch_fir_set(hf, notch->notch_decimation(), notch->notch_k()), name);     // auto_notch's public tunables, as set by now
    if (sch->verbose) fprintf(stderr, "fir_filter: fused with auto_notch (lsdr_notch_fir)\n");
  }
  void run() {
    if (hf) { run_fused(); return; }
    if (in.readable() < n) return;
    if (freq_tap) {  // dsp.h:236-244
      int shifted = 0;
      float before = lsdr_fir_filter_current_freq(h);
      lsdr_check(lsdr_fir_filter_track(h, *freq_tap, tap_multiplier, freq_tol, &shifted), name);
      if (shifted && sch->verbose)
        fprintf(stderr, "Shifting filter %f -> %f\n", before, lsdr_fir_filter_current_freq(h));
    }
    unsigned long room = out.writable();  // may pack(): take it before the read pointer
    size_t consumed = 0, produced = 0;
    lsdr_check(lsdr_fir_filter_run(h, in.rd(), in.readable(), (lsdr_cf32 *)out.wr(), room, &consumed, &produced), name);
    in.read(consumed);
    out.written(produced);
  }

 private:
  // auto_notch + fir_filter as one block: the raw stream is read at the notch's input end, at this filter's pace
  void run_fused() {
    if (freq_tap) {  // dsp.h:236-244
      int shifted = 0;
      float before = lsdr_notch_fir_current_freq(hf);
      lsdr_check(lsdr_notch_fir_track(hf, *freq_tap, tap_multiplier, freq_tol, &shifted), name);
      if (shifted && sch->verbose)
        fprintf(stderr, "Shifting filter %f -> %f\n", before, lsdr_notch_fir_current_freq(hf));
    }
    dev_reader<complex<float> > *src = notch->raw_input();
    unsigned long room = out.writable();
    size_t consumed = 0, produced = 0;
    lsdr_check(lsdr_notch_fir_run(hf, (const lsdr_cf32 *)src->rd(), src->readable(), (lsdr_cf32 *)out.wr(), room, &consumed, &produced), name);
    if (sch->debug) fprintf(stderr, "fir_filter(fused): readable %lu room %lu -> consumed %lu produced %lu\n", (unsigned long)src->readable(), room,
                            (unsigned long)consumed, (unsigned long)produced);
    src->read(consumed);
    out.written(produced);
  }

  lsdr_ctx *ctx;
  unsigned n, d;
  dev_reader<complex<float> > in;
  dev_writer<complex<float> > out;
  lsdr_fir_filter *h;
  lsdr_notch_fir *hf;               // non-NULL: fused with the auto_notch in front
  notch_tap_point *notch;
  pipebuf<complex<float> > *fused_pipe;
  bool fused_ready;
};

// decimator<cf32> (generic.h:247-267 of the reference) on device pipebufs.
template <typename T>
struct decimator;

template <>
struct decimator<complex<float> > : runnable {
  unsigned int d;
  decimator(scheduler *sch, int _d, pipebuf<complex<float> > &i, pipebuf<complex<float> > &o)
      : runnable(sch, "decimator"), d(_d), ctx(pipe_ctx(i, o, "decimator: pipebufs of two device contexts")),
        in(i), out(o) {}
  void run() {
    unsigned long room = out.writable();
    size_t produced = 0;
    lsdr_check(lsdr_decimator_run(ctx, d, (const lsdr_cf32 *)in.rd(), in.readable(), (lsdr_cf32 *)out.wr(), room, &produced), name);
    in.read(produced * d);
    out.written(produced);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<complex<float> > in;
  dev_writer<complex<float> > out;
};

// fir_resampler<cf32,float> (dsp.h:290-364): polyphase interpolator on device pipebufs (decim must be 1, as in the reference).
template <typename T, typename Tc>
struct fir_resampler;

template <>
struct fir_resampler<complex<float>, float> : runnable {
  float *freq_tap;
  float tap_multiplier;
  float freq_tol;
  fir_resampler(scheduler *sch, int ncoeffs, float *coeffs, pipebuf<complex<float> > &i, pipebuf<complex<float> > &o, int interp_ = 1,
                int decim_ = 1)
      : runnable(sch, "fir_resampler"), freq_tap(NULL), tap_multiplier(1), freq_tol(0.1),
        ctx(pipe_ctx(i, o, "fir_resampler: pipebufs of two device contexts")), n(ncoeffs), interp(interp_), in(i),
        out(o, interp_), current_freq(0) {
    if (decim_ != 1) fail("fir_resampler: decim not implemented");
    lsdr_check(lsdr_fir_resampler_create(ctx, ncoeffs, coeffs, interp_, &h), name);
  }
  void run() {
    if (in.readable() < n) return;
    if (freq_tap) {   // dsp.h:309-316
      float new_freq = *freq_tap * tap_multiplier;
      if (fabs(current_freq - new_freq) > freq_tol) {
        lsdr_check(lsdr_fir_resampler_set_freq(h, new_freq), name);
        current_freq = new_freq;
      }
    }
    unsigned long room = out.writable();
    size_t consumed = 0, produced = 0;
    lsdr_check(lsdr_fir_resampler_run(h, (const lsdr_cf32 *)in.rd(), in.readable(), (lsdr_cf32 *)out.wr(), room, &consumed, &produced), name);
    in.read(consumed);
    out.written(produced);
  }

 private:
  lsdr_ctx *ctx;
  unsigned n;
  int interp;
  dev_reader<complex<float> > in;
  dev_writer<complex<float> > out;
  lsdr_fir_resampler *h;
  float current_freq;
};

// ---- channel-simulator blocks of leanchansim (dsp.h:33-54, 118-138, 164-190) on device pipebufs.
template <>
struct cconverter<float, 0, u8, 128, 1, 1> : runnable {
  cconverter(scheduler *sch, pipebuf<complex<float> > &i, pipebuf<complex<u8> > &o)
      : runnable(sch, "cconverter"), ctx(pipe_ctx(i, o, "cconverter: pipebufs of two device contexts")),
        in(i), out(o) {}
  void run() {
    unsigned long count = min(in.readable(), out.writable());
    if (!count) return;
    lsdr_check(lsdr_cconverter_f32_u8_run(ctx, (const lsdr_cf32 *)in.rd(), count, (lsdr_cu8 *)out.wr()), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<complex<float> > in;
  dev_writer<complex<u8> > out;
};

template <>
struct cconverter<float, 0, int16_t, 0, 32768, 1> : runnable {   // leandvbtx --s16 (leandvbtx.cc:179)
  cconverter(scheduler *sch, pipebuf<complex<float> > &i, pipebuf<complex<int16_t> > &o)
      : runnable(sch, "cconverter"), ctx(pipe_ctx(i, o, "cconverter: pipebufs of two device contexts")),
        in(i), out(o) {}
  void run() {
    unsigned long count = min(in.readable(), out.writable());
    if (!count) return;
    lsdr_check(lsdr_cconverter_f32_s16_run(ctx, (const lsdr_cf32 *)in.rd(), count, (int16_t *)out.wr()), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<complex<float> > in;
  dev_writer<complex<int16_t> > out;
};

template <typename T>
struct adder;

template <>
struct adder<complex<float> > : runnable {
  adder(scheduler *sch, pipebuf<complex<float> > &i1, pipebuf<complex<float> > &i2, pipebuf<complex<float> > &o)
      : runnable(sch, "adder"), ctx(pipe_ctx(i1, o, "adder: pipebufs of two device contexts")), in1(i1), in2(i2),
        out(o) {
    pipe_ctx(i2, o, "adder: pipebufs of two device contexts");
  }
  void run() {
    unsigned long n = out.writable();
    if (in1.readable() < n) n = in1.readable();
    if (in2.readable() < n) n = in2.readable();
    if (!n) return;
    lsdr_check(lsdr_adder_run(ctx, (const lsdr_cf32 *)in1.rd(), (const lsdr_cf32 *)in2.rd(), n, (lsdr_cf32 *)out.wr()), name);
    in1.read(n);
    in2.read(n);
    out.written(n);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<complex<float> > in1, in2;
  dev_writer<complex<float> > out;
};

// wgn_c<float>: glibc's drand48 stream, continued on the device.  `seed()` stands for the srand48() call the reference's
// main() makes before sch.run() (leanchansim.cc:146-147); without it the stream is that of a process that never seeds.
template <typename T>
struct wgn_c;

template <>
struct wgn_c<float> : runnable {
  float stddev;
  wgn_c(scheduler *sch, pipebuf<complex<float> > &o)
      : runnable(sch, "awgn"), stddev(1.0), ctx(pipe_ctx(o, "wgn_c")), out(o), h(NULL), seeded(false), seedval(0) {}
  void seed(long s) { seeded = true; seedval = s; }
  void run() {
    if (!h) lsdr_check(lsdr_wgn_create(ctx, seeded, seedval, &h), name);
    unsigned long n = out.writable();
    if (!n) return;
    lsdr_check(lsdr_wgn_run(h, stddev, NULL, (lsdr_cf32 *)out.wr(), n), name);
    out.written(n);
  }

 private:
  lsdr_ctx *ctx;
  dev_writer<complex<float> > out;
  lsdr_wgn *h;
  bool seeded;
  long seedval;
};

}  // namespace leansdr
#endif

// leansdr_amd/host/apps/leanchansim_amd.cc — channel simulator: IQ on stdin → scaled, noisy, drifting IQ on stdout, every
// block on the GPU.  It is the graph of the reference's leanchansim (src/apps/leanchansim.cc:120-176) with the same options;
// with --deterministic and stdin from a file the output bytes are the reference's (tests/test_gpu_chan.py).
#include <math.h>
#include <unistd.h>

#include "cli.h"
#include "leansdr/dsp.h"
#include "leansdr/generic.h"

using namespace leansdr;

typedef float f32;
typedef complex<f32> cf32;
typedef complex<u8> cu8;

// LO drift: up to three sinusoidal FM components (amplitude and rate in cycles per sample) on a 16-bit phase accumulator.
// The reference's drifter (leanchansim.cc:34-88) keeps that accumulator in a local of run(), so its output depends on how
// its 4096-sample pipes cut the stream; `chunk` reproduces that cut whatever the size of the device pipes: only whole
// chunks are consumed, and a short remainder is flushed once a scheduler pass has brought nothing new (end of input).
template <typename T>
struct drifter;

template <>
struct drifter<float> : runnable {
  static const int NCOMPONENTS = 3;
  struct component {
    float amp, freq;
  } drifts[NCOMPONENTS];
  unsigned long chunk;

  drifter(scheduler *s, pipebuf<cf32> &src, pipebuf<cf32> &dst)
      : runnable(s, "drifter"), chunk(4096), ctx_(pipe_ctx(src, dst, "drifter: pipebufs must be device pipebufs of one ctx")),
        from_(src), to_(dst), handle_(NULL), seen_(0) {
    for (int i = 0; i < NCOMPONENTS; ++i) drifts[i].amp = drifts[i].freq = 0;
    lsdr_check(lsdr_drifter_create(ctx_, &handle_), name);
  }
  void run() {
    unsigned long n = min(from_.readable(), to_.writable());
    if (chunk && n >= chunk) n -= n % chunk;
    else if (chunk && n && from_.readable() != seen_) { seen_ = from_.readable(); return; }
    if (!n) return;
    seen_ = 0;
    for (int i = 0; i < NCOMPONENTS; ++i) lsdr_check(lsdr_drifter_set_component(handle_, i, drifts[i].amp, drifts[i].freq), name);
    lsdr_check(lsdr_drifter_run(handle_, (const lsdr_cf32 *)from_.rd(), n, (lsdr_cf32 *)to_.wr(), chunk), name);
    from_.read(n);
    to_.written(n);
  }

 private:
  lsdr_ctx *ctx_;
  dev_reader<cf32> from_;
  dev_writer<cf32> to_;
  lsdr_drifter *handle_;
  unsigned long seen_;
};

namespace {

struct settings {
  bool in_u8 = false, out_u8 = false, loop = false, deterministic = false;
  float gain = 1, noise = 0;                    // --scale, --awgn (dB → standard deviation)
  float rate = 0, lo = 0, ppm = -1;             // -f, --lo, --ppm
  float period = 0, slope = 0;                  // --drift-period, --drift-rate
  float amp2 = 0, freq2 = 0;                    // --drift2-amp, --drift2-freq
  int device = 0;
  unsigned long pipe = 1ul << 20;               // the reference: 4096 (a CPU cache); a GPU wants batches
};

template <typename T>
pipebuf<T> &from_stdin(cli::graph &g, const settings &o) {
  pipebuf<T> &host = g.host<T>("stdin", o.pipe);
  (new file_reader<T>(&g.sch, 0, host))->loop = o.loop;
  pipebuf<T> &dev = g.hbm<T>("stdin(dev)", o.pipe);
  new h2d_copier<T>(&g.sch, g.ctx, host, dev);
  return dev;
}
template <typename T>
void to_stdout(cli::graph &g, const settings &o, pipebuf<T> &dev) {
  pipebuf<T> &host = g.host<T>("stdout", o.pipe);
  new d2h_copier<T>(&g.sch, g.ctx, dev, host);
  new file_writer<T>(&g.sch, host, 1);
}

void simulate(const settings &o) {
  cli::graph g(o.device);
  pipebuf<cf32> *x;
  if (o.in_u8) {
    x = &g.hbm<cf32>("stdinf", o.pipe);
    new cconverter<u8, 128, f32, 0, 1, 1>(&g.sch, from_stdin<cu8>(g, o), *x);
  } else {
    x = &from_stdin<cf32>(g, o);
  }

  pipebuf<cf32> &scaled = g.hbm<cf32>("scaled", o.pipe);
  new scaler<float, cf32, cf32>(&g.sch, o.gain, *x, scaled);

  pipebuf<cf32> &noise = g.hbm<cf32>("noise", o.pipe);
  wgn_c<f32> *gen = new wgn_c<f32>(&g.sch, noise);
  gen->stddev = o.noise;
  if (!o.deterministic) gen->seed(getpid());      // the reference seeds drand48 with its pid (leanchansim.cc:146-147)
  pipebuf<cf32> &noisy = g.hbm<cf32>("noisy", o.pipe);
  new adder<cf32>(&g.sch, scaled, noise, noisy);

  pipebuf<cf32> &drifting = g.hbm<cf32>("drift", o.pipe);
  drifter<float> *lo = new drifter<float>(&g.sch, noisy, drifting);
  // component 0: ±ppm of the LO, swept with a period or a maximum rate; component 1: a secondary wobble.  The arithmetic
  // (float fields, double constants) is the reference's, leanchansim.cc:155-170 — without -f it yields NaN, i.e. no drift.
  const float max_offset = o.lo * o.ppm * 1e-6;
  lo->drifts[0].amp = max_offset / o.rate;
  if (o.period && o.slope) fail("Specify only one of --drift-rate and --drift-period");
  if (o.period) lo->drifts[0].freq = (1.0 / o.period) / o.rate;
  if (o.slope) {
    if (!o.ppm) fail("Need --ppm with --drift-rate");
    lo->drifts[0].freq = (o.slope / (2 * M_PI * o.ppm)) / o.rate;
  }
  if (o.amp2 && o.freq2) {
    lo->drifts[1].amp = o.amp2 / o.rate;
    lo->drifts[1].freq = o.freq2 / o.rate;
  }

  if (o.out_u8) {
    pipebuf<cu8> &bytes = g.hbm<cu8>("stdout(dev)", o.pipe);
    new cconverter<f32, 0, u8, 128, 1, 1>(&g.sch, drifting, bytes);
    to_stdout(g, o, bytes);
  } else {
    to_stdout(g, o, drifting);
  }
  g.run();
}

}  // namespace

int main(int argc, char **argv) {
  settings o;
  cli::parser p;
  p.summary = "Simulate an imperfect channel on the GPU: IQ.in on stdin, IQ.out on stdout (leanchansim on MI355X).";
  p.options = {
      {"--iu8", NULL, "stdin is complex unsigned char", [&](const char *) { o.in_u8 = true; }},
      {"--if32", NULL, "stdin is complex float (default)", [&](const char *) { o.in_u8 = false; }},
      {"-f", "HZ", "sample rate", [&](const char *a) { o.rate = atof(a); }},
      {"--loop", NULL, "repeat (stdin must be a file)", [&](const char *) { o.loop = true; }},
      {"--scale", "FACTOR", "multiply by a constant", [&](const char *a) { o.gain = atof(a); }},
      {"--awgn", "DB", "add white gaussian noise of this standard deviation", [&](const char *a) { o.noise = cli::db_to_amplitude(a); }},
      {"--deterministic", NULL, "do not seed the noise generator", [&](const char *) { o.deterministic = true; }},
      {"--lo", "HZ", "nominal LO frequency", [&](const char *a) { o.lo = atof(a); }},
      {"--ppm", "PPM", "LO accuracy", [&](const char *a) { o.ppm = atof(a); }},
      {"--drift-period", "S", "drift +-ppm every S seconds", [&](const char *a) { o.period = atof(a); }},
      {"--drift-rate", "R", "drift with maximum rate R (Hz/s)", [&](const char *a) { o.slope = atof(a); }},
      {"--drift2-amp", "HZ", "secondary drift, range", [&](const char *a) { o.amp2 = atof(a); }},
      {"--drift2-freq", "HZ", "secondary drift, rate", [&](const char *a) { o.freq2 = atof(a); }},
      {"--ou8", NULL, "stdout is complex unsigned char", [&](const char *) { o.out_u8 = true; }},
      {"--of32", NULL, "stdout is complex float (default)", [&](const char *) { o.out_u8 = false; }},
      {"--device", "N", "GPU index", [&](const char *a) { o.device = atoi(a); }},
      {"--buf", "SAMPLES", "pipe size (default 1 Mi)", [&](const char *a) { o.pipe = strtoul(a, NULL, 0); }},
  };
  p.parse(argc, argv);
  simulate(o);
  return 0;
}

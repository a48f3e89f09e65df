// leansdr_amd/host/apps/leanchansim_amd.cc — the graph of leanchansim (src/apps/leanchansim.cc:120-176 of the reference) built
// against the MI355X host framework: IQ on stdin → scaled, noisy, drifting IQ on stdout, every block on the GPU.
// Same options: --iu8 | --if32, --ou8 | --of32, -f HZ, --loop, --scale K, --awgn DB, --deterministic,
// --lo HZ, --ppm PPM, --drift-period S, --drift-rate R, --drift2-amp HZ, --drift2-freq HZ.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "leansdr/dsp.h"
#include "leansdr/framework.h"
#include "leansdr/generic.h"

using namespace leansdr;

typedef float f32;
typedef complex<f32> cf32;
typedef unsigned char u8;
typedef complex<u8> cu8;

// drifter<float> (leanchansim.cc:34-88).  The reference restarts its phase accumulator on every run(), i.e. every time its
// 4096-sample pipes hand it a chunk; `chunk` keeps that cut whatever the size of the device pipes.
template <typename T>
struct drifter;

template <>
struct drifter<float> : runnable {
  static const int NCOMPONENTS = 3;
  struct component {
    float amp;
    float freq;
  } drifts[NCOMPONENTS];
  unsigned long chunk;
  drifter(scheduler *sch, pipebuf<cf32> &i, pipebuf<cf32> &o)
      : runnable(sch, "drifter"), chunk(4096), ctx(pipe_ctx(i.dev, o.dev, "drifter: pipebufs must be device pipebufs of one ctx")), in(i),
        out(o), h(NULL), stalled(0) {
    memset(drifts, 0, sizeof(drifts));
    lsdr_check(lsdr_drifter_create(ctx, &h), name);
  }
  void run() {
    unsigned long count = min(in.readable(), out.writable());
    if (!count) return;
    if (chunk && count >= chunk) count -= count % chunk;
    else if (chunk && in.readable() != stalled) {   // a partial chunk: wait one scheduler pass for more input (only EOF leaves it)
      stalled = in.readable();
      return;
    }
    stalled = 0;
    for (int i = 0; i < NCOMPONENTS; ++i) lsdr_check(lsdr_drifter_set_component(h, i, drifts[i].amp, drifts[i].freq), name);
    lsdr_check(lsdr_drifter_run(h, (const lsdr_cf32 *)in.rd(), count, (lsdr_cf32 *)out.wr(), chunk), name);
    in.read(count);
    out.written(count);
  }

 private:
  lsdr_ctx *ctx;
  pipereader<cf32> in;
  pipewriter<cf32> out;
  lsdr_drifter *h;
  unsigned long stalled;
};

struct config {
  bool loop_input;
  enum { IO_F32, IO_U8 } input_format, output_format;
  float scale, awgn;
  bool deterministic;
  float Fs, Flo, ppm, drift_period, drift_rate, drift2_amp, drift2_freq;
  int device;
  unsigned long buf;
  config()
      : loop_input(false), input_format(IO_F32), output_format(IO_F32), scale(1), awgn(0), deterministic(false), Fs(0), Flo(0), ppm(-1),
        drift_period(0), drift_rate(0), drift2_amp(0), drift2_freq(0), device(0), buf(1 << 20) {}
};

static int run(config &cfg) {
  scheduler sch;
  lsdr_ctx *ctx = NULL;
  lsdr_check(lsdr_ctx_create(cfg.device, NULL, &ctx), "lsdr_ctx_create");
  const unsigned long BUF_BASEBAND = cfg.buf;   // the reference: 4096 (a CPU cache); a GPU wants batches

  pipebuf<cf32> *pipe = NULL;
  if (cfg.input_format == config::IO_F32) {
    pipebuf<cf32> *p_stdin = new pipebuf<cf32>(&sch, "stdin", BUF_BASEBAND);
    file_reader<cf32> *r_stdin = new file_reader<cf32>(&sch, 0, *p_stdin);
    r_stdin->loop = cfg.loop_input;
    pipebuf<cf32> *p_dev = new pipebuf<cf32>(&sch, "stdin(dev)", BUF_BASEBAND, ctx);
    new h2d_copier<cf32>(&sch, ctx, *p_stdin, *p_dev);
    pipe = p_dev;
  } else {
    pipebuf<cu8> *p_stdin = new pipebuf<cu8>(&sch, "stdin", BUF_BASEBAND);
    file_reader<cu8> *r_stdin = new file_reader<cu8>(&sch, 0, *p_stdin);
    r_stdin->loop = cfg.loop_input;
    pipebuf<cu8> *p_dev = new pipebuf<cu8>(&sch, "stdin(dev)", BUF_BASEBAND, ctx);
    new h2d_copier<cu8>(&sch, ctx, *p_stdin, *p_dev);
    pipebuf<cf32> *p_stdinf = new pipebuf<cf32>(&sch, "stdinf", BUF_BASEBAND, ctx);
    new cconverter<u8, 128, f32, 0, 1, 1>(&sch, *p_dev, *p_stdinf);
    pipe = p_stdinf;
  }

  pipebuf<cf32> p_scaled(&sch, "scaled", BUF_BASEBAND, ctx);
  scaler<float, cf32, cf32> r_scale(&sch, cfg.scale, *pipe, p_scaled);
  pipe = &p_scaled;

  pipebuf<cf32> p_noise(&sch, "noise", BUF_BASEBAND, ctx);
  wgn_c<f32> r_noise(&sch, p_noise);
  if (!cfg.deterministic) r_noise.seed(getpid());   // leanchansim.cc:146-147
  r_noise.stddev = cfg.awgn;
  pipebuf<cf32> p_noisy(&sch, "noisy", BUF_BASEBAND, ctx);
  adder<cf32> r_addnoise(&sch, *pipe, p_noise, p_noisy);
  pipe = &p_noisy;

  pipebuf<cf32> p_drift(&sch, "drift", BUF_BASEBAND, ctx);
  drifter<float> r_drift(&sch, *pipe, p_drift);
  float maxoffs = cfg.Flo * cfg.ppm * 1e-6;
  r_drift.drifts[0].amp = maxoffs / cfg.Fs;
  if (cfg.drift_period && cfg.drift_rate) fail("Specify only one of --drift-rate and --drift-period");
  if (cfg.drift_period) r_drift.drifts[0].freq = (1.0 / cfg.drift_period) / cfg.Fs;
  if (cfg.drift_rate) {
    if (!cfg.ppm) fail("Need --ppm with --drift-rate");
    r_drift.drifts[0].freq = (cfg.drift_rate / (2 * M_PI * cfg.ppm)) / cfg.Fs;
  }
  if (cfg.drift2_amp && cfg.drift2_freq) {
    r_drift.drifts[1].amp = cfg.drift2_amp / cfg.Fs;
    r_drift.drifts[1].freq = cfg.drift2_freq / cfg.Fs;
  }
  pipe = &p_drift;

  if (cfg.output_format == config::IO_U8) {
    pipebuf<cu8> *p_out = new pipebuf<cu8>(&sch, "stdout(dev)", BUF_BASEBAND, ctx);
    new cconverter<f32, 0, u8, 128, 1, 1>(&sch, *pipe, *p_out);
    pipebuf<cu8> *p_stdout = new pipebuf<cu8>(&sch, "stdout", BUF_BASEBAND);
    new d2h_copier<cu8>(&sch, ctx, *p_out, *p_stdout);
    new file_writer<cu8>(&sch, *p_stdout, 1);
  } else {
    pipebuf<cf32> *p_stdout = new pipebuf<cf32>(&sch, "stdout", BUF_BASEBAND);
    new d2h_copier<cf32>(&sch, ctx, *pipe, *p_stdout);
    new file_writer<cf32>(&sch, *p_stdout, 1);
  }

  sch.run();
  sch.shutdown();
  return 0;
}

static void usage(const char *name, FILE *f, int c) {
  fprintf(f, "Usage: %s [options]  < IQ.in  > IQ.out\n", name);
  fprintf(f, "Simulate an imperfect communication channel on the GPU (leanchansim on MI355X).\n"
             "  --iu8 | --if32 | -f HZ | --loop | --scale K | --awgn DB | --deterministic\n"
             "  --lo HZ | --ppm PPM | --drift-period S | --drift-rate R | --drift2-amp HZ | --drift2-freq HZ\n"
             "  --ou8 | --of32 | --device N | --buf SAMPLES\n");
  exit(c);
}

int main(int argc, char *argv[]) {
  config cfg;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-h")) usage(argv[0], stdout, 0);
    else if (!strcmp(argv[i], "--iu8")) cfg.input_format = config::IO_U8;
    else if (!strcmp(argv[i], "--if32")) cfg.input_format = config::IO_F32;
    else if (!strcmp(argv[i], "--loop")) cfg.loop_input = true;
    else if (!strcmp(argv[i], "--ou8")) cfg.output_format = config::IO_U8;
    else if (!strcmp(argv[i], "--of32")) cfg.output_format = config::IO_F32;
    else if (!strcmp(argv[i], "-f") && i + 1 < argc) cfg.Fs = atof(argv[++i]);
    else if (!strcmp(argv[i], "--scale") && i + 1 < argc) cfg.scale = atof(argv[++i]);
    else if (!strcmp(argv[i], "--awgn") && i + 1 < argc) cfg.awgn = expf(logf(10) * atof(argv[++i]) / 20);
    else if (!strcmp(argv[i], "--deterministic")) cfg.deterministic = true;
    else if (!strcmp(argv[i], "--lo") && i + 1 < argc) cfg.Flo = atof(argv[++i]);
    else if (!strcmp(argv[i], "--ppm") && i + 1 < argc) cfg.ppm = atof(argv[++i]);
    else if (!strcmp(argv[i], "--drift-period") && i + 1 < argc) cfg.drift_period = atof(argv[++i]);
    else if (!strcmp(argv[i], "--drift-rate") && i + 1 < argc) cfg.drift_rate = atof(argv[++i]);
    else if (!strcmp(argv[i], "--drift2-amp") && i + 1 < argc) cfg.drift2_amp = atof(argv[++i]);
    else if (!strcmp(argv[i], "--drift2-freq") && i + 1 < argc) cfg.drift2_freq = atof(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) cfg.device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--buf") && i + 1 < argc) cfg.buf = strtoul(argv[++i], NULL, 0);
    else usage(argv[0], stderr, 1);
  }
  return run(cfg);
}

// leansdr_amd/host/apps/leandvbtx_amd.cc — the graph of leandvbtx (src/apps/leandvbtx.cc:79-197 of the reference) built
// against the MI355X host framework: TS packets on stdin → cf32 (or s16) baseband on stdout, every block on the GPU.
// Same options: --cr N/D, --const NAME, -f INTERP[/DECIM], --roll-off R, --rrc-rej, --power DB, --agc, --f32 | --s16, --fill, -v, -d.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "leansdr/dsp.h"
#include "leansdr/dvb.h"
#include "leansdr/filtergen.h"
#include "leansdr/framework.h"
#include "leansdr/generic.h"
#include "leansdr/sdr.h"

using namespace leansdr;

struct config {
  cstln_lut<256>::predef constellation;
  code_rate fec;
  float amp;
  bool agc;
  int interp, decim;
  float rolloff, rrc_rej;
  enum { OUTPUT_F32, OUTPUT_S16 } output_format;
  bool fill;
  bool verbose, debug;
  int device;
  config() : constellation(cstln_lut<256>::QPSK), fec(FEC12), amp(1.0), agc(false), interp(2), decim(1), rolloff(0.35), rrc_rej(10),
             output_format(OUTPUT_F32), fill(false), verbose(false), debug(false), device(0) {}
};

static int log2i(int x) { int n = -1; for (; x; ++n, x >>= 1); return n; }

static void run(config &cfg) {
  scheduler sch;
  sch.verbose = cfg.verbose;
  sch.debug = cfg.debug;
  lsdr_ctx *ctx = NULL;
  lsdr_check(lsdr_ctx_create(cfg.device, NULL, &ctx), "lsdr_ctx_create");
  // The reference sizes its pipes for a CPU (12·2 packets); a GPU wants batches.
  const unsigned long bf = 512;
  unsigned long BUF_PACKETS = 12 * bf, BUF_BYTES = SIZE_RSPACKET * BUF_PACKETS, BUF_SYMBOLS = BUF_BYTES * 8 * 2;

  pipebuf<tspacket> p_stdin(&sch, "TS packets(host)", BUF_PACKETS);
  file_reader<tspacket> r_stdin(&sch, 0, p_stdin);
  pipebuf<tspacket> p_tspackets(&sch, "TS packets", BUF_PACKETS, ctx);
  h2d_copier<tspacket> r_h2d(&sch, ctx, p_stdin, p_tspackets);
  pipebuf<tspacket> p_rtspackets(&sch, "rand TS packets", BUF_PACKETS, ctx);
  randomizer r_rand(&sch, p_tspackets, p_rtspackets);
  pipebuf<rspacket<u8> > p_rspackets(&sch, "RS-enc packets", BUF_PACKETS, ctx);
  rs_encoder r_rsenc(&sch, p_rtspackets, p_rspackets);
  pipebuf<u8> p_mpegbytes(&sch, "mpegbytes", BUF_BYTES, ctx);
  interleaver r_inter(&sch, p_rspackets, p_mpegbytes);

  cstln_lut<256> *cstln = make_dvbs2_constellation(cfg.constellation, cfg.fec);
  int bits_per_symbol = log2i(cstln->nsymbols);
  if (cfg.fec == FEC23 && (cstln->nsymbols == 4 || cstln->nsymbols == 64)) cfg.fec = FEC46;   // leandvbtx.cc:117-121
  pipebuf<u8> p_symbols(&sch, "symbols", BUF_SYMBOLS, ctx);
  dvb_convol r_convol(&sch, p_mpegbytes, p_symbols, cfg.fec, bits_per_symbol);
  pipebuf<cf32> p_iqsymbols(&sch, "IQ symbols", BUF_SYMBOLS, ctx);
  cstln_transmitter<f32, 0> r_mod(&sch, p_symbols, p_iqsymbols);
  r_mod.cstln = cstln;

  float Fm = 1.0 / cfg.interp;
  int order = cfg.interp * cfg.rrc_rej;
  float *coeffs;
  int ncoeffs = filtergen::root_raised_cosine(order, Fm, cfg.rolloff, &coeffs);
  filtergen::normalize_power(ncoeffs, coeffs, cfg.amp / cstln_amp);
  if (sch.verbose) fprintf(stderr, "Interpolation: ratio %d/%d, rolloff %f, %d coeffs\n", cfg.interp, cfg.decim, cfg.rolloff, ncoeffs);
  pipebuf<cf32> p_interp(&sch, "interpolated", BUF_SYMBOLS * cfg.interp, ctx);
  fir_resampler<cf32, float> r_resampler(&sch, ncoeffs, coeffs, p_iqsymbols, p_interp, cfg.interp, 1);
  pipebuf<cf32> p_resampled(&sch, "resampled", BUF_SYMBOLS * cfg.interp, ctx);
  decimator<cf32> r_decim(&sch, cfg.decim, p_interp, p_resampled);
  pipebuf<cf32> *tail = &p_resampled;
  if (cfg.agc) {
    pipebuf<cf32> *p_agc = new pipebuf<cf32>(&sch, "AGC", BUF_SYMBOLS * cfg.interp, ctx);
    simple_agc<f32> *r_agc = new simple_agc<f32>(&sch, *tail, *p_agc);
    r_agc->out_rms = cfg.amp / sqrtf((float)cfg.interp / cfg.decim);
    r_agc->bw = 0.001 * cfg.decim / cfg.interp;
    tail = p_agc;
  }
  if (cfg.output_format == config::OUTPUT_F32) {
    pipebuf<cf32> *p_host = new pipebuf<cf32>(&sch, "baseband(host)", BUF_SYMBOLS * cfg.interp);
    new d2h_copier<cf32>(&sch, ctx, *tail, *p_host);
    new file_writer<cf32>(&sch, *p_host, 1);
  } else {   // leandvbtx.cc:176-182
    typedef complex<int16_t> cs16;
    pipebuf<cs16> *p_s16 = new pipebuf<cs16>(&sch, "stdout(dev)", BUF_SYMBOLS * cfg.interp, ctx);
    new cconverter<f32, 0, int16_t, 0, 32768, 1>(&sch, *tail, *p_s16);
    pipebuf<cs16> *p_host = new pipebuf<cs16>(&sch, "stdout", BUF_SYMBOLS * cfg.interp);
    new d2h_copier<cs16>(&sch, ctx, *p_s16, *p_host);
    new file_writer<cs16>(&sch, *p_host, 1);
  }
  if (cfg.fill) {   // leandvbtx.cc:187-193
    if (cfg.verbose) fprintf(stderr, "Realtime mode\n");
    tspacket blank;
    memset(blank.data, 0, 188);
    blank.data[0] = 0x47;
    r_stdin.set_realtime(blank);
  }

  sch.run();
  sch.shutdown();
  if (sch.verbose) sch.dump();
  lsdr_ctx_destroy(ctx);
}

static void usage(const char *name, FILE *f, int c) {
  fprintf(f, "Usage: %s [options]  < TS  > IQ\n", name);
  fprintf(f, "Modulate MPEG packets into a DVB-S baseband signal on the GPU (leandvbtx on MI355X)\n"
             "  --cr N/D | --const NAME | -f INTERP[/DECIM] | --roll-off R | --rrc-rej N | --power DB | --agc\n"
             "  --f32 | --s16 | --fill | --device N | -v | -d\n");
  exit(c);
}

int main(int argc, char *argv[]) {
  config cfg;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-h")) usage(argv[0], stdout, 0);
    else if (!strcmp(argv[i], "-v")) cfg.verbose = true;
    else if (!strcmp(argv[i], "-d")) cfg.debug = true;
    else if (!strcmp(argv[i], "--cr") && i + 1 < argc) {
      ++i;
      if (!strcmp(argv[i], "1/2")) cfg.fec = FEC12;
      else if (!strcmp(argv[i], "2/3")) cfg.fec = FEC23;
      else if (!strcmp(argv[i], "3/4")) cfg.fec = FEC34;
      else if (!strcmp(argv[i], "5/6")) cfg.fec = FEC56;
      else if (!strcmp(argv[i], "7/8")) cfg.fec = FEC78;
      else if (!strcmp(argv[i], "4/5")) cfg.fec = FEC45;
      else if (!strcmp(argv[i], "8/9")) cfg.fec = FEC89;
      else if (!strcmp(argv[i], "9/10")) cfg.fec = FEC910;
      else usage(argv[0], stderr, 1);
    } else if (!strcmp(argv[i], "--const") && i + 1 < argc) {
      ++i;
      static const struct { const char *n; cstln_lut<256>::predef v; } tab[] = {
          {"BPSK", cstln_lut<256>::BPSK}, {"QPSK", cstln_lut<256>::QPSK}, {"8PSK", cstln_lut<256>::PSK8},
          {"16APSK", cstln_lut<256>::APSK16}, {"32APSK", cstln_lut<256>::APSK32}, {"64APSKe", cstln_lut<256>::APSK64E},
          {"16QAM", cstln_lut<256>::QAM16}, {"64QAM", cstln_lut<256>::QAM64}, {"256QAM", cstln_lut<256>::QAM256}};
      bool ok = false;
      for (auto &t : tab) if (!strcmp(argv[i], t.n)) { cfg.constellation = t.v; ok = true; }
      if (!ok) usage(argv[0], stderr, 1);
    } else if (!strcmp(argv[i], "-f") && i + 1 < argc) {
      ++i;
      cfg.decim = 1;
      if (sscanf(argv[i], "%d/%d", &cfg.interp, &cfg.decim) < 1) usage(argv[0], stderr, 1);
    } else if (!strcmp(argv[i], "--roll-off") && i + 1 < argc) cfg.rolloff = atof(argv[++i]);
    else if (!strcmp(argv[i], "--rrc-rej") && i + 1 < argc) cfg.rrc_rej = atof(argv[++i]);
    else if (!strcmp(argv[i], "--power") && i + 1 < argc) cfg.amp = expf(logf(10) * atof(argv[++i]) / 20);
    else if (!strcmp(argv[i], "--agc")) cfg.agc = true;
    else if (!strcmp(argv[i], "--f32")) cfg.output_format = config::OUTPUT_F32;
    else if (!strcmp(argv[i], "--s16")) cfg.output_format = config::OUTPUT_S16;
    else if (!strcmp(argv[i], "--fill")) cfg.fill = true;
    else if (!strcmp(argv[i], "--version")) { printf("leansdr_amd\n"); exit(0); }
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) cfg.device = atoi(argv[++i]);
    else usage(argv[0], stderr, 1);
  }
  run(cfg);
  return 0;
}

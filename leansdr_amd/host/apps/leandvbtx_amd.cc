// leansdr_amd/host/apps/leandvbtx_amd.cc — DVB-S modulator: MPEG-TS packets on stdin → baseband IQ on stdout, every block
// on the GPU.  It is the graph of the reference's leandvbtx (src/apps/leandvbtx.cc:79-197) with the same options and, for
// the same input, the same output bytes (tests/test_gpu_tx.py); pipes are sized for GPU batches instead of a CPU cache.
#include <math.h>

#include "cli.h"
#include "leansdr/dsp.h"
#include "leansdr/dvb.h"
#include "leansdr/filtergen.h"
#include "leansdr/generic.h"
#include "leansdr/sdr.h"

using namespace leansdr;

namespace {

struct settings {
  cstln_lut<256>::predef constellation = cstln_lut<256>::QPSK;
  code_rate rate = FEC12;
  float amplitude = 1.0f;     // RMS constellation amplitude (--power, dB)
  bool agc = false;
  int up = 2, down = 1;       // -f UP[/DOWN]
  float rolloff = 0.35f, rrc_rej = 10.0f;
  bool s16 = false, realtime = false, verbose = false, debug = false;
  int device = 0;
};

void modulate(const settings &o) {
  cli::graph g(o.device);
  g.sch.verbose = o.verbose;
  g.sch.debug = o.debug;
  const unsigned long packets = 12 * 512, bytes = packets * SIZE_RSPACKET, symbols = bytes * 16;   // worst case BPSK 1/2
  const unsigned long samples = symbols * o.up;

  // stdin → HBM
  pipebuf<tspacket> &ts_host = g.host<tspacket>("TS packets(host)", packets);
  file_reader<tspacket> *source = new file_reader<tspacket>(&g.sch, 0, ts_host);
  pipebuf<tspacket> &ts = g.hbm<tspacket>("TS packets", packets);
  new h2d_copier<tspacket>(&g.sch, g.ctx, ts_host, ts);

  // outer code
  pipebuf<tspacket> &scrambled = g.hbm<tspacket>("rand TS packets", packets);
  new randomizer(&g.sch, ts, scrambled);
  pipebuf<rspacket<u8> > &coded = g.hbm<rspacket<u8> >("RS-enc packets", packets);
  new rs_encoder(&g.sch, scrambled, coded);
  pipebuf<u8> &interleaved = g.hbm<u8>("mpegbytes", bytes);
  new interleaver(&g.sch, coded, interleaved);

  // inner code and mapping.  The constellation is built for the rate the user asked for; rate 2/3 on a constellation
  // whose label width does not divide 3 coded bits runs on the equivalent 4/6 code (leandvbtx.cc:117-121).
  cstln_lut<256> *points = make_dvbs2_constellation(o.constellation, o.rate);
  const int label_bits = log2i(points->nsymbols);
  const code_rate inner = (o.rate == FEC23 && (points->nsymbols == 4 || points->nsymbols == 64)) ? FEC46 : o.rate;
  pipebuf<u8> &labels = g.hbm<u8>("symbols", symbols);
  new dvb_convol(&g.sch, interleaved, labels, inner, label_bits);
  pipebuf<cf32> &mapped = g.hbm<cf32>("IQ symbols", symbols);
  (new cstln_transmitter<f32, 0>(&g.sch, labels, mapped))->cstln = points;

  // pulse shaping: root raised cosine at UP samples per symbol, scaled so that the output RMS is about `amplitude`
  float *taps = NULL;
  const int ntaps = filtergen::root_raised_cosine((int)(o.up * o.rrc_rej), (float)(1.0 / o.up), o.rolloff, &taps);
  filtergen::normalize_power(ntaps, taps, o.amplitude / cstln_amp);
  if (o.verbose) fprintf(stderr, "Interpolation: ratio %d/%d, rolloff %f, %d coeffs\n", o.up, o.down, o.rolloff, ntaps);
  pipebuf<cf32> &shaped = g.hbm<cf32>("interpolated", samples);
  new fir_resampler<cf32, float>(&g.sch, ntaps, taps, mapped, shaped, o.up, 1);
  pipebuf<cf32> *baseband = &g.hbm<cf32>("resampled", samples);
  new decimator<cf32>(&g.sch, o.down, shaped, *baseband);
  if (o.agc) {
    pipebuf<cf32> &levelled = g.hbm<cf32>("AGC", samples);
    simple_agc<f32> *agc = new simple_agc<f32>(&g.sch, *baseband, levelled);
    agc->out_rms = o.amplitude / sqrtf((float)o.up / o.down);
    agc->bw = 0.001 * o.down / o.up;            // slower loop for large interpolation ratios
    baseband = &levelled;
  }

  // HBM → stdout
  if (o.s16) {
    typedef complex<int16_t> cs16;
    pipebuf<cs16> &fixed = g.hbm<cs16>("stdout(dev)", samples);
    new cconverter<f32, 0, int16_t, 0, 32768, 1>(&g.sch, *baseband, fixed);
    pipebuf<cs16> &out = g.host<cs16>("stdout", samples);
    new d2h_copier<cs16>(&g.sch, g.ctx, fixed, out);
    new file_writer<cs16>(&g.sch, out, 1);
  } else {
    pipebuf<cf32> &out = g.host<cf32>("baseband(host)", samples);
    new d2h_copier<cf32>(&g.sch, g.ctx, *baseband, out);
    new file_writer<cf32>(&g.sch, out, 1);
  }
  if (o.realtime) {                               // --fill: null packets while stdin has nothing
    if (o.verbose) fprintf(stderr, "Realtime mode\n");
    tspacket idle;
    memset(idle.data, 0, sizeof(idle.data));
    idle.data[0] = 0x47;
    source->set_realtime(idle);
  }
  g.run();
}

}  // namespace

int main(int argc, char **argv) {
  settings o;
  static const cli::named<code_rate> rates[] = {{"1/2", FEC12}, {"2/3", FEC23}, {"3/4", FEC34}, {"5/6", FEC56}, {"7/8", FEC78},
                                                {"4/5", FEC45}, {"8/9", FEC89}, {"9/10", FEC910}};
  static const cli::named<cstln_lut<256>::predef> constellations[] = {
      {"BPSK", cstln_lut<256>::BPSK},       {"QPSK", cstln_lut<256>::QPSK},       {"8PSK", cstln_lut<256>::PSK8},
      {"16APSK", cstln_lut<256>::APSK16},   {"32APSK", cstln_lut<256>::APSK32},   {"64APSKe", cstln_lut<256>::APSK64E},
      {"16QAM", cstln_lut<256>::QAM16},     {"64QAM", cstln_lut<256>::QAM64},     {"256QAM", cstln_lut<256>::QAM256}};
  cli::parser p;
  p.summary = "Modulate MPEG packets from stdin into a DVB-S baseband signal on stdout (leandvbtx on MI355X).";
  auto bad = [&]() { p.usage(argv[0], stderr, 1); };
  p.options = {
      {"--cr", "N/D", "code rate (default 1/2)", [&](const char *a) { if (!cli::pick(a, rates, &o.rate)) bad(); }},
      {"--const", "NAME", "constellation: BPSK QPSK 8PSK 16APSK 32APSK 64APSKe 16QAM 64QAM 256QAM",
       [&](const char *a) { if (!cli::pick(a, constellations, &o.constellation)) bad(); }},
      {"-f", "UP[/DOWN]", "samples per symbol (default 2)",
       [&](const char *a) { o.down = 1; if (sscanf(a, "%d/%d", &o.up, &o.down) < 1) bad(); }},
      {"--roll-off", "R", "RRC roll-off (default 0.35)", [&](const char *a) { o.rolloff = atof(a); }},
      {"--rrc-rej", "N", "RRC filter length in symbols (default 10)", [&](const char *a) { o.rrc_rej = atof(a); }},
      {"--power", "DB", "output power (default 0 dB)", [&](const char *a) { o.amplitude = cli::db_to_amplitude(a); }},
      {"--agc", NULL, "regulate the output power", [&](const char *) { o.agc = true; }},
      {"--f32", NULL, "output complex float (default)", [&](const char *) { o.s16 = false; }},
      {"--s16", NULL, "output complex int16", [&](const char *) { o.s16 = true; }},
      {"--fill", NULL, "insert null packets when stdin starves", [&](const char *) { o.realtime = true; }},
      {"--device", "N", "GPU index", [&](const char *a) { o.device = atoi(a); }},
      {"-v", NULL, "verbose", [&](const char *) { o.verbose = true; }},
      {"-d", NULL, "debug", [&](const char *) { o.debug = true; }},
      {"--version", NULL, "print the version", [&](const char *) { printf("%s\n", VERSION); exit(0); }},
  };
  p.parse(argc, argv);
  modulate(o);
  return 0;
}

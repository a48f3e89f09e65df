// leansdr_amd/host/leansdr/dvb.h — DVB-S FEC blocks with the reference's class surface
// (dvb.h:122-513 deconvol_sync / make_deconvol_sync_simple, :712-891 mpeg_sync, :926-948
// deinterleaver, :985-1058 rs_decoder, :1107-1163 derandomizer) whose run() is one C-ABI call.
// All data pipebufs are device pipebufs; the small int/float report pipes stay on the host.
#ifndef LEANSDR_AMD_DVB_H
#define LEANSDR_AMD_DVB_H

#include "leansdr/framework.h"
#include "leansdr/sdr.h"

namespace leansdr {

static const int SIZE_RSPACKET = 204;
static const int SIZE_TSPACKET = 188;
static const int MPEG_SYNC = 0x47;
static const int MPEG_SYNC_INV = (MPEG_SYNC ^ 0xff);
static const int MPEG_SYNC_CORRUPTED = 0x55;

enum code_rate { FEC12, FEC23, FEC46, FEC34, FEC56, FEC78, FEC45, FEC89, FEC910, FEC_MAX };

inline cstln_lut<256> *make_dvbs2_constellation(cstln_lut<256>::predef c, code_rate r) {
  return new cstln_lut<256>(c, (int)r);   // radii per code rate are applied by the table builder (dvb.h:45-81)
}

// fec_specs[code_rate] (dvb.h:553-565): parameters of the convolutional coder, filled from the C ABI's table
// (leandvb prints bits_in/bits_out as the "CR" line of --fd-info).  Rates the coder does not have read 0/0, as in the reference.
struct fec_spec {
  int bits_in, bits_out;
  const uint16_t *polys;
};
namespace detail {
struct fec_spec_table {
  fec_spec spec[FEC_MAX];
  uint16_t polys[FEC_MAX][8];
  fec_spec_table() {
    for (int r = 0; r < FEC_MAX; ++r) {
      spec[r].bits_in = spec[r].bits_out = 0;
      spec[r].polys = polys[r];
      memset(polys[r], 0, sizeof(polys[r]));
      (void)lsdr_fec_spec(r, &spec[r].bits_in, &spec[r].bits_out, polys[r]);
    }
  }
};
inline fec_spec *fec_spec_array() {
  static fec_spec_table t;
  return t.spec;
}
}  // namespace detail
static fec_spec *const fec_specs = detail::fec_spec_array();

template <typename Tbyte>
struct rspacket { Tbyte data[SIZE_RSPACKET]; };
struct tspacket { u8 data[SIZE_TSPACKET]; };

template <typename Tbyte, Tbyte BYTE_ERASED>
struct deconvol_sync;

template <>
struct deconvol_sync<u8, 0> : runnable {
  bool fastlock;
  deconvol_sync(scheduler *sch, pipebuf<softsymbol> &i, pipebuf<u8> &o, code_rate rate)
      : runnable(sch, "deconvol_sync"), fastlock(false),
        ctx(pipe_ctx(i, o, "deconvol_sync: pipebufs of two device contexts")), in(i),
        out(o, SIZE_RSPACKET), rate_(rate), h(NULL) {}
  void run() {
    if (!h) lsdr_check(lsdr_deconv_create(ctx, (int)rate_, fastlock, &h), name);
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_deconv_run(h, (const lsdr_softsymbol *)in.rd(), in.readable(), out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }
  void next_sync() {
    if (fastlock) fail("Bug: next_sync() called with fastlock");
    if (!h) lsdr_check(lsdr_deconv_create(ctx, (int)rate_, fastlock, &h), name);
    lsdr_check(lsdr_deconv_next_sync(h), name);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<softsymbol> in;
  dev_writer<u8> out;
  code_rate rate_;
  lsdr_deconv *h;
};
typedef deconvol_sync<u8, 0> deconvol_sync_simple;

inline deconvol_sync_simple *make_deconvol_sync_simple(scheduler *sch, pipebuf<softsymbol> &in, pipebuf<u8> &out,
                                                       enum code_rate rate) {
  return new deconvol_sync_simple(sch, in, out, rate);
}

// viterbi_sync (dvb.h:1173-1416): same constructor and the public resync_period.
struct viterbi_sync : runnable {
  int resync_period;
  viterbi_sync(scheduler *sch, pipebuf<softsymbol> &i, pipebuf<unsigned char> &o, cstln_lut<256> *cstln, code_rate cr)
      : runnable(sch, "viterbi_sync"), resync_period(32),
        ctx(pipe_ctx(i, o, "viterbi_sync: pipebufs of two device contexts")), in(i), out(o, 128), h(NULL) {
    lsdr_check(lsdr_viterbi_create(ctx, (int)cstln->type, (int)cr, &h), name);
  }
  void run() {
    lsdr_check(lsdr_viterbi_set_resync_period(h, resync_period), name);
    // The reference's run() decodes chunks while input and room last (dvb.h:1371-1412).  lsdr_viterbi_run may stop earlier
    // (after an alignment switch it only looks one resync period ahead), so it is called until it makes no progress: one
    // scheduler pass then moves the same amount of data as the reference's, which is what the cadence of the --fd-info
    // reports downstream depends on.
    for (;;) {
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_viterbi_run(h, (const lsdr_softsymbol *)in.rd(), in.readable(), out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<softsymbol> in;
  dev_writer<unsigned char> out;
  lsdr_viterbi *h;
};

template <typename Tbyte, Tbyte BYTE_ERASED>
struct mpeg_sync;

// dvb_deconvol_sync<u8> (dvb.h:612-707), the --hs deconvolver: QPSK 1/2 only, hard symbols in.
template <typename Tin>
struct dvb_deconvol_sync;

template <>
struct dvb_deconvol_sync<u8> : runnable {
  typedef u8 decoded_byte;
  int resync_period;
  static const int chunk_size = 64;
  dvb_deconvol_sync(scheduler *sch, pipebuf<u8> &i, pipebuf<decoded_byte> &o)
      : runnable(sch, "deconvol_sync_multipoly"), resync_period(32),
        ctx(pipe_ctx(i, o, "dvb_deconvol_sync: pipebufs of two device contexts")), in(i), out(o, chunk_size),
        h(NULL) {}
  void run() {
    if (!h) lsdr_check(lsdr_hsdeconv_create(ctx, resync_period, &h), name);
    for (;;) {   // until no progress, like the while loop of dvb.h:634-637
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_hsdeconv_run(h, in.rd(), in.readable(), out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<u8> in;
  dev_writer<decoded_byte> out;
  lsdr_hsdeconv *h;
};
typedef dvb_deconvol_sync<u8> dvb_deconvol_sync_hard;

template <>
struct mpeg_sync<u8, 0> : runnable {
  int scan_syncs, want_syncs;
  unsigned long lock_timeout;
  bool fastlock;
  int resync_period;

  mpeg_sync(scheduler *sch, pipebuf<u8> &i, pipebuf<u8> &o, deconvol_sync<u8, 0> *dc, pipebuf<int> *state_o = NULL,
            pipebuf<unsigned long> *locktime_o = NULL)
      : runnable(sch, "sync_detect"), scan_syncs(8), want_syncs(4), lock_timeout(4), fastlock(false), resync_period(1),
        ctx(pipe_ctx(i, o, "mpeg_sync: pipebufs of two device contexts")), in(i),
        out(o, SIZE_RSPACKET * (scan_syncs + 1)), deconv(dc), h(NULL), last_locktime(0), first_run(true) {
    state_out = opt_writer(state_o);
    locktime_out = opt_writer(locktime_o);
  }
  void run() {
    if (scan_syncs != 8 || want_syncs != 4 || lock_timeout != 4)   // dvb.h:727-729 defaults: what the device state machine implements
      fail("mpeg_sync: scan_syncs / want_syncs / lock_timeout other than the reference's defaults (8 / 4 / 4) are not implemented");
    if (!h) lsdr_check(lsdr_mpeg_sync_create(ctx, fastlock, &h), name);
    lsdr_check(lsdr_mpeg_sync_set_resync_period(h, resync_period), name);
    // one run() writes at most: the initial "unlocked" report plus one lock/unlock event (dvb.h:744-754)
    if (state_out && state_out->writable() < (first_run ? 2ul : 1ul)) return;
    first_run = false;
    unsigned long room = out.writable();
    if (locktime_out) {   // one locktime value per packet (dvb.h:858-859): bound the packets by the pipe's room
      unsigned long lt_room = locktime_out->writable();
      if (lsdr_mpeg_sync_locked(h) && room > lt_room * SIZE_RSPACKET) room = lt_room * SIZE_RSPACKET;
    }
    size_t consumed = 0, produced = 0;
    int events[8], n_events = 0, call_next = 0;
    unsigned long locktime = 0;
    bool was_locked = lsdr_mpeg_sync_locked(h);
    lsdr_check(lsdr_mpeg_sync_run(h, in.rd(), in.readable(), out.wr(), room, &consumed, &produced, events, &n_events,
                                  &locktime, &call_next), name);
    in.read(consumed);
    out.written(produced);
    for (int k = 0; k < n_events; ++k)
      if (state_out) state_out->write(events[k]);
    if (locktime_out && was_locked) {
      unsigned long n = produced / SIZE_RSPACKET;
      for (unsigned long k = 0; k < n; ++k) locktime_out->write(locktime - n + 1 + k);
    }
    if (call_next && deconv) deconv->next_sync();
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<u8> in;
  dev_writer<u8> out;
  deconvol_sync<u8, 0> *deconv;
  lsdr_mpeg_sync *h;
  unsigned long last_locktime;
  bool first_run;
  pipewriter<int> *state_out;
  pipewriter<unsigned long> *locktime_out;
};

template <typename Tbyte>
struct deinterleaver;

template <>
struct deinterleaver<u8> : runnable {
  deinterleaver(scheduler *sch, pipebuf<u8> &i, pipebuf<rspacket<u8> > &o)
      : runnable(sch, "deinterleaver"), ctx(pipe_ctx(i, o, "deinterleaver: pipebufs of two device contexts")),
        in(i), out(o) {}
  void run() {
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_deinterleaver_run(ctx, in.rd(), in.readable(), (uint8_t *)out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<u8> in;
  dev_writer<rspacket<u8> > out;
};

template <typename Tbyte, int BYTE_ERASED>
struct rs_decoder;

template <>
struct rs_decoder<u8, 0> : runnable {
  rs_decoder(scheduler *sch, pipebuf<rspacket<u8> > &i, pipebuf<tspacket> &o, pipebuf<int> *bitcount_o = NULL,
             pipebuf<int> *errcount_o = NULL)
      : runnable(sch, "RS decoder"), ctx(pipe_ctx(i, o, "rs_decoder: pipebufs of two device contexts")),
        in(i), out(o) {
    bitcount = opt_writer(bitcount_o);
    errcount = opt_writer(errcount_o);
  }
  void run() {
    if (bitcount && bitcount->writable() < 1) return;
    if (errcount && errcount->writable() < 1) return;
    unsigned long n = min(in.readable(), out.writable());
    if (!n) return;
    long bits = 0, errs = 0;
    lsdr_check(lsdr_rs_decoder_run(ctx, (uint8_t *)in.rd(), n, (uint8_t *)out.wr(), &bits, &errs), name);
    in.read(n);
    out.written(n);
    if (bitcount) bitcount->write((int)bits);
    if (errcount) errcount->write((int)errs);
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<rspacket<u8> > in;
  dev_writer<tspacket> out;
  pipewriter<int> *bitcount, *errcount;
};

struct derandomizer : runnable {
  derandomizer(scheduler *sch, pipebuf<tspacket> &i, pipebuf<tspacket> &o)
      : runnable(sch, "derandomizer"), ctx(pipe_ctx(i, o, "derandomizer: pipebufs of two device contexts")),
        in(i), out(o), h(NULL) {}
  void run() {
    if (!h) lsdr_check(lsdr_derandomizer_create(ctx, &h), name);
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_derandomizer_run(h, (const uint8_t *)in.rd(), in.readable(), (uint8_t *)out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<tspacket> in;
  dev_writer<tspacket> out;
  lsdr_derandomizer *h;
};

// ---- transmit chain (leandvbtx.cc:79-175): same constructor signatures as the reference blocks, device pipebufs.
struct randomizer : runnable {   // dvb.h:1063-1102
  randomizer(scheduler *sch, pipebuf<tspacket> &i, pipebuf<tspacket> &o)
      : runnable(sch, "derandomizer"), ctx(pipe_ctx(i, o, "randomizer: pipebufs of two device contexts")), in(i), out(o) {
    lsdr_check(lsdr_randomizer_create(ctx, &h), name);
  }
  void run() {
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_randomizer_run(h, (const uint8_t *)in.rd(), in.readable(), (uint8_t *)out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<tspacket> in;
  dev_writer<tspacket> out;
  lsdr_randomizer *h;
};

struct rs_encoder : runnable {   // dvb.h:957-980
  rs_encoder(scheduler *sch, pipebuf<tspacket> &i, pipebuf<rspacket<u8> > &o)
      : runnable(sch, "RS encoder"), ctx(pipe_ctx(i, o, "rs_encoder: pipebufs of two device contexts")), in(i), out(o) {}
  void run() {
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_rs_encoder_run(ctx, (const uint8_t *)in.rd(), in.readable(), (uint8_t *)out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<tspacket> in;
  dev_writer<rspacket<u8> > out;
};

struct interleaver : runnable {   // dvb.h:899-921
  interleaver(scheduler *sch, pipebuf<rspacket<u8> > &i, pipebuf<u8> &o)
      : runnable(sch, "interleaver"), ctx(pipe_ctx(i, o, "interleaver: pipebufs of two device contexts")), in(i),
        out(o, SIZE_RSPACKET) {}
  void run() {
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_interleaver_run(ctx, (const uint8_t *)in.rd(), in.readable(), out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<rspacket<u8> > in;
  dev_writer<u8> out;
};

struct dvb_convol : runnable {   // dvb.h:567-604
  typedef u8 uncoded_byte;
  typedef u8 hardsymbol;
  dvb_convol(scheduler *sch, pipebuf<uncoded_byte> &i, pipebuf<hardsymbol> &o, code_rate fec, int bits_per_symbol)
      : runnable(sch, "dvb_convol"), ctx(pipe_ctx(i, o, "dvb_convol: pipebufs of two device contexts")), in(i),
        out(o, 64) {
    lsdr_check(lsdr_convol_create(ctx, (int)fec, bits_per_symbol, &h), name);
  }
  void run() {
    for (;;) {   // until no progress: the reference's run() is a while loop that re-evaluates writable() (which may pack)
      unsigned long room = out.writable();
      size_t consumed = 0, produced = 0;
      lsdr_check(lsdr_convol_run(h, in.rd(), in.readable(), out.wr(), room, &consumed, &produced), name);
      if (!consumed && !produced) break;
      in.read(consumed);
      out.written(produced);
    }
  }

 private:
  lsdr_ctx *ctx;
  dev_reader<uncoded_byte> in;
  dev_writer<hardsymbol> out;
  lsdr_convol *h;
};

}  // namespace leansdr
#endif

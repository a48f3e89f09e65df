// leansdr_amd/host/leansdr/framework.h — host-side data-flow runtime of the MI355X build.
//
// What a leandvb-shaped graph builder needs from the reference's framework.h is its vocabulary:
//   scheduler{verbose, debug, run(), shutdown(), dump()}, runnable(sch, name){run(), shutdown()},
//   pipebuf<T>(sch, name, size), pipewriter<T>(buf, min_write){writable(), wr(), written(n), write(v)},
//   pipereader<T>(buf){readable(), rd(), read(n)}, opt_writer/opt_writable/opt_write, fail()/fatal(), u8 … s32.
// This file provides that vocabulary on its own implementation: pipes are cursor-based (one write index, one read index
// per attached reader, storage compacted on demand), the scheduler keeps its blocks and pipes in vectors, and — the
// reason the file exists — a pipe has a host side and an MI355X-HBM side (see pipebuf below): blocks written like the
// reference's attach on the host side, GPU-backed blocks on the device side, and the pipe carries the items across PCIe
// on side streams where the two meet.
//
// Behaviour that graphs rely on (checked against framework.h:45-249 by the app-level golden tests):
//   * readers always see one contiguous span; a writer that finds less tail room than the largest min_write of the
//     pipe's writers triggers compaction of the unread span to the front; a pipe nobody reads never fills up;
//   * the scheduler calls every block once per pass, in construction order, and stops after the first pass in which
//     no pipe moved (its fingerprint weighs pipe i by i+1) — end of input is a fixpoint, not an event.
#ifndef LEANSDR_AMD_FRAMEWORK_H
#define LEANSDR_AMD_FRAMEWORK_H

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "lsdr_hip.h"

#ifndef VERSION
#define VERSION "leansdr_amd"
#endif

namespace leansdr {

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned long u32;   // 8 bytes on x86-64, like the reference's (graphs carry pipebuf<u32> lock times)
typedef signed char s8;
typedef signed short s16;
typedef signed long s32;

// ---- errors: print and leave, no exceptions, no status codes ------------------------------------------------------
inline void fail(const char *what) {
  fprintf(stderr, "** %s\n", what);
  exit(1);
}
inline void fatal(const char *syscall) {
  perror(syscall);
  exit(1);
}
inline void lsdr_check(int status, const char *where) {   // C-ABI status → fail(), as include/lsdr_hip.h prescribes
  if (status == 0) return;
  fprintf(stderr, "** %s: lsdr error %d: %s\n", where, status, lsdr_last_error());
  exit(1);
}

template <typename T>
T min(const T &a, const T &b) { return b < a ? b : a; }
template <typename T>
T max(const T &a, const T &b) { return a < b ? b : a; }

inline float gen_abs(float v) { return fabsf(v); }
inline int gen_abs(int v) { return abs(v); }
inline long int gen_abs(long int v) { return labs(v); }
inline float gen_sqrt(float v) { return sqrtf(v); }
inline unsigned int gen_sqrt(unsigned int v) { return sqrtl(v); }
inline long double gen_sqrt(long double v) { return sqrtl(v); }
inline float gen_hypot(float a, float b) { return hypotf(a, b); }
inline long double gen_hypot(long double a, long double b) { return hypotl(a, b); }
inline float gen_atan2(float y, float x) { return atan2f(y, x); }
inline long double gen_atan2(long double y, long double x) { return atan2l(y, x); }

struct window_placement {   // --gui layout table entry (name == NULL ends a table); kept for option compatibility
  const char *name;
  int x, y, w, h;
};

struct scheduler;

namespace detail {
// What the scheduler needs to know about a pipe and a block.
struct pipe_base {
  const char *name;
  // A block that writes this pipe and can be FUSED with the block that reads it (auto_notch → fir_filter, dsp.h / sdr.h) leaves
  // itself here; NULL otherwise.  The reader decides (and checks that it is the pipe's only one).
  void *fusable_producer;
  explicit pipe_base(const char *n) : name(n), fusable_producer(NULL) {}
  virtual ~pipe_base() {}
  virtual unsigned long long traffic() const = 0;   // items written + items read so far
  virtual size_t bytes() const = 0;
  virtual void describe(FILE *f) const = 0;
};
struct block_base {
  const char *name;
  explicit block_base(const char *n) : name(n) {}
  virtual ~block_base() {}
  virtual void run() {}
  virtual void shutdown() {}
  virtual void prepare() {}      // once, before the scheduler's first pass: the whole graph exists by then (every end is attached)
};
}  // namespace detail

struct scheduler {
  bool verbose, debug;
  window_placement *windows;
  scheduler() : verbose(false), debug(false), windows(NULL), prepared_(false) {}

  void attach(detail::pipe_base *p) { pipes_.push_back(p); }
  void attach(detail::block_base *b) { blocks_.push_back(b); }

  void step() {
    if (!prepared_) {            // decisions that need the complete graph (dsp.h: fir_filter taking over the auto_notch in front of it)
      prepared_ = true;
      for (size_t i = 0; i < blocks_.size(); ++i) blocks_[i]->prepare();
    }
    for (size_t i = 0; i < blocks_.size(); ++i) blocks_[i]->run();
  }
  void run() {
    unsigned long long last = fingerprint();
    do {
      step();
    } while (moved(&last));
  }
  void shutdown() {
    for (size_t i = 0; i < blocks_.size(); ++i) blocks_[i]->shutdown();
  }
  void dump() {
    size_t total = 0;
    fputc('\n', stderr);
    for (size_t i = 0; i < pipes_.size(); ++i) {
      pipes_[i]->describe(stderr);
      total += pipes_[i]->bytes();
    }
    fprintf(stderr, "Total buffer memory: %ld KiB\n", (unsigned long)(total >> 10));
  }

 private:
  std::vector<detail::pipe_base *> pipes_;
  std::vector<detail::block_base *> blocks_;
  bool prepared_;
  unsigned long long fingerprint() const {
    unsigned long long f = 0;
    for (size_t i = 0; i < pipes_.size(); ++i) f += (i + 1) * pipes_[i]->traffic();
    return f;
  }
  bool moved(unsigned long long *last) const {
    const unsigned long long now = fingerprint();
    const bool changed = now != *last;
    *last = now;
    return changed;
  }
};

struct runnable : detail::block_base {
  runnable(scheduler *s, const char *n) : detail::block_base(n), sch(s) { s->attach(this); }

 protected:
  scheduler *sch;
};

// The process-default device context: what a pipebuf constructed the reference's way — pipebuf(sch, name, size) — uses
// as soon as a GPU-backed block attaches to it.  LSDR_DEVICE selects the GPU (default 0).
// LSDR_ARENA_GIB=n: the device storage of the pipes (the `new T[size]` of the reference's pipebuf, framework.h:141-143) comes out of ONE
// allocation of n GiB, the fastest windows first (lsdr_arena, include/lsdr_hip.h: where a stream buffer lands in HBM is worth ±8 % on a
// filter that streams it) — worth it with large pipes (--buf-factor in the thousands); unset: ordinary allocations.
inline lsdr_ctx *default_ctx() {
  static lsdr_ctx *c = NULL;
  if (!c) {
    const char *e = getenv("LSDR_DEVICE");
    lsdr_check(lsdr_ctx_create(e ? atoi(e) : 0, NULL, &c), "default device context");
    const char *ag = getenv("LSDR_ARENA_GIB");
    if (ag && atoi(ag) > 0) {
      static lsdr_arena *arena = NULL;          // lives as long as the process, like the context
      if (lsdr_arena_create(c, (size_t)atoi(ag) << 30, &arena) == LSDR_OK) lsdr_check(lsdr_ctx_set_arena(c, arena), "LSDR_ARENA_GIB");
      else fprintf(stderr, "LSDR_ARENA_GIB=%s: %s (ordinary allocations)\n", ag, lsdr_last_error());
    }
  }
  return c;
}

enum pipe_side { HOST_SIDE = 0, DEVICE_SIDE = 1 };

// Multi-reader FIFO over one linear index space with TWO address spaces.  `head` is the write index; reader r has
// consumed everything below tails[r].  Nothing wraps: when the tail room is short the span [oldest tail, head) is moved
// to index 0.
//
// Every end (writer or reader) lives on one side: HOST_SIDE — pipewriter/pipereader, i.e. any block that dereferences
// wr()/rd() on the CPU, exactly like a block written for the reference — or DEVICE_SIDE — dev_writer/dev_reader, the
// GPU-backed blocks, whose wr()/rd() are HBM pointers handed to the C ABI.  A side's storage exists only if an end lives
// there (host storage is pinned when the other side exists too).  Items committed on the writer's side are mirrored to
// the other side when — and only when — a reader lives there:
//   host writer → device readers   hipMemcpyAsync on the context's upload stream, enqueued at commit time; the compute
//                                  stream waits for it (an event, not the host) the next time a device reader looks;
//                                  the host goes on filling the next stretch of the pipe meanwhile (north_star's
//                                  double-buffered side-stream transfer: the pipe itself is the multi-buffer);
//   device writer → host readers   hipMemcpyAsync on the download stream after the producing kernels; a host reader's
//                                  readable() waits for it, so what it sees is what has arrived (the scheduler's
//                                  "no progress" test stays exact).
// So the graph of the reference's leandvb.cc — host file_reader, GPU blocks, host file_writer/printers on plain
// three-argument pipebufs — runs unchanged, and PCIe is crossed exactly where a pipe has ends on both sides.
template <typename T>
struct pipebuf : detail::pipe_base {
  lsdr_ctx *dev;   // device context: explicit (fourth constructor argument) or the process default once needed

  pipebuf(scheduler *s, const char *n, unsigned long size, lsdr_ctx *device = NULL)
      : detail::pipe_base(n), dev(device), host_(NULL), devp_(NULL), host_pinned_(false), wside_(-1), cap_(size), head_(0), need_(1),
        n_in_(0), n_out_(0), mirrored_(0), h2d_unfenced_(false), d2h_inflight_(false) {
    used_[0] = used_[1] = false;
    s->attach(this);
  }

  // -- ends
  void attach_writer(pipe_side side, unsigned long min_write) {
    if (wside_ >= 0 && wside_ != (int)side) { fprintf(stderr, "** %s: writers on both the host and the device side\n", name); exit(1); }
    wside_ = side;
    used_[side] = true;
    if (min_write > need_) need_ = min_write;
  }
  size_t n_readers() const { return tails_.size(); }
  // a writer that only works in larger steps than it said when it attached (dsp.h: the fused auto_notch + fir_filter produces a
  // whole 4096-sample block's outputs or nothing): compaction is triggered while that much tail room is missing
  void need_room(unsigned long items) { if (items > need_) need_ = items; }
  // a reader may say what it is (dsp.h passive_tap: a reader that can be switched off when nobody reads ITS output)
  void set_reader_owner(int r, void *owner) { owners_[r] = owner; }
  void *reader_owner(int r) const { return owners_[r]; }
  int attach_reader(pipe_side side) {
    used_[side] = true;
    tails_.push_back(head_);
    rside_.push_back((char)side);
    owners_.push_back(NULL);
    return (int)tails_.size() - 1;
  }
  // -- writer side
  unsigned long room() {
    if (cap_ - head_ < need_) compact();
    return cap_ - head_;
  }
  T *write_ptr() { return store(wside_ < 0 ? HOST_SIDE : (pipe_side)wside_) + head_; }
  void commit(unsigned long items) {
    if (items > cap_ - head_) { fprintf(stderr, "Bug: overflow to %s\n", name); exit(1); }
    head_ += items;
    n_in_ += items;
    if (items) mirror();
  }
  // -- reader side
  unsigned long pending(int r) {
    arrive((pipe_side)rside_[r]);
    return head_ - tails_[r];
  }
  T *read_ptr(int r) { return store((pipe_side)rside_[r]) + tails_[r]; }
  void consume(int r, unsigned long items) {
    if (items > head_ - tails_[r]) { fprintf(stderr, "Bug: underflow from %s\n", name); exit(1); }
    tails_[r] += items;
    n_out_ += items;
  }

  unsigned long long traffic() const { return n_in_ + n_out_; }
  size_t bytes() const { return cap_ * sizeof(T); }
  void describe(FILE *f) const {
    unsigned long div = 1;
    const char *suffix = "";
    if (n_in_ >= 1000000) { div = 1000000; suffix = "M"; }
    else if (n_in_ >= 10000) { div = 1000; suffix = "k"; }
    const unsigned long tail_room = cap_ - head_;
    fprintf(f, ".%-16s : %4ld%s/%4ld%s %6ld writable %c, %6d unread (", name, n_out_ / div, suffix, n_in_ / div, suffix, tail_room,
            tail_room < need_ ? '!' : ' ', (int)(head_ - oldest()));
    for (size_t r = 0; r < tails_.size(); ++r) fprintf(f, " %d", (int)(head_ - tails_[r]));
    fprintf(f, " )%s%s\n", used_[DEVICE_SIDE] ? " [HBM]" : "", used_[DEVICE_SIDE] && used_[HOST_SIDE] ? " [PCIe]" : "");
  }

 private:
  T *host_, *devp_;
  bool host_pinned_;
  bool used_[2];
  int wside_;
  unsigned long cap_, head_, need_;
  unsigned long n_in_, n_out_;
  unsigned long mirrored_;            // items below this index have been sent to the non-writer side
  bool h2d_unfenced_, d2h_inflight_;
  std::vector<unsigned long> tails_;
  std::vector<void *> owners_;
  std::vector<char> rside_;

  lsdr_ctx *ctx() {
    if (!dev) dev = default_ctx();
    return dev;
  }
  T *store(pipe_side side) {
    if (side == DEVICE_SIDE) {
      if (!devp_) {
        void *p = NULL;
        lsdr_check(lsdr_malloc(ctx(), cap_ * sizeof(T), &p), name);
        devp_ = static_cast<T *>(p);
      }
      return devp_;
    }
    if (!host_) {
      if (used_[DEVICE_SIDE]) {          // staging for PCIe transfers: pinned
        void *p = NULL;
        lsdr_check(lsdr_malloc_host(cap_ * sizeof(T), &p), name);
        host_ = static_cast<T *>(p);
        host_pinned_ = true;
      } else {
        host_ = reinterpret_cast<T *>(new char[cap_ * sizeof(T)]);
      }
    }
    return host_;
  }
  // send what the writer has committed to the other side, if somebody reads there
  void mirror() {
    if (wside_ < 0 || mirrored_ >= head_) return;
    const pipe_side other = wside_ == HOST_SIDE ? DEVICE_SIDE : HOST_SIDE;
    bool wanted = false;
    for (size_t r = 0; r < rside_.size(); ++r) wanted = wanted || rside_[r] == (char)other;
    if (!wanted) return;
    const size_t bytes = (head_ - mirrored_) * sizeof(T);
    if (wside_ == HOST_SIDE) {
      lsdr_check(lsdr_copy_h2d_async(ctx(), store(DEVICE_SIDE) + mirrored_, store(HOST_SIDE) + mirrored_, bytes), name);
      h2d_unfenced_ = true;
    } else {
      lsdr_check(lsdr_copy_d2h_async(ctx(), store(HOST_SIDE) + mirrored_, store(DEVICE_SIDE) + mirrored_, bytes), name);
      d2h_inflight_ = true;
    }
    mirrored_ = head_;
  }
  // make committed items visible to a reader on `side`
  void arrive(pipe_side side) {
    if (wside_ < 0 || (int)side == wside_) return;
    mirror();
    if (side == DEVICE_SIDE && h2d_unfenced_) {
      lsdr_check(lsdr_copy_fence(ctx()), name);       // GPU-side wait: the host does not block
      h2d_unfenced_ = false;
    }
    if (side == HOST_SIDE && d2h_inflight_) {
      lsdr_check(lsdr_copy_sync_d2h(ctx()), name);
      d2h_inflight_ = false;
    }
  }
  unsigned long oldest() const {
    unsigned long o = head_;   // no reader: everything written is already "consumed"
    for (size_t r = 0; r < tails_.size(); ++r)
      if (tails_[r] < o) o = tails_[r];
    return o;
  }
  void compact() {
    const unsigned long from = oldest();
    if (from == 0) return;
    const size_t live = (head_ - from) * sizeof(T);
    if (devp_) {
      // A pipe that also has a host side may have uploads / downloads in flight that use the old layout (they run on the
      // side streams): drain, move, drain.  A device-only pipe needs no host wait at all: the move is stream-ordered behind
      // the kernels that still read the old layout and ahead of the ones that will use the new one.
      const bool two_sided = host_ != NULL || h2d_unfenced_ || d2h_inflight_;
      if (two_sided) lsdr_check(lsdr_copy_sync_all(ctx()), name);
      h2d_unfenced_ = d2h_inflight_ = false;
      if (live) {
        lsdr_check(lsdr_memcpy_d2d(ctx(), devp_, devp_ + from, live), name);
        if (two_sided) lsdr_check(lsdr_ctx_sync(ctx()), name);
      }
    }
    if (host_ && live) memmove(host_, host_ + from, live);
    head_ -= from;
    mirrored_ = mirrored_ > from ? mirrored_ - from : 0;
    for (size_t r = 0; r < tails_.size(); ++r) tails_[r] -= from;
  }
};

// Host-side ends: the reference's pipewriter / pipereader (framework.h:185-249).
template <typename T>
struct pipewriter {
  pipebuf<T> &buf;
  pipewriter(pipebuf<T> &b, unsigned long min_write = 1) : buf(b) { buf.attach_writer(HOST_SIDE, min_write); }
  unsigned long writable() { return buf.room(); }
  T *wr() { return buf.write_ptr(); }
  void written(unsigned long n) { buf.commit(n); }
  void write(const T &v) {
    *buf.write_ptr() = v;
    buf.commit(1);
  }
};

template <typename T>
struct pipereader {
  pipebuf<T> &buf;
  int id;
  explicit pipereader(pipebuf<T> &b) : buf(b), id(b.attach_reader(HOST_SIDE)) {}
  unsigned long readable() { return buf.pending(id); }
  T *rd() { return buf.read_ptr(id); }
  void read(unsigned long n) { buf.consume(id, n); }
};

// Device-side ends, used by the GPU-backed blocks: wr()/rd() are HBM pointers for the C ABI.
template <typename T>
struct dev_writer {
  pipebuf<T> &buf;
  dev_writer(pipebuf<T> &b, unsigned long min_write = 1) : buf(b) { buf.attach_writer(DEVICE_SIDE, min_write); }
  unsigned long writable() { return buf.room(); }
  T *wr() { return buf.write_ptr(); }
  void written(unsigned long n) { buf.commit(n); }
};

template <typename T>
struct dev_reader {
  pipebuf<T> &buf;
  int id;
  explicit dev_reader(pipebuf<T> &b) : buf(b), id(b.attach_reader(DEVICE_SIDE)) {}
  unsigned long readable() { return buf.pending(id); }
  T *rd() { return buf.read_ptr(id); }
  void read(unsigned long n) { buf.consume(id, n); }
};

// The context a GPU-backed block works in: the explicit context of its pipes if any (all the same), else the default.
template <typename A>
lsdr_ctx *pipe_ctx(pipebuf<A> &a, const char *who) {
  if (!a.dev) a.dev = default_ctx();
  return a.dev;
}
template <typename A, typename B>
lsdr_ctx *pipe_ctx(pipebuf<A> &a, pipebuf<B> &b, const char *who) {
  lsdr_ctx *c = a.dev ? a.dev : b.dev;
  if (!c) c = default_ctx();
  if ((a.dev && a.dev != c) || (b.dev && b.dev != c)) fail(who);
  a.dev = b.dev = c;
  return c;
}

// Optional side outputs (measurement pipes): a NULL pipe means "not wired".
template <typename T>
pipewriter<T> *opt_writer(pipebuf<T> *b) { return b ? new pipewriter<T>(*b) : NULL; }
template <typename T>
bool opt_writable(pipewriter<T> *w, int n = 1) { return !w || w->writable() >= (unsigned long)n; }
template <typename T>
void opt_write(pipewriter<T> *w, T v) { if (w) w->write(v); }

}  // namespace leansdr

#endif  // LEANSDR_AMD_FRAMEWORK_H

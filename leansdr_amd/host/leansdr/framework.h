// leansdr_amd/host/leansdr/framework.h — host-side data-flow runtime of the MI355X build.
//
// What a leandvb-shaped graph builder needs from the reference's framework.h is its vocabulary:
//   scheduler{verbose, debug, run(), shutdown(), dump()}, runnable(sch, name){run(), shutdown()},
//   pipebuf<T>(sch, name, size), pipewriter<T>(buf, min_write){writable(), wr(), written(n), write(v)},
//   pipereader<T>(buf){readable(), rd(), read(n)}, opt_writer/opt_writable/opt_write, fail()/fatal(), u8 … s32.
// This file provides that vocabulary on its own implementation: pipes are cursor-based (one write index, one read index
// per attached reader, storage compacted on demand), the scheduler keeps its blocks and pipes in vectors, and — the
// reason the file exists — a pipe may live in MI355X HBM: pipebuf(sch, name, size, ctx).  Device pipes are touched only by
// GPU-backed blocks and by the h2d/d2h bridges of generic.h; compaction is then a stream-ordered device copy.
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
  explicit pipe_base(const char *n) : name(n) {}
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
};
}  // namespace detail

struct scheduler {
  bool verbose, debug;
  window_placement *windows;
  scheduler() : verbose(false), debug(false), windows(NULL) {}

  void attach(detail::pipe_base *p) { pipes_.push_back(p); }
  void attach(detail::block_base *b) { blocks_.push_back(b); }

  void step() {
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

// Multi-reader FIFO over one linear allocation.  `head` is the write index; reader r has consumed everything below
// tails[r].  Nothing wraps: when the tail room is short the span [oldest tail, head) is moved to index 0.
template <typename T>
struct pipebuf : detail::pipe_base {
  lsdr_ctx *dev;   // NULL: host memory; otherwise the storage is in this context's HBM

  pipebuf(scheduler *s, const char *n, unsigned long size, lsdr_ctx *device = NULL)
      : detail::pipe_base(n), dev(device), store_(NULL), cap_(size), head_(0), need_(1), n_in_(0), n_out_(0) {
    if (dev) {
      void *p = NULL;
      lsdr_check(lsdr_malloc(dev, cap_ * sizeof(T), &p), n);
      store_ = static_cast<T *>(p);
    } else {
      store_ = new T[cap_];
    }
    s->attach(this);
  }

  // -- writer side
  void require_room(unsigned long items) { if (items > need_) need_ = items; }
  unsigned long room() {
    if (cap_ - head_ < need_) compact();
    return cap_ - head_;
  }
  T *write_ptr() { return store_ + head_; }
  void commit(unsigned long items) {
    if (items > cap_ - head_) { fprintf(stderr, "Bug: overflow to %s\n", name); exit(1); }
    head_ += items;
    n_in_ += items;
  }
  // -- reader side
  int attach_reader() {
    tails_.push_back(head_);
    return (int)tails_.size() - 1;
  }
  unsigned long pending(int r) const { return head_ - tails_[r]; }
  T *read_ptr(int r) { return store_ + tails_[r]; }
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
    fprintf(f, " )%s\n", dev ? " [HBM]" : "");
  }

 private:
  T *store_;
  unsigned long cap_, head_, need_;
  unsigned long n_in_, n_out_;
  std::vector<unsigned long> tails_;

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
    if (dev) lsdr_check(lsdr_memcpy_d2d(dev, store_, store_ + from, live), name);
    else memmove(store_, store_ + from, live);
    head_ -= from;
    for (size_t r = 0; r < tails_.size(); ++r) tails_[r] -= from;
  }
};

template <typename T>
struct pipewriter {
  pipebuf<T> &buf;
  pipewriter(pipebuf<T> &b, unsigned long min_write = 1) : buf(b) { buf.require_room(min_write); }
  unsigned long writable() { return buf.room(); }
  T *wr() { return buf.write_ptr(); }
  void written(unsigned long n) { buf.commit(n); }
  void write(const T &v) {   // host pipes only
    *buf.write_ptr() = v;
    buf.commit(1);
  }
};

template <typename T>
struct pipereader {
  pipebuf<T> &buf;
  int id;
  explicit pipereader(pipebuf<T> &b) : buf(b), id(b.attach_reader()) {}
  unsigned long readable() { return buf.pending(id); }
  T *rd() { return buf.read_ptr(id); }
  void read(unsigned long n) { buf.consume(id, n); }
};

// Optional side outputs (measurement pipes): a NULL pipe means "not wired".
template <typename T>
pipewriter<T> *opt_writer(pipebuf<T> *b) { return b ? new pipewriter<T>(*b) : NULL; }
template <typename T>
bool opt_writable(pipewriter<T> *w, int n = 1) { return !w || w->writable() >= (unsigned long)n; }
template <typename T>
void opt_write(pipewriter<T> *w, T v) { if (w) w->write(v); }

}  // namespace leansdr

#endif  // LEANSDR_AMD_FRAMEWORK_H

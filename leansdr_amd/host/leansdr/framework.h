// leansdr_amd/host/leansdr/framework.h — host-side data-flow runtime.
//
// Same public surface as the reference's framework.h (scheduler, runnable,
// pipebuf, pipewriter, pipereader, opt_* helpers, fail/fatal, u8…s32), written
// from scratch, so that a leandvb-shaped flow graph compiles against it
// unchanged.  Semantics follow framework.h:45-249:
//   * pipebuf is a LINEAR multi-reader FIFO; readers always see contiguous
//     memory; when fewer than min_write slots remain the unread span is moved
//     to the front (pack);
//   * the scheduler runs every runnable once per step, in construction order,
//     until a whole step moves no pipe counter (fixpoint = end of input).
// Addition: a pipebuf may live in MI355X HBM (pipebuf(sch, name, size, ctx)).
// Device pipebufs are read/written only by GPU-backed blocks and by the
// h2d/d2h bridges in generic.h; pack() is then a stream-ordered device copy.
#ifndef LEANSDR_AMD_FRAMEWORK_H
#define LEANSDR_AMD_FRAMEWORK_H

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lsdr_hip.h"

#ifndef VERSION
#define VERSION "leansdr_amd"
#endif

namespace leansdr {

// Error convention of the reference: no exceptions, no return codes — a fatal
// condition prints and exits (framework.h:32-33).
inline void fatal(const char *s) { perror(s); exit(1); }
inline void fail(const char *s) { fprintf(stderr, "** %s\n", s); exit(1); }
// C-ABI status → fail(), as include/lsdr_hip.h prescribes for the shim blocks.
inline void lsdr_check(int rc, const char *where) {
  if (rc != 0) {
    fprintf(stderr, "** %s: lsdr error %d: %s\n", where, rc, lsdr_last_error());
    exit(1);
  }
}

static const int MAX_PIPES = 64;
static const int MAX_RUNNABLES = 64;
static const int MAX_READERS = 8;

struct pipebuf_common {
  const char *name;
  explicit pipebuf_common(const char *n) : name(n) {}
  virtual ~pipebuf_common() {}
  virtual int sizeofT() { return 0; }
  virtual long long hash() { return 0; }
  virtual void dump(size_t *total_bufs) { (void)total_bufs; }
};

struct runnable_common {
  const char *name;
  explicit runnable_common(const char *n) : name(n) {}
  virtual ~runnable_common() {}
  virtual void run() {}
  virtual void shutdown() {}
};

struct window_placement {
  const char *name;  // NULL terminates a table
  int x, y, w, h;
};

struct scheduler {
  pipebuf_common *pipes[MAX_PIPES];
  int npipes;
  runnable_common *runnables[MAX_RUNNABLES];
  int nrunnables;
  window_placement *windows;
  bool verbose, debug;

  scheduler() : npipes(0), nrunnables(0), windows(NULL), verbose(false), debug(false) {}

  void add_pipe(pipebuf_common *p) {
    if (npipes == MAX_PIPES) fail("MAX_PIPES");
    pipes[npipes++] = p;
  }
  void add_runnable(runnable_common *r) {
    if (nrunnables == MAX_RUNNABLES) fail("MAX_RUNNABLES");
    runnables[nrunnables++] = r;
  }
  // One pass over all blocks, construction order.
  void step() {
    for (int i = 0; i < nrunnables; ++i) runnables[i]->run();
  }
  // Progress fingerprint: pipe i contributes (1+i)·(items written + items read).
  unsigned long long hash() {
    unsigned long long h = 0;
    for (int i = 0; i < npipes; ++i) h += (unsigned long long)(1 + i) * pipes[i]->hash();
    return h;
  }
  // Run to the fixpoint: stop after the first step that moved nothing.
  void run() {
    unsigned long long before = 0;
    for (;;) {
      step();
      unsigned long long now = hash();
      if (now == before) return;
      before = now;
    }
  }
  void shutdown() {
    for (int i = 0; i < nrunnables; ++i) runnables[i]->shutdown();
  }
  void dump() {
    fprintf(stderr, "\n");
    size_t total = 0;
    for (int i = 0; i < npipes; ++i) pipes[i]->dump(&total);
    fprintf(stderr, "Total buffer memory: %ld KiB\n", (unsigned long)total / 1024);
  }
};

struct runnable : runnable_common {
  runnable(scheduler *s, const char *n) : runnable_common(n), sch(s) { sch->add_runnable(this); }

 protected:
  scheduler *sch;
};

template <typename T>
struct pipebuf : pipebuf_common {
  T *buf;
  T *rds[MAX_READERS];
  int nrd;
  T *wr;
  T *end;
  unsigned long min_write;
  unsigned long total_written, total_read;
  lsdr_ctx *dev;  // NULL: host memory.  Otherwise the buffer lives in this context's HBM.

  int sizeofT() { return sizeof(T); }

  pipebuf(scheduler *sch, const char *n, unsigned long size, lsdr_ctx *device = NULL)
      : pipebuf_common(n), nrd(0), min_write(1), total_written(0), total_read(0), dev(device) {
    if (dev) {
      void *p = NULL;
      lsdr_check(lsdr_malloc(dev, size * sizeof(T), &p), n);
      buf = (T *)p;
    } else {
      buf = new T[size];
    }
    wr = buf;
    end = buf + size;
    sch->add_pipe(this);
  }

  int add_reader() {
    if (nrd == MAX_READERS) fail("too many readers");
    rds[nrd] = wr;
    return nrd++;
  }

  // Slide the unread span [oldest reader, wr) to the start of the buffer.
  void pack() {
    T *oldest = wr;
    for (int i = 0; i < nrd; ++i)
      if (rds[i] < oldest) oldest = rds[i];
    size_t shift = oldest - buf;
    if (!shift) return;
    size_t bytes = (wr - oldest) * sizeof(T);
    if (dev) lsdr_check(lsdr_memcpy_d2d(dev, buf, oldest, bytes), name);
    else memmove(buf, oldest, bytes);
    wr -= shift;
    for (int i = 0; i < nrd; ++i) rds[i] -= shift;
  }

  long long hash() { return total_written + total_read; }

  void dump(size_t *total_bufs) {
    const unsigned long k = total_written < 10000 ? 1 : (total_written < 1000000 ? 1000 : 1000000);
    const char *unit = k == 1 ? "" : (k == 1000 ? "k" : "M");
    fprintf(stderr, ".%-16s : %4ld%s/%4ld%s", name, total_read / k, unit, total_written / k, unit);
    *total_bufs += (end - buf) * sizeof(T);
    unsigned long room = end - wr;
    fprintf(stderr, " %6ld writable %c,", room, room < min_write ? '!' : ' ');
    T *oldest = wr;
    for (int i = 0; i < nrd; ++i)
      if (rds[i] < oldest) oldest = rds[i];
    fprintf(stderr, " %6d unread (", (int)(wr - oldest));
    for (int i = 0; i < nrd; ++i) fprintf(stderr, " %d", (int)(wr - rds[i]));
    fprintf(stderr, " )%s\n", dev ? " [HBM]" : "");
  }
};

template <typename T>
struct pipewriter {
  pipebuf<T> &buf;
  pipewriter(pipebuf<T> &b, unsigned long min_write = 1) : buf(b) {
    if (min_write > buf.min_write) buf.min_write = min_write;
  }
  // Items writable at wr(); packs first when the tail room fell below min_write.
  unsigned long writable() {
    if ((unsigned long)(buf.end - buf.wr) < buf.min_write) buf.pack();
    return buf.end - buf.wr;
  }
  T *wr() { return buf.wr; }
  void written(unsigned long n) {
    if (buf.wr + n > buf.end) {
      fprintf(stderr, "Bug: overflow to %s\n", buf.name);
      exit(1);
    }
    buf.wr += n;
    buf.total_written += n;
  }
  void write(const T &e) {  // host pipebufs only
    *wr() = e;
    written(1);
  }
};

template <typename T>
pipewriter<T> *opt_writer(pipebuf<T> *buf) {
  return buf ? new pipewriter<T>(*buf) : NULL;
}
template <typename T>
bool opt_writable(pipewriter<T> *p, int n = 1) {
  return p == NULL || p->writable() >= (unsigned long)n;
}
template <typename T>
void opt_write(pipewriter<T> *p, T val) {
  if (p) p->write(val);
}

template <typename T>
struct pipereader {
  pipebuf<T> &buf;
  int id;
  explicit pipereader(pipebuf<T> &b) : buf(b), id(b.add_reader()) {}
  unsigned long readable() { return buf.wr - buf.rds[id]; }
  T *rd() { return buf.rds[id]; }
  void read(unsigned long n) {
    if (buf.rds[id] + n > buf.wr) {
      fprintf(stderr, "Bug: underflow from %s\n", buf.name);
      exit(1);
    }
    buf.rds[id] += n;
    buf.total_read += n;
  }
};

// Math helpers used by templated blocks (framework.h:251-277).
inline float gen_sqrt(float x) { return sqrtf(x); }
inline unsigned int gen_sqrt(unsigned int x) { return sqrtl(x); }
inline long double gen_sqrt(long double x) { return sqrtl(x); }
inline float gen_abs(float x) { return fabsf(x); }
inline int gen_abs(int x) { return abs(x); }
inline long int gen_abs(long int x) { return labs(x); }
inline float gen_hypot(float x, float y) { return hypotf(x, y); }
inline long double gen_hypot(long double x, long double y) { return hypotl(x, y); }
inline float gen_atan2(float y, float x) { return atan2f(y, x); }
inline long double gen_atan2(long double y, long double x) { return atan2l(y, x); }

template <typename T>
T min(const T &x, const T &y) { return (x < y) ? x : y; }
template <typename T>
T max(const T &x, const T &y) { return (x < y) ? y : x; }

// Integer abbreviations.  As in the reference (framework.h:281-286) u32/s32 are
// `long`, i.e. 8 bytes on x86-64; graphs rely on it (e.g. pipebuf<u32> locktime).
typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned long u32;
typedef signed char s8;
typedef signed short s16;
typedef signed long s32;

}  // namespace leansdr

#endif  // LEANSDR_AMD_FRAMEWORK_H

// leansdr_amd/host/leansdr/generic.h — host-side end points of a graph: descriptor I/O, the text reports of --fd-info /
// --fd-const / --fd-spectrum, the VBER ratio — all host-side ends: their pipes carry the items over PCIe when the other
// end is a GPU block (framework.h) — and two explicit host↔HBM bridge blocks.  Block names and constructor arguments are the reference's (generic.h:37-375) so that a graph builder
// written for it compiles; the bodies sit on two small helpers, fdio (whole-item descriptor transfers) and text_out.
#ifndef LEANSDR_AMD_GENERIC_H
#define LEANSDR_AMD_GENERIC_H

#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <sys/types.h>
#include <unistd.h>

#include <string>

#include "leansdr/framework.h"
#include "leansdr/math.h"

namespace leansdr {

namespace fdio {
// Up to max_bytes from fd; never returns a fraction of an item (a short tail is completed with blocking reads).
// Result: bytes read (multiple of item), 0 at end of file, −1 when a non-blocking descriptor has nothing yet.
inline ssize_t pull(int fd, void *dst, size_t max_bytes, size_t item) {
  ssize_t got = ::read(fd, dst, max_bytes);
  if (got < 0) {
    if (errno == EWOULDBLOCK || errno == EAGAIN) return -1;
    fatal("read");
  }
  for (size_t part = (size_t)got % item; part; part = (size_t)got % item) {
    ssize_t more = ::read(fd, static_cast<char *>(dst) + got, item - part);
    if (more <= 0) fatal("partial read");
    got += more;
  }
  return got;
}
// One write(); a short count is fine as long as it ends on an item boundary.
inline size_t push(int fd, const void *src, size_t bytes, size_t item) {
  ssize_t done = ::write(fd, src, bytes);
  if (done == 0) fatal("pipe");
  if (done < 0) fatal("write");
  if ((size_t)done % item) fatal("partial write");
  return (size_t)done;
}
}  // namespace fdio

// printf into a growing line, flushed to a descriptor with one write() per flush.
struct text_out {
  explicit text_out(int fd) : fd_(fd) {}
  void add(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
    char piece[256];
    va_list ap;
    va_start(ap, fmt);
    int len = vsnprintf(piece, sizeof(piece), fmt, ap);
    va_end(ap);
    if (len < 0) fatal("vsnprintf");
    line_.append(piece, (size_t)len < sizeof(piece) ? (size_t)len : sizeof(piece) - 1);
  }
  void flush() {
    for (size_t off = 0; off < line_.size();) {
      ssize_t w = ::write(fd_, line_.data() + off, line_.size() - off);
      if (w <= 0) fatal("partial write");
      off += (size_t)w;
    }
    line_.clear();
  }

 private:
  int fd_;
  std::string line_;
};

// ---- raw item streams ------------------------------------------------------------------------------------------------
// End of input is simply "no progress": the scheduler's fixpoint ends the run.  `loop` rewinds a seekable input instead;
// set_realtime() makes the descriptor non-blocking and substitutes `filler` items while nothing is available.
template <typename T>
struct file_reader : runnable {
  bool loop;
  file_reader(scheduler *s, int fd, pipebuf<T> &dst) : runnable(s, dst.name), loop(false), fd_(fd), to_(dst), have_filler_(false) {}
  void run() {
    const size_t room = to_.writable() * sizeof(T);
    if (room == 0) return;
    ssize_t got = fdio::pull(fd_, to_.wr(), room, sizeof(T));
    while (got == 0 && loop) {
      if (lseek(fd_, 0, SEEK_SET) == (off_t)-1) fatal("lseek");
      got = fdio::pull(fd_, to_.wr(), room, sizeof(T));
    }
    if (got > 0) to_.written((size_t)got / sizeof(T));
    else if (got < 0) {
      if (!have_filler_) fatal("read");        // EAGAIN on a descriptor nobody made non-blocking on purpose: an error, like the reference
      to_.write(filler_);
    }
  }
  void set_realtime(T &filler) {
    const int fl = fcntl(fd_, F_GETFL);
    if (fcntl(fd_, F_SETFL, fl | O_NONBLOCK) != 0) fatal("fcntl");
    filler_ = filler;
    have_filler_ = true;
  }

 private:
  int fd_;
  pipewriter<T> to_;
  T filler_;
  bool have_filler_;
};

template <typename T>
struct file_writer : runnable {
  file_writer(scheduler *s, pipebuf<T> &src, int fd) : runnable(s, src.name), from_(src), fd_(fd) {}
  void run() {
    const size_t avail = from_.readable() * sizeof(T);
    if (avail) from_.read(fdio::push(fd_, from_.rd(), avail, sizeof(T)) / sizeof(T));
  }

 private:
  pipereader<T> from_;
  int fd_;
};

// ---- text reports ----------------------------------------------------------------------------------------------------
// One formatted line per `decimation` items ("FREQ %.0f\n", "LOCK %d\n", …).
template <typename T>
struct file_printer : runnable {
  T scale;
  int decimation;
  file_printer(scheduler *s, const char *fmt, pipebuf<T> &src, int fd, int every = 1)
      : runnable(s, src.name), scale(1), decimation(every), from_(src), fmt_(fmt), sink_(fd), skipped_(0) {}
  void run() {
    const unsigned long n = from_.readable();
    const T *v = from_.rd();
    for (unsigned long i = 0; i < n; ++i) {
      if (++skipped_ < decimation) continue;
      skipped_ -= decimation;
      sink_.add(fmt_, v[i] * scale);
      sink_.flush();
    }
    from_.read(n);
  }

 private:
  pipereader<T> from_;
  const char *fmt_;
  text_out sink_;
  int skipped_;
};

// A batch of complex items per line: head(count) item sep item … tail.  fixed_size > 0 prints exactly that many per line
// (and waits for them); 0 prints whatever is readable.  --fd-const uses it for the SYMBOLS lines.
template <typename T>
struct file_carrayprinter : runnable {
  T scale;
  int fixed_size;
  file_carrayprinter(scheduler *s, const char *head, const char *item, const char *sep, const char *tail, pipebuf<complex<T> > &src,
                     int fd)
      : runnable(s, src.name), scale(1), fixed_size(0), from_(src), head_(head), item_(item), sep_(sep), tail_(tail), sink_(fd) {}
  void run() {
    const unsigned long least = fixed_size > 0 ? (unsigned long)fixed_size : 1ul;
    for (unsigned long n = from_.readable(); n >= least; n = from_.readable()) {
      if (fixed_size > 0) n = least;
      const complex<T> *z = from_.rd();
      sink_.add(head_, (int)n);
      for (unsigned long i = 0; i < n; ++i) {
        if (i) sink_.add("%s", sep_);
        sink_.add(item_, z[i].re * scale, z[i].im * scale);
      }
      sink_.add("%s", tail_);
      sink_.flush();
      from_.read(n);
    }
  }

 private:
  pipereader<complex<T> > from_;
  const char *head_, *item_, *sep_, *tail_;
  text_out sink_;
};

// One line per N-vector item (the 1024-bin SPECTRUM lines).
template <typename T, int N>
struct file_vectorprinter : runnable {
  T scale;
  file_vectorprinter(scheduler *s, const char *head, const char *item, const char *sep, const char *tail, pipebuf<T[N]> &src, int fd)
      : runnable(s, src.name), scale(1), from_(src), head_(head), item_(item), sep_(sep), tail_(tail), sink_(fd) {}
  void run() {
    for (; from_.readable(); from_.read(1)) {
      const T *v = *from_.rd();
      sink_.add(head_, N);
      for (int i = 0; i < N; ++i) {
        if (i) sink_.add("%s", sep_);
        sink_.add(item_, v[i] * scale);
      }
      sink_.add("%s", tail_);
      sink_.flush();
    }
  }

 private:
  pipereader<T[N]> from_;
  const char *head_, *item_, *sep_, *tail_;
  text_out sink_;
};

// Σ numerator / Σ denominator over windows of at least sample_size denominator counts — VBER = corrected bits / bits.
template <typename T>
struct rate_estimator : runnable {
  int sample_size;
  rate_estimator(scheduler *s, pipebuf<int> &numerator, pipebuf<int> &denominator, pipebuf<float> &ratio)
      : runnable(s, "rate_estimator"), sample_size(10000), num_(numerator), den_(denominator), ratio_(ratio), sum_num_(0), sum_den_(0) {}
  void run() {
    if (ratio_.writable() == 0) return;
    const unsigned long pairs = min(num_.readable(), den_.readable());
    const int *a = num_.rd(), *b = den_.rd();
    for (unsigned long i = 0; i < pairs; ++i) {
      sum_num_ += a[i];
      sum_den_ += b[i];
    }
    num_.read(pairs);
    den_.read(pairs);
    if (sum_den_ < sample_size) return;
    ratio_.write((float)sum_num_ / sum_den_);
    sum_num_ = sum_den_ = 0;
  }

 private:
  pipereader<int> num_, den_;
  pipewriter<float> ratio_;
  T sum_num_, sum_den_;
};

// ---- explicit bridges between two pipes -----------------------------------------------------------------------------
// A pipebuf carries its items across PCIe by itself wherever it has ends on both sides (framework.h); these two blocks
// remain for graphs that want the crossing as a block of its own (a host-side pipe copied into a separate HBM pipe).
template <typename T>
struct h2d_copier : runnable {
  h2d_copier(scheduler *s, lsdr_ctx *c, pipebuf<T> &host_src, pipebuf<T> &hbm_dst) : runnable(s, "h2d"), ctx_(c), from_(host_src), to_(hbm_dst) {}
  void run() {
    const unsigned long n = min(from_.readable(), to_.writable());
    if (n == 0) return;
    lsdr_check(lsdr_memcpy_h2d(ctx_, to_.wr(), from_.rd(), n * sizeof(T)), name);
    lsdr_check(lsdr_ctx_sync(ctx_), name);      // the producer may overwrite the host items once they are read()
    from_.read(n);
    to_.written(n);
  }

 private:
  lsdr_ctx *ctx_;
  pipereader<T> from_;
  dev_writer<T> to_;
};

template <typename T>
struct d2h_copier : runnable {
  d2h_copier(scheduler *s, lsdr_ctx *c, pipebuf<T> &hbm_src, pipebuf<T> &host_dst) : runnable(s, "d2h"), ctx_(c), from_(hbm_src), to_(host_dst) {}
  void run() {
    const unsigned long n = min(from_.readable(), to_.writable());
    if (n == 0) return;
    lsdr_check(lsdr_memcpy_d2h(ctx_, to_.wr(), from_.rd(), n * sizeof(T)), name);
    lsdr_check(lsdr_ctx_sync(ctx_), name);      // a host consumer must see the data
    from_.read(n);
    to_.written(n);
  }

 private:
  lsdr_ctx *ctx_;
  dev_reader<T> from_;
  pipewriter<T> to_;
};

}  // namespace leansdr
#endif

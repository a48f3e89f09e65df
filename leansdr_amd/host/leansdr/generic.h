// leansdr_amd/host/leansdr/generic.h — host glue blocks (generic.h:37-375 of the
// reference: file_reader/writer/printer, decimator, rate_estimator,
// buffer_reader/writer) plus the two bridges between host and HBM pipebufs.
// None of this is GPU work; the bridges are the only place where PCIe is crossed.
#ifndef LEANSDR_AMD_GENERIC_H
#define LEANSDR_AMD_GENERIC_H

#include <errno.h>
#include <fcntl.h>
#include <sys/types.h>
#include <unistd.h>

#include "leansdr/framework.h"
#include "leansdr/math.h"

namespace leansdr {

// Reads raw items from a file descriptor into a (host) pipebuf; end of input is
// simply "no progress" (generic.h:58-59) which stops the scheduler.
template <typename T>
struct file_reader : runnable {
  bool loop;
  file_reader(scheduler *sch, int fd, pipebuf<T> &o) : runnable(sch, o.name), loop(false), filler(NULL), fdin(fd), out(o) {}
  void run() {
    size_t room = out.writable() * sizeof(T);
    if (!room) return;
    for (;;) {
      ssize_t got = read(fdin, out.wr(), room);
      if (got < 0 && errno == EWOULDBLOCK && filler) { out.write(*filler); return; }
      if (got < 0) fatal("read");
      if (got == 0) {
        if (!loop) return;
        if (lseek(fdin, 0, SEEK_SET) == (off_t)-1) fatal("lseek");
        continue;
      }
      size_t tail = got % sizeof(T);  // complete a partially read item
      for (size_t need = tail ? sizeof(T) - tail : 0; need;) {
        ssize_t more = read(fdin, (char *)out.wr() + got, need);
        if (more <= 0) fatal("partial read");
        got += more;
        need -= more;
      }
      out.written(got / sizeof(T));
      return;
    }
  }
  void set_realtime(T &f) {
    int flags = fcntl(fdin, F_GETFL);
    if (fcntl(fdin, F_SETFL, flags | O_NONBLOCK)) fatal("fcntl");
    filler = new T(f);
  }

 private:
  T *filler;
  int fdin;
  pipewriter<T> out;
};

template <typename T>
struct file_writer : runnable {
  file_writer(scheduler *sch, pipebuf<T> &i, int fd) : runnable(sch, i.name), in(i), fdout(fd) {}
  void run() {
    size_t bytes = in.readable() * sizeof(T);
    if (!bytes) return;
    ssize_t nw = write(fdout, in.rd(), bytes);
    if (!nw) fatal("pipe");
    if (nw < 0) fatal("write");
    if (nw % sizeof(T)) fatal("partial write");
    in.read(nw / sizeof(T));
  }

 private:
  pipereader<T> in;
  int fdout;
};

// printf-style text output with optional decimation and scaling (generic.h:116-147).
template <typename T>
struct file_printer : runnable {
  T scale;
  int decimation;
  file_printer(scheduler *sch, const char *fmt, pipebuf<T> &i, int fd, int decim = 1)
      : runnable(sch, i.name), scale(1), decimation(decim), in(i), format(fmt), fdout(fd), phase(0) {}
  void run() {
    int n = in.readable();
    T *p = in.rd();
    for (int k = 0; k < n; ++k) {
      if (++phase >= decimation) {
        phase -= decimation;
        char line[256];
        int len = snprintf(line, sizeof(line), format, p[k] * scale);
        if (len < 0) fatal("obsolete glibc");
        if (write(fdout, line, len) != len) fatal("partial write");
      }
    }
    in.read(n);
  }

 private:
  pipereader<T> in;
  const char *format;
  int fdout;
  int phase;
};

// Batches of complex items as one text line each (generic.h:153-189) — SYMBOLS lines of --fd-const.
template <typename T>
struct file_carrayprinter : runnable {
  T scale;
  int fixed_size;   // items per batch, or 0
  file_carrayprinter(scheduler *sch, const char *head_, const char *format_, const char *sep_, const char *tail_,
                     pipebuf<complex<T> > &i, int fd)
      : runnable(sch, i.name), scale(1), fixed_size(0), in(i), head(head_), format(format_), sep(sep_), tail(tail_),
        fout(fdopen(fd, "w")) {}
  void run() {
    int n, nmin = fixed_size ? fixed_size : 1;
    while ((n = in.readable()) >= nmin) {
      if (fixed_size) n = fixed_size;
      if (fout) {
        fprintf(fout, head, n);
        complex<T> *pin = in.rd();
        for (int k = 0; k < n; ++k) {
          if (k) fprintf(fout, "%s", sep);
          fprintf(fout, format, pin[k].re * scale, pin[k].im * scale);
        }
        fprintf(fout, "%s", tail);
      }
      fflush(fout);
      in.read(n);
    }
  }

 private:
  pipereader<complex<T> > in;
  const char *head, *format, *sep, *tail;
  FILE *fout;
};

// One text line per vector item: head, N formatted values joined by sep, tail (generic.h:191-222).
template <typename T, int N>
struct file_vectorprinter : runnable {
  T scale;
  file_vectorprinter(scheduler *sch, const char *head_, const char *format_, const char *sep_, const char *tail_,
                     pipebuf<T[N]> &i, int fd)
      : runnable(sch, i.name), scale(1), in(i), head(head_), format(format_), sep(sep_), tail(tail_) {
    fout = fdopen(fd, "w");
    if (!fout) fatal("fdopen");
  }
  void run() {
    while (in.readable() >= 1) {
      fprintf(fout, head, N);
      T(*pin)[N] = in.rd();
      for (int k = 0; k < N; ++k) {
        if (k) fprintf(fout, "%s", sep);
        fprintf(fout, format, (*pin)[k] * scale);
      }
      fprintf(fout, "%s", tail);
      in.read(1);
    }
    fflush(fout);
  }

 private:
  pipereader<T[N]> in;
  const char *head, *format, *sep, *tail;
  FILE *fout;
};

// Ratio of two accumulated integer streams, emitted once the denominator
// reaches sample_size (generic.h:272-305) — VBER in leandvb.
template <typename T>
struct rate_estimator : runnable {
  int sample_size;
  rate_estimator(scheduler *sch, pipebuf<int> &n, pipebuf<int> &d, pipebuf<float> &r)
      : runnable(sch, "rate_estimator"), sample_size(10000), num(n), den(d), rate(r), acc_num(0), acc_den(0) {}
  void run() {
    if (rate.writable() < 1) return;
    int count = min(num.readable(), den.readable());
    int *pn = num.rd(), *pd = den.rd();
    for (int k = 0; k < count; ++k) { acc_num += pn[k]; acc_den += pd[k]; }
    num.read(count);
    den.read(count);
    if (acc_den >= sample_size) {
      rate.write((float)acc_num / acc_den);
      acc_num = acc_den = 0;
    }
  }

 private:
  pipereader<int> num, den;
  pipewriter<float> rate;
  T acc_num, acc_den;
};

template <typename T>
struct buffer_reader : runnable {
  buffer_reader(scheduler *sch, T *d, int n, pipebuf<T> &o) : runnable(sch, "buffer_reader"), data(d), count(n), out(o), pos(0) {}
  void run() {
    int n = min(out.writable(), (unsigned long)(count - pos));
    memcpy(out.wr(), data + pos, n * sizeof(T));
    pos += n;
    out.written(n);
  }

 private:
  T *data;
  int count;
  pipewriter<T> out;

 public:
  int pos;
};

template <typename T>
struct buffer_writer : runnable {
  buffer_writer(scheduler *sch, pipebuf<T> &i, T *d, int n) : runnable(sch, "buffer_writer"), in(i), data(d), count(n), pos(0) {}
  void run() {
    int n = min(in.readable(), (unsigned long)(count - pos));
    memcpy(data + pos, in.rd(), n * sizeof(T));
    in.read(n);
    pos += n;
  }

 private:
  pipereader<T> in;
  T *data;
  int count;

 public:
  int pos;
};

// ---- bridges (new): host pipebuf <-> HBM pipebuf --------------------------------
// The copy is enqueued on the context's stream, so it is ordered with the kernels
// of the neighbouring GPU blocks.  h2d waits for the copy before releasing the host
// items (the file_reader may overwrite them); d2h waits before publishing them.
template <typename T>
struct h2d_copier : runnable {
  h2d_copier(scheduler *sch, lsdr_ctx *c, pipebuf<T> &host_in, pipebuf<T> &dev_out)
      : runnable(sch, "h2d"), ctx(c), in(host_in), out(dev_out) {
    if (host_in.dev || !dev_out.dev) fail("h2d_copier: needs host input and device output pipebufs");
  }
  void run() {
    unsigned long n = min(in.readable(), out.writable());
    if (!n) return;
    lsdr_check(lsdr_memcpy_h2d(ctx, out.wr(), in.rd(), n * sizeof(T)), "h2d");
    lsdr_check(lsdr_ctx_sync(ctx), "h2d");
    in.read(n);
    out.written(n);
  }

 private:
  lsdr_ctx *ctx;
  pipereader<T> in;
  pipewriter<T> out;
};

template <typename T>
struct d2h_copier : runnable {
  d2h_copier(scheduler *sch, lsdr_ctx *c, pipebuf<T> &dev_in, pipebuf<T> &host_out)
      : runnable(sch, "d2h"), ctx(c), in(dev_in), out(host_out) {
    if (!dev_in.dev || host_out.dev) fail("d2h_copier: needs device input and host output pipebufs");
  }
  void run() {
    unsigned long n = min(in.readable(), out.writable());
    if (!n) return;
    lsdr_check(lsdr_memcpy_d2h(ctx, out.wr(), in.rd(), n * sizeof(T)), "d2h");
    lsdr_check(lsdr_ctx_sync(ctx), "d2h");
    in.read(n);
    out.written(n);
  }

 private:
  lsdr_ctx *ctx;
  pipereader<T> in;
  pipewriter<T> out;
};

}  // namespace leansdr
#endif

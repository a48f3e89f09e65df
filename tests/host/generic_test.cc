// tests/host/generic_test.cc — the text reports of generic.h (file_printer with decimation and scale, rate_estimator windows,
// file_carrayprinter batches, file_vectorprinter) on host pipes; prints everything to stdout.  The same source compiles
// against leansdr_amd/host and against the reference's headers: the two outputs must be identical.
#include <stdio.h>
#include <unistd.h>

#include "leansdr/framework.h"
#include "leansdr/generic.h"

using namespace leansdr;

template <typename T>
struct feeder : runnable {   // n items per pass from a table
  feeder(scheduler *s, pipebuf<T> &o, const T *v, int count, int per_pass_) : runnable(s, "feeder"), out(o), vals(v), n(count), pos(0), per_pass(per_pass_) {}
  void run() {
    for (int k = 0; k < per_pass && pos < n && out.writable() >= 1; ++k) out.write(vals[pos++]);
  }
  pipewriter<T> out;
  const T *vals;
  int n, pos, per_pass;
};

int main() {
  scheduler sch;
  static float fv[40];
  static int num[40], den[40];
  static complex<float> cv[23];
  static float vec[3][4];
  for (int i = 0; i < 40; ++i) { fv[i] = 0.25f * i - 3; num[i] = i % 5; den[i] = 1000 + 37 * i; }
  for (int i = 0; i < 23; ++i) cv[i] = complex<float>(i * 1.5f, -i * 0.5f);
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) vec[r][k] = r + k / 8.0f;

  pipebuf<float> p_f(&sch, "f", 8);
  feeder<float> src_f(&sch, p_f, fv, 40, 3);
  file_printer<float> pr_f(&sch, "F %.2f\n", p_f, 1, 4);   // every 4th item
  pr_f.scale = 2;

  pipebuf<int> p_n(&sch, "num", 4), p_d(&sch, "den", 4);
  pipebuf<float> p_r(&sch, "ratio", 4);
  feeder<int> src_n(&sch, p_n, num, 40, 2), src_d(&sch, p_d, den, 40, 2);
  rate_estimator<float> est(&sch, p_n, p_d, p_r);
  est.sample_size = 5000;
  file_printer<float> pr_r(&sch, "RATE %.6f\n", p_r, 1);

  pipebuf<complex<float> > p_c(&sch, "c", 16);
  feeder<complex<float> > src_c(&sch, p_c, cv, 23, 5);
  file_carrayprinter<float> pr_c(&sch, "SYMBOLS %d", " %.1f,%.1f", "", "\n", p_c, 1);
  pr_c.fixed_size = 6;
  pr_c.scale = 2;

  pipebuf<float[4]> p_v(&sch, "v", 2);
  feeder<float[4]> *unused = NULL; (void)unused;
  pipewriter<float[4]> w_v(p_v);
  file_vectorprinter<float, 4> pr_v(&sch, "VEC [", "%.3f", ",", "]\n", p_v, 1);
  for (int r = 0; r < 2; ++r) { for (int k = 0; k < 4; ++k) (*w_v.wr())[k] = vec[r][k]; w_v.written(1); }

  sch.run();
  sch.shutdown();
  return 0;
}

// tests/host/pipe_sides_test.cc — pipebufs with ends on both sides of PCIe (leansdr_amd/host/leansdr/framework.h).
//   stdin (cf32) → file_reader ─ p_raw ─┬─ scaler(×2) [GPU] ─ p_x2 ─┬─ scaler(×0.25) [GPU] ─ p_half ─ file_writer → fd 1
//                                        │                            └─ file_writer → fd 4          (host reader of a device-written pipe)
//                                        └─ file_writer → fd 3                                        (host reader next to a device reader)
// argv[1] = pipe size in items (small sizes force compaction while transfers are in flight).  The Python side checks the
// three outputs against the input (×0.5, ×2, ×1; exact in binary floating point).
#include <stdio.h>
#include <stdlib.h>
#include "leansdr/framework.h"
#include "leansdr/generic.h"
#include "leansdr/dsp.h"
#include "leansdr/sdr.h"
using namespace leansdr;

int main(int argc, char **argv) {
  unsigned long size = argc > 1 ? strtoul(argv[1], NULL, 10) : 4096;
  scheduler sch;
  pipebuf<cf32> p_raw(&sch, "raw", size);             // three-argument constructor, like the reference's graphs
  pipebuf<cf32> p_x2(&sch, "x2", size);
  pipebuf<cf32> p_half(&sch, "half", size + 37);      // different sizes: the pipes compact at different times
  file_reader<cf32> rd(&sch, 0, p_raw);
  scaler<float, cf32, cf32> a(&sch, 2.0f, p_raw, p_x2);
  scaler<float, cf32, cf32> b(&sch, 0.25f, p_x2, p_half);
  file_writer<cf32> w1(&sch, p_half, 1);
  file_writer<cf32> w4(&sch, p_x2, 4);
  file_writer<cf32> w3(&sch, p_raw, 3);
  sch.run();
  sch.shutdown();
  if (argc > 2) sch.dump();
  return 0;
}

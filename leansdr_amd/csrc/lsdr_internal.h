// leansdr_amd/csrc/lsdr_internal.h — shared internals of liblsdr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/lsdr_hip.h"

struct lsdr_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  hipEvent_t ev0, ev1;
  int num_cu;
  void *bounce;          // scratch of lsdr_memcpy_d2d for overlapping ranges (pipebuf::pack)
  size_t bounce_cap;
  // copy engine (lsdr_copy_*): one upload and one download stream next to the compute stream, created on first use
  hipStream_t up, down;
  hipEvent_t ev_up, ev_compute;
  bool copy_ready;
  // rs_decoder (fec.hip): GF(256) tables and the corrected-bits counter of THIS context (two contexts on one device decode
  // concurrently on their own streams)
  void *rs_tables;
  unsigned long long *rs_counter;
};

struct lsdr_event {
  lsdr_ctx *ctx;
  hipEvent_t ev;
};

void lsdr_set_error(const char *fmt, ...);
int lsdr_hip_fail(hipError_t e, const char *what, const char *file, int line);

#define LSDR_HIP(call)                                                  \
  do {                                                                  \
    hipError_t e__ = (call);                                            \
    if (e__ != hipSuccess) return lsdr_hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

#define LSDR_ARG(cond)                                                  \
  do {                                                                  \
    if (!(cond)) {                                                      \
      lsdr_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); \
      return LSDR_E_ARG;                                                \
    }                                                                   \
  } while (0)

// Host-side table builders (host_tables.cpp)
namespace lsdr {
void fir_shift_coeffs(unsigned ncoeffs, const float *coeffs, float freq, lsdr_cf32 *shifted);
struct cstln_tables {
  int nsymbols, nrotations;
  int8_t symbols[256][2];
  std::vector<int16_t> cost, phase_error;
  std::vector<uint8_t> symbol;
};
int build_cstln(int predef, int fec, cstln_tables &t);
// fast_qpsk_receiver::init_lookup_tables (sdr.h:1154-1171), packed: polar[re*256+im] = a | r<<16,
// rect[a*256+r] = re | im<<8, sincos[a] = re | im<<8.
void build_fastqpsk_tables(unsigned *polar, unsigned short *rect, unsigned short *sincos);
}  // namespace lsdr

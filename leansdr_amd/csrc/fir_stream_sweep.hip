// leansdr_amd/csrc/fir_stream_sweep.hip — k_fir_mfma_stream for every decimation 1 … 64, real and complex taps, with run-time tap blocks
// (any ncoeffs ≤ 16·D) and with ELEVEN as a compile-time constant — what leandvb's own filter design gives at every decimation with its
// default --resample-rej 10 and roll-off 0.35 (order ≈ 10.39·Fs/Fm ∈ (10·D, 11·D], leandvb.cc:364-366): the diagonal sums' addresses
// become immediates, 3–7 % at decimation 30:
// leandvb computes decim = Fs / (4·Fm) for whatever ratio it is given (leandvb.cc:353-378), and fir_filter<cf32,float>::run
// (dsp.h:233-280) takes any; the benchmark's 30 (and 10) have their hand-tuned forms in fir_filter.hip.  Compiled once per
// LSDR_SWEEP_PART = 0 … 7 (decimations ≡ part mod 8), so the instances build in parallel; each part exports one lookup.
#ifndef LSDR_SWEEP_PART
#error "compile with -DLSDR_SWEEP_PART=0..7"
#endif
#define LSDR_SWEEP_CAT2(a, b) a##b
#define LSDR_SWEEP_CAT(a, b) LSDR_SWEEP_CAT2(a, b)
#define LSDR_STREAM_NS LSDR_SWEEP_CAT(lsdr_fir_sweep, LSDR_SWEEP_PART)
#include "fir_stream.h"
using namespace lsdr_fir;
using namespace LSDR_STREAM_NS;

namespace {
template <int DT>
fir_kernel_t sweep_kernel(bool cplx, bool eleven) {
  constexpr int NPC = (int)stream_sweep_np(DT, true), NPR = (int)stream_sweep_np(DT, false);
  if (eleven) return cplx ? k_fir_mfma_stream<DT, 1, 11, 0, NPC, false> : k_fir_mfma_stream<DT, 0, 11, 0, NPR, false>;
  return cplx ? k_fir_mfma_stream<DT, 1, 0, 0, NPC, false> : k_fir_mfma_stream<DT, 0, 0, 0, NPR, false>;
}
template <int DT>
fir_kernel_t sweep_pick(unsigned D, bool cplx, bool eleven) {
  if constexpr (DT > (int)kStreamMaxD) return nullptr;
  else {
    if (D == (unsigned)DT) {
      if constexpr (DT >= 1 && DT != 10 && DT != 30) return sweep_kernel<DT>(cplx, eleven);    // (10, 30: fir_filter.hip)
      else return nullptr;
    }
    return sweep_pick<DT + 8>(D, cplx, eleven);
  }
}
}  // namespace

lsdr_fir::fir_kernel_t LSDR_SWEEP_CAT(lsdr_fir_stream_sweep_, LSDR_SWEEP_PART)(unsigned D, bool cplx, bool eleven) { return sweep_pick<LSDR_SWEEP_PART>(D, cplx, eleven); }

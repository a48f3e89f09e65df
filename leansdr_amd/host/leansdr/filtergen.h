// leansdr_amd/host/leansdr/filtergen.h — coefficient design with the reference's
// filtergen:: signatures (filtergen.h:26-104), forwarding to the C ABI's host-side
// builders so every consumer (C++ graph, Python tests, kernels) sees identical taps.
#ifndef LEANSDR_AMD_FILTERGEN_H
#define LEANSDR_AMD_FILTERGEN_H
#include <stdio.h>
#include "lsdr_hip.h"

namespace leansdr {
namespace filtergen {

inline void normalize_power(int n, float *coeffs, float gain = 1) { lsdr_filtergen_normalize_power(n, coeffs, gain); }
inline void normalize_dcgain(int n, float *coeffs, float gain = 1) { lsdr_filtergen_normalize_dcgain(n, coeffs, gain); }

inline int lowpass(int order, float Fcut, float **coeffs, float gain = 1) {
  *coeffs = new float[order + 1];
  return lsdr_filtergen_lowpass(order, Fcut, gain, *coeffs);
}
inline int root_raised_cosine(int order, float Fs, float rolloff, float **coeffs) {
  *coeffs = new float[(order + 1) | 1];
  return lsdr_filtergen_root_raised_cosine(order, Fs, rolloff, *coeffs);
}
inline void dump_filter(const char *name, int ncoeffs, float *coeffs) {
  fprintf(stderr, "%s = [", name);
  for (int i = 0; i < ncoeffs; ++i) fprintf(stderr, "%s %f", (i ? "," : ""), coeffs[i]);
  fprintf(stderr, " ];\n");
}

}  // namespace filtergen
}  // namespace leansdr
#endif

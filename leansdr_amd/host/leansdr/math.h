// leansdr_amd/host/leansdr/math.h — complex<T> and small integer helpers with
// the reference's names (math.h:22-115).  trig16 lives on the device; the host
// copy is built on demand through the C ABI (lsdr_trig16_table).
#ifndef LEANSDR_AMD_MATH_H
#define LEANSDR_AMD_MATH_H

#include <math.h>
#include <stdint.h>
#include "lsdr_hip.h"

namespace leansdr {

template <typename T>
struct complex {
  T re, im;
  complex() {}
  complex(T x) : re(x), im(0) {}
  complex(T x, T y) : re(x), im(y) {}
  inline void operator+=(const complex<T> &x) { re += x.re; im += x.im; }
};
template <typename T>
complex<T> operator+(const complex<T> &a, const complex<T> &b) { return complex<T>(a.re + b.re, a.im + b.im); }
template <typename T>
complex<T> operator*(const complex<T> &a, const complex<T> &b) {
  return complex<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <typename T>
complex<T> operator*(const complex<T> &a, const T &k) { return complex<T>(a.re * k, a.im * k); }
template <typename T>
complex<T> operator*(const T &k, const complex<T> &a) { return complex<T>(k * a.re, k * a.im); }

inline int hamming_weight(uint64_t x) { return __builtin_popcountll(x); }
inline int hamming_weight(uint32_t x) { return __builtin_popcount(x); }
inline int hamming_weight(uint16_t x) { return __builtin_popcount(x); }
inline int hamming_weight(uint8_t x) { return __builtin_popcount(x); }
inline unsigned char parity(uint64_t x) { return __builtin_parityll(x); }
inline unsigned char parity(uint32_t x) { return __builtin_parity(x); }
inline unsigned char parity(uint16_t x) { return __builtin_parity(x); }
inline unsigned char parity(uint8_t x) { return __builtin_parity(x); }
inline int log2i(uint64_t x) { return x ? 63 - __builtin_clzll(x) : -1; }

// 16-bit-angle cos/sin table (math.h:95-111), values from the C ABI's builder.
struct trig16 {
  complex<float> lut[65536];
  trig16() { lsdr_trig16_table(reinterpret_cast<lsdr_cf32 *>(lut)); }
  inline const complex<float> &expi(uint16_t a) const { return lut[a]; }
  inline const complex<float> &expi(float a) const { return expi((uint16_t)(int16_t)(int32_t)a); }
};

}  // namespace leansdr
#endif

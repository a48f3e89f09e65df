// leansdr_amd/host/leansdr/math.h — the one arithmetic type graphs name: complex<T> with public re / im (the layout of
// lsdr_cf32 / lsdr_cu8, so pipes of complex<float> are handed to the C ABI as they are).  All signal arithmetic of this
// build runs on the device; the host only moves and prints these values.
#ifndef LEANSDR_AMD_MATH_H
#define LEANSDR_AMD_MATH_H

#include <stdint.h>

namespace leansdr {

template <typename T>
struct complex {
  T re, im;
  complex() : re(), im() {}
  complex(T r, T i = T()) : re(r), im(i) {}
};

template <typename T>
inline complex<T> operator+(complex<T> a, complex<T> b) { return complex<T>(a.re + b.re, a.im + b.im); }
template <typename T>
inline complex<T> operator*(complex<T> a, T k) { return complex<T>(a.re * k, a.im * k); }

inline int log2i(uint64_t v) {   // position of the highest set bit, −1 for 0
  int n = -1;
  while (v) { v >>= 1; ++n; }
  return n;
}

}  // namespace leansdr
#endif

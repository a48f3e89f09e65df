// tools/membench.hip — read-bandwidth calibration for the roofline's practical ceiling.
// hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
template <typename V> __global__ void k_read(const V* __restrict__ in, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  for (; i < n; i += stride) { V v = in[i]; acc += reinterpret_cast<float*>(&v)[0]; }
  if (acc == 12345.678f) out[0] = acc;
}
// tile-style: each block reads a contiguous 64 KB chunk, all loads issued up front (like k_fir staging)
template <int NL> __global__ void k_read_tile(const float2* __restrict__ in, size_t ntiles, float* out) {
  float acc = 0;
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const float2* p = in + t * (size_t)(NL * 256) + threadIdx.x;
    float2 v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) v[k] = p[k * 256];
#pragma unroll
    for (int k = 0; k < NL; ++k) acc += v[k].x;
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  size_t bytes = 512ull << 20;
  void* d; float* o; CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 64)); CK(hipMemset(d, 1, bytes));
  for (int bpc : {2, 4, 8, 16}) {
    int grid = 256 * bpc;
    float ms = timeit([&] { hipLaunchKernelGGL(k_read<float2>, dim3(grid), dim3(256), 0, 0, (const float2*)d, bytes / 8, o); }, 5);
    printf("read float2 grid-stride  %4d blocks: %.3f ms %7.1f GB/s\n", grid, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_read<float4>, dim3(grid), dim3(256), 0, 0, (const float4*)d, bytes / 16, o); }, 5);
    printf("read float4 grid-stride  %4d blocks: %.3f ms %7.1f GB/s\n", grid, ms, bytes / ms / 1e6);
  }
  for (int bpc : {1, 2, 4, 8}) {
    int grid = 256 * bpc;
    size_t nt = bytes / (32 * 256 * 8);
    float ms = timeit([&] { hipLaunchKernelGGL(k_read_tile<32>, dim3(grid), dim3(256), 0, 0, (const float2*)d, nt, o); }, 5);
    printf("read 64KB tiles x32 f2   %4d blocks: %.3f ms %7.1f GB/s\n", grid, ms, bytes / ms / 1e6);
    nt = bytes / (8 * 256 * 8);
    ms = timeit([&] { hipLaunchKernelGGL(k_read_tile<8>, dim3(grid), dim3(256), 0, 0, (const float2*)d, nt, o); }, 5);
    printf("read 16KB tiles x8 f2    %4d blocks: %.3f ms %7.1f GB/s\n", grid, ms, bytes / ms / 1e6);
  }
  return 0;
}

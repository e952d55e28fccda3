// Round 6 probe: VALU issue rate of a LONE wave per SIMD (256 threads per CU, one workgroup per CU) on the GELU + GELU' arithmetic of the fc1 epilogue, as a function of
// how many independent value pairs are interleaved stage by stage (W = 2: one pair ... W = 16: eight pairs), packed fp32 ops.   hipcc --offload-arch=gfx950 -O3 -I visiondk_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "vdk_device.h"
template <int W> struct Vec { typedef float T __attribute__((ext_vector_type(W))); };
template <int W>
__device__ __forceinline__ void both_w(typename Vec<W>::T x, typename Vec<W>::T& g, typename Vec<W>::T& d) {
  typedef typename Vec<W>::T T;
  T ax, e, t;
#pragma unroll
  for (int i = 0; i < W; ++i) ax[i] = __uint_as_float(__float_as_uint(x[i]) & 0x7fffffffu);
  const T xx = x * x;
  const T ea = xx * (T)(-0.72134752044448170f);
  const T den = __builtin_elementwise_fma(ax, (T)(0.23164188843369479f), (T)(1.0f));
#pragma unroll
  for (int i = 0; i < W; ++i) { e[i] = __builtin_amdgcn_exp2f(ea[i]); t[i] = __builtin_amdgcn_rcpf(den[i]); }
  T p = __builtin_elementwise_fma(t, (T)(1.061405429f), (T)(-1.453152027f));
  p = __builtin_elementwise_fma(p, t, (T)(1.421413741f));
  p = __builtin_elementwise_fma(p, t, (T)(-0.284496736f));
  p = __builtin_elementwise_fma(p, t, (T)(0.254829592f));
  const T q = (p * t) * e;
  const T h = __builtin_elementwise_fma(q, (T)(-0.5f), (T)(0.5f));
  g = __builtin_elementwise_fma(ax, h, x * (T)(0.5f));
  T cs;
#pragma unroll
  for (int i = 0; i < W; ++i) cs[i] = copysignf(h[i], x[i]);
  d = __builtin_elementwise_fma(x * (T)(0.3989422804014327f), e, cs + (T)(0.5f));
}
template <int W>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  typedef typename Vec<W>::T T;
  T x;
#pragma unroll
  for (int i = 0; i < W; ++i) x[i] = (float)(threadIdx.x + i) * 0.01f - 1.0f;
  T accg = (T)(0.f), accd = (T)(0.f);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    T g, d;
    both_w<W>(x, g, d);
    accg += g; accd += d;
    x = x + (T)(0.001f);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < W; ++i) s += accg[i] + accd[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int W> void run(float* out, unsigned long long* cyc, int values) {
  const int iters = values / W;
  hipLaunchKernelGGL(k<W>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<W>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("W = %2d values interleaved: %.2f cycles per value (GELU + GELU', packed fp32, lone wave per SIMD)\n", W, (double)c / (double)(iters * W));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<2>(out, cyc, 1 << 14); run<4>(out, cyc, 1 << 14); run<8>(out, cyc, 1 << 14); run<16>(out, cyc, 1 << 14);
  return 0;
}

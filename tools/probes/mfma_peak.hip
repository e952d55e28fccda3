// Sustained MFMA issue rate on gfx950: back-to-back v_mfma_f32_32x32x16_bf16 with no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o gpurun_mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  s16x8 x = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, y = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int threads, int blocks, const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, 8);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double waves = (double)blocks * threads / 64, mf = waves * iters * 8 * NACC;
  const double flops = mf * 32.0 * 32 * 16 * 2;
  printf("%s: %d blocks x %d thr, NACC=%d: %.3f ms, %.1f TFLOP/s, counter %.0f ticks/wave-MFMA (counter is 100 MHz-class if small)\n", name, blocks, threads, NACC, ms,
         flops / ms / 1e9, (double)c / (iters * 8.0 * NACC));
}
int main() {
  run<4>(512, 256, "2 waves/SIMD");
  run<4>(512, 512, "2 waves/SIMD, 2 rounds");
  run<4>(256, 256, "1 wave/SIMD");
  run<2>(1024, 256, "4 waves/SIMD");
  run<4>(512, 256, "2 waves/SIMD again");
  return 0;
}

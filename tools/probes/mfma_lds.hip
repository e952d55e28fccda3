// What feeds v_mfma_f32_32x32x16_bf16 on gfx950 when the operands are realistic: (a) random register operands, (b) A fragments read
// from LDS every k-step (2 x ds_read_b128 per 4 MFMAs, the CBIR pre-filter's inner loop), 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void k(const s16x8* __restrict__ rnd, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 256];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  for (int i = threadIdx.x; i < 128 * 16; i += 512) ((s16x8*)lds)[i] = rnd[(blockIdx.x * 2048 + i) & 65535];
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  s16x8 qf[2][8];
  for (int q = 0; q < 2; ++q) for (int s = 0; s < 8; ++s) qf[q][s] = rnd[(threadIdx.x * 16 + q * 8 + s + blockIdx.x * 77) & 65535];
  s16x8 af[2] = {rnd[threadIdx.x & 65535], rnd[(threadIdx.x + 512) & 65535]};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (MODE == 1) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            const int row = h * 64 + rt * 32 + l31;
            af[rt] = *(const s16x8*)(lds + row * 256 + (((ks * 2 + hi) ^ (row & 15)) * 16));
          }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) acc[rt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt], qf[qt][ks], acc[rt][qt], 0, 0, 0);
      }
    if (MODE == 1) __syncthreads();
  }
  float s = 0;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const s16x8* rnd, const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, rnd, out, 50);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, rnd, out, iters);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * iters * 64 * 32.0 * 32 * 16 * 2;
  printf("%s: %.3f ms, %.1f TFLOP/s, %.2f us per 128x512x128 tile\n", name, ms, flops / ms / 1e9, ms * 1e3 / iters);
}
int main() {
  unsigned short* h = (unsigned short*)malloc(65536 * 16);
  srand(1);
  for (int i = 0; i < 65536 * 8; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.3f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  s16x8* rnd; (void)hipMalloc(&rnd, 65536 * 16); (void)hipMemcpy(rnd, h, 65536 * 16, hipMemcpyHostToDevice);
  run<0>(rnd, "random register operands");
  run<1>(rnd, "A fragments from LDS each k-step + barrier per tile");
  run<0>(rnd, "random register operands (again)");
  return 0;
}

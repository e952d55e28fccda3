// Empirical semantics of ds_read_b64_tr_b16 on gfx950: every lane passes the LDS address of ITS OWN 8-byte row
// (lane l -> element index 4*l .. 4*l+3 of a u16 ramp), and we print which ramp elements each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int stride_elems) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int lane = threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + lane * stride_elems));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {4, 16, 64}) {
    probe<<<1, 64>>>(d, stride);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d elems (lane l passes &lds[%d*l]); each entry = source element index -> (source lane, elem)\n", stride, stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf("  %4d=(L%2d,e%d)", h[l * 4 + j], h[l * 4 + j] / stride, h[l * 4 + j] % stride);
      printf("\n");
      if (l == 19) { printf("  ...\n"); l = 31; }
      if (l == 35) { printf("  ...\n"); l = 59; }
    }
  }
  return 0;
}

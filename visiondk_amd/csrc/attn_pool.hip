// attn_pool.hip — the attention step of timm's AttentionPoolLatent (global_pool='map': SigLIP ViTs, BASELINE.json configs[4]; timm layers/attention_pool.py;
// reference call site: models/classifier/classify_model.py:49-54 -> timm.create_model, and timm_wrapper.py:16-21 for the faceX / CBIR backbones):
//   out[b, h*64 + d] = sum_n softmax_n(scale * <q[h], k[b, n, h]>) v[b, n, h, d]        ONE latent query, N keys per image
// q is the projected latent (f32 [H*64], the same for every image), kv the bf16 [B*N, 2*H*64] output of the kv Linear (k | v halves).  This is HBM-bound
// streaming work (each k / v element is used once): one workgroup per (image, head), 256 threads, scores and probabilities in LDS, no MFMA.
// Algorithmic bytes: forward reads kv once (4 B per token and channel pair) ; backward reads kv once and writes dkv once.
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"

#define AP_MAXN 4096   // keys per image (probabilities live in LDS)

template <int OF>
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv, long ldkv, int N, int H, float scale,
                                                            float* __restrict__ out, long ldo, float* __restrict__ probs) {
  __shared__ float p_s[AP_MAXN];
  __shared__ float red[4];
  __shared__ float acc_s[4][64];
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x, D = H * 64;
  const bf16_t* kb = kv + (long)b * N * ldkv + h * 64;
  const bf16_t* vb = kb + D;
  // scores: one key per thread per pass; q[h] in registers (64 floats)
  float qr[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) qr[d] = q[h * 64 + d] * scale;
  float mx = -3.0e38f;
  for (int n = tid; n < N; n += 256) {
    const u32x4* kr = (const u32x4*)(kb + (long)n * ldkv);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 u = kr[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) { s = fmaf(qr[c * 8 + 2 * e], op_lo<OF>(u[e]), s); s = fmaf(qr[c * 8 + 2 * e + 1], op_hi<OF>(u[e]), s); }
    }
    p_s[n] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_max<4>(mx, red);
  float se = 0.f;
  for (int n = tid; n < N; n += 256) { const float e = expf(p_s[n] - mx); p_s[n] = e; se += e; }
  se = block_sum<4>(se, red);
  const float inv = 1.0f / se;
  for (int n = tid; n < N; n += 256) { const float p = p_s[n] * inv; p_s[n] = p; if (probs) probs[((long)b * H + h) * N + n] = p; }
  __syncthreads();
  // out[d] = sum_n p[n] v[n][d]: 4 key groups x 64 channels
  const int d = tid & 63, g = tid >> 6;
  float a = 0.f;
  for (int n = g; n < N; n += 4) a = fmaf(p_s[n], op2f<OF>(vb[(long)n * ldkv + d]), a);
  acc_s[g][d] = a;
  __syncthreads();
  if (tid < 64) out[(long)b * ldo + h * 64 + tid] = (acc_s[0][tid] + acc_s[1][tid]) + (acc_s[2][tid] + acc_s[3][tid]);
}

// dout f32 [B, H*64] -> dkv bf16 [B*N, 2*H*64] (dk | dv), dq_part f32 [B, H*64] (sum over images = dL/dq)
//   dp[n] = <dout[h], v[n]>, ds[n] = p[n] (dp[n] - sum_m p[m] dp[m]), dk[n] = scale * ds[n] q[h], dv[n] = p[n] dout[h], dq[h] += scale * sum_n ds[n] k[n]
template <int OF>
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv, long ldkv, const float* __restrict__ probs,
                                                            const float* __restrict__ dout, long lddo, int N, int H, float scale, bf16_t* __restrict__ dkv,
                                                            long lddkv, float* __restrict__ dq_part) {
  __shared__ float ds_s[AP_MAXN];
  __shared__ float red[4];
  __shared__ float acc_s[4][64];
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x, D = H * 64;
  const bf16_t* kb = kv + (long)b * N * ldkv + h * 64;
  const bf16_t* vb = kb + D;
  bf16_t* dkb = dkv + (long)b * N * lddkv + h * 64;
  bf16_t* dvb = dkb + D;
  const float* pr = probs + ((long)b * H + h) * N;
  float gr[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) gr[d] = dout[(long)b * lddo + h * 64 + d];
  float dot = 0.f;
  for (int n = tid; n < N; n += 256) {
    const u32x4* vr = (const u32x4*)(vb + (long)n * ldkv);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 u = vr[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) { s = fmaf(gr[c * 8 + 2 * e], op_lo<OF>(u[e]), s); s = fmaf(gr[c * 8 + 2 * e + 1], op_hi<OF>(u[e]), s); }
    }
    ds_s[n] = s;
    dot = fmaf(pr[n], s, dot);
  }
  dot = block_sum<4>(dot, red);
  for (int n = tid; n < N; n += 256) ds_s[n] = pr[n] * (ds_s[n] - dot);
  __syncthreads();
  const int d = tid & 63, g = tid >> 6;
  const float qd = q[h * 64 + d] * scale, gd = dout[(long)b * lddo + h * 64 + d];
  float a = 0.f;
  for (int n = g; n < N; n += 4) {
    const float ds = ds_s[n];
    dkb[(long)n * lddkv + d] = f2op<OF>(ds * qd);
    dvb[(long)n * lddkv + d] = f2op<OF>(pr[n] * gd);
    a = fmaf(ds, op2f<OF>(kb[(long)n * ldkv + d]), a);
  }
  acc_s[g][d] = a;
  __syncthreads();
  if (tid < 64) dq_part[(long)b * D + h * 64 + tid] = scale * ((acc_s[0][tid] + acc_s[1][tid]) + (acc_s[2][tid] + acc_s[3][tid]));
}

extern "C" {

// kv / dkv: 16-bit [B*N, 2*H*64] in the format `dtype` (VDK_BF16 | VDK_F16: the trunk's operand format)
int vdk_attn_pool_fwd_dt(const float* q, const void* kv, int64_t ldkv, int32_t B, int32_t N, int32_t H, float scale, float* out, int64_t ldo, float* probs, int32_t dtype,
                         void* stream) {
  if (!q || !kv || !out || B <= 0 || N <= 0 || H <= 0 || N > AP_MAXN || (ldkv & 7) || ldkv < 2L * H * 64 || (dtype != VDK_BF16 && dtype != VDK_F16))
    return vdk_fail(VDK_EINVAL, "vdk_attn_pool_fwd: bad argument (head_dim 64, N <= 4096, ldkv % 8 == 0)");
  if (dtype == VDK_F16)
    hipLaunchKernelGGL(attn_pool_fwd_kernel<VDK_OPF_F16>, dim3((unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q, (const bf16_t*)kv, (long)ldkv, (int)N, (int)H, scale, out,
                       (long)ldo, probs);
  else
    hipLaunchKernelGGL(attn_pool_fwd_kernel<VDK_OPF_BF16>, dim3((unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q, (const bf16_t*)kv, (long)ldkv, (int)N, (int)H, scale, out,
                       (long)ldo, probs);
  return vdk_check_launch("vdk_attn_pool_fwd");
}
int vdk_attn_pool_bwd_dt(const float* q, const void* kv, int64_t ldkv, const float* probs, const float* dout, int64_t lddo, int32_t B, int32_t N, int32_t H, float scale,
                         void* dkv, int64_t lddkv, float* dq_part, int32_t dtype, void* stream) {
  if (!q || !kv || !probs || !dout || !dkv || !dq_part || B <= 0 || N <= 0 || H <= 0 || N > AP_MAXN || (ldkv & 7) || ldkv < 2L * H * 64 || lddkv < 2L * H * 64 ||
      (dtype != VDK_BF16 && dtype != VDK_F16))
    return vdk_fail(VDK_EINVAL, "vdk_attn_pool_bwd: bad argument");
  if (dtype == VDK_F16)
    hipLaunchKernelGGL(attn_pool_bwd_kernel<VDK_OPF_F16>, dim3((unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q, (const bf16_t*)kv, (long)ldkv, probs, dout, (long)lddo,
                       (int)N, (int)H, scale, (bf16_t*)dkv, (long)lddkv, dq_part);
  else
    hipLaunchKernelGGL(attn_pool_bwd_kernel<VDK_OPF_BF16>, dim3((unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q, (const bf16_t*)kv, (long)ldkv, probs, dout, (long)lddo,
                       (int)N, (int)H, scale, (bf16_t*)dkv, (long)lddkv, dq_part);
  return vdk_check_launch("vdk_attn_pool_bwd");
}
int vdk_attn_pool_fwd(const float* q, const void* kv, int64_t ldkv, int32_t B, int32_t N, int32_t H, float scale, float* out, int64_t ldo, float* probs, void* stream) {
  return vdk_attn_pool_fwd_dt(q, kv, ldkv, B, N, H, scale, out, ldo, probs, VDK_BF16, stream);
}
int vdk_attn_pool_bwd(const float* q, const void* kv, int64_t ldkv, const float* probs, const float* dout, int64_t lddo, int32_t B, int32_t N, int32_t H, float scale,
                      void* dkv, int64_t lddkv, float* dq_part, void* stream) {
  return vdk_attn_pool_bwd_dt(q, kv, ldkv, probs, dout, lddo, B, N, H, scale, dkv, lddkv, dq_part, VDK_BF16, stream);
}

}  // extern "C"

// norm_loss.hip — K4 LayerNorm fwd/bwd (timm LayerNorm eps=1e-6 inside vision_transformer blocks and the
// final `norm`; F.layer_norm over the last dim), column reductions (bias / gamma / beta / pos_embed
// gradients) and K10 softmax cross-entropy with label smoothing (+ mixup pairs) fwd+bwd
// (models/losses/loss.py:71-73 `nn.CrossEntropyLoss(label_smoothing)`, engine/procedure/train.py:24-35
// mixup_criterion; BCE-with-logits loss.py:68-70).  All HBM-bound: one wave per row, 16-B accesses.
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"

// ------------------------------------------------------------------------------------------------
// y[r] = (x[r] - mean) * rstd * gamma + beta  (x fp32 rows at stride ldx; y bf16 or fp32; C % 4 == 0)
// MAXJ * 256 >= C: the row is loaded ONCE into registers (all loads in flight together) and mean, variance and output come from there; re-reading x for
// each of the three passes cost three dependent memory round trips per row (stage-0 ConvNeXt rows, C = 128: 672 us against a 246 us byte floor).
// The summation order per lane is the one of the three-pass form, so results are bit-identical to it.
// Q8 (fp8 mode of the ViT engine): the bf16 row is also written as fp8(clamp(value * q8_scale)) and max |value| goes to q8_amax -- what vdk_quant_fp8 would make of y
// (bit-identical: the ROUNDED bf16 values are quantised), without the extra pass over the tensor.  One wave per row: a wave publishes its maximum only when it can raise the
// (monotone) global value, so after the first wavefront almost nobody touches the atomic.
__device__ __forceinline__ unsigned ln_q8_pack4(const u32x2 bf, float s, float lim, int fmt, float& am) {
  float c[4];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float a = bf_lo(bf[e]), b = bf_hi(bf[e]);
    am = fmaxf(am, fmaxf(fabsf(a), fabsf(b)));
    c[2 * e] = fminf(fmaxf(a * s, -lim), lim); c[2 * e + 1] = fminf(fmaxf(b * s, -lim), lim);
  }
  int v = 0;
  if (fmt == 0) { v = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], v, false); v = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], v, true); }
  else { v = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], v, false); v = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], v, true); }
  return (unsigned)v;
}
// block-wide (4 waves, every thread arrives): one atomic per workgroup, and only when it can raise the (monotone) global value -- same-address atomics cost ~4.6 ns each
__device__ __forceinline__ void ln_q8_publish(float am, float* amax, int lane, float* red4) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) am = fmaxf(am, __shfl_xor(am, off));
  if (lane == 0) red4[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0 && amax) {
    const float m = fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
    if (m > *(volatile float*)amax) atomicMax((unsigned*)amax, __float_as_uint(m));      // non-negative floats order like their bit patterns
  }
}
template <bool OUT_BF16 /* 16-bit output in operand format OF (else fp32) */, int MAXJ, bool Q8 = false, int OF = 0>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, int T, int C,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, void* __restrict__ y, long ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, LnQ8 q8 = LnQ8()) {
  const int lane = threadIdx.x & 63;
  __shared__ float q8red[4];
  float q8s = 1.0f, q8lim = 448.0f, q8am = 0.f;
  if (Q8) { q8s = q8.scale ? q8.scale[0] : 1.0f; q8lim = q8.fmt == 0 ? 448.0f : 57344.0f; }
  // Q8: a bounded grid whose waves walk over rows, so that the amax is published by at most gridDim * 4 waves (one row per wave = 73 728 publishing waves at ViT-L/14 @336
  // cost more than the quantisation pass the by-product replaces: 171 vs 80 + 43 us); otherwise one row per wave, a single trip through the loop
  const int rstride = Q8 ? (int)gridDim.x * 4 : T;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < T; row += rstride) {
  const float* xr = x + (long)row * ldx;
  f32x4 v[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) { const int c = lane * 4 + j * 256; if (c < C) v[j] = *(const f32x4*)(xr + c); }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) { const int c = lane * 4 + j * 256; if (c < C) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]); }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = lane * 4 + j * 256;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[j][e] - mu; q = fmaf(d, d, q); }
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = lane * 4 + j * 256;
    if (c < C) {
      f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * g[e] + b[e];
      if (OUT_BF16) {
        const u32x2 ob = {pack_op2<OF>(o[0], o[1]), pack_op2<OF>(o[2], o[3])};
        *(u32x2*)((bf16_t*)y + (long)row * ldy + c) = ob;
        if (Q8) *(unsigned*)(q8.out + (long)row * q8.ld + c) = ln_q8_pack4(ob, q8s, q8lim, q8.fmt, q8am);
      }
      else *(f32x4*)((float*)y + (long)row * ldy + c) = (f32x4){o[0], o[1], o[2], o[3]};
    }
  }
  }
  if (Q8) ln_q8_publish(q8am, q8.amax, lane, q8red);
}

// C <= 128: a row is at most 32 lanes x 4 floats, so a wave takes TWO rows (one per half) instead of idling half of its lanes; same arithmetic per row
// (the half-wave butterfly adds the same 32 lane partials in the same order as the full-wave one does when lanes 32..63 hold zeros)
template <bool OUT_BF16, int OF = 0>
__global__ __launch_bounds__(256) void ln_fwd_narrow_kernel(const float* __restrict__ x, long ldx, int T, int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, void* __restrict__ y, long ldy, float* __restrict__ mean,
                                                            float* __restrict__ rstd) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l32 = threadIdx.x & 31, c = l32 * 4;
  const bool on = row < T && c < C;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (on) v = *(const f32x4*)(x + (long)row * ldx + c);
  float s = on ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)C;
  float q = 0.f;
  if (on) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { float d = v[e] - mu; q = fmaf(d, d, q); }
  }
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rs = 1.0f / sqrtf(q / (float)C + eps);
  if (row < T && l32 == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  if (on) {
    f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
    float o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = (v[e] - mu) * rs * g[e] + b[e];
    if (OUT_BF16) *(u32x2*)((bf16_t*)y + (long)row * ldy + c) = (u32x2){pack_op2<OF>(o4[0], o4[1]), pack_op2<OF>(o4[2], o4[3])};
    else *(f32x4*)((float*)y + (long)row * ldy + c) = (f32x4){o4[0], o4[1], o4[2], o4[3]};
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres],  g = dy * gamma;  per-block partial
// dgamma = sum dy * xhat, dbeta = sum dy.   MAXJ * 256 >= C.
// OCS: also per-block column sums of the bf16-rounded output dxb (= the bias gradient of the Linear whose dY this tensor is: the producer sums what it stores,
// instead of the consuming dgrad GEMM re-reading its A tiles from LDS) -> pout [block][C].
// NR rows per wave and pass, all loaded (x, dy and the residual gradient) before the first reduction: with narrow rows (Swin's 128 .. 512 channels) a wave's single row
// is too few bytes in flight to stream at HBM rate.  HALF (C <= 128): a row is 32 lanes, so a wave takes two rows side by side.
template <int MAXJ, bool DY_BF16 /* 16-bit dy in operand format OF (else fp32) */, bool OCS, bool Q8 = false, int OF = 0 /* format of a 16-bit dy and of dxb */, int NR = 1,
          bool HALF = false, bool DR16 = false /* the residual gradient arrives as 16-bit rows (dres16, format OF) instead of fp32 dres */>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                     long ldx, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     long lddres, int T, int C, int rows_per_block, float* __restrict__ dx,
                                                     long lddx, bf16_t* __restrict__ dxb, long lddxb,
                                                     float* __restrict__ pgamma, float* __restrict__ pbeta, float* __restrict__ pout, LnQ8 q8 = LnQ8(),
                                                     const float* __restrict__ dy_scale = nullptr /* device scalar: dy is multiplied by dy_scale[0] as it is loaded (a gradient branch kept at its own power-of-two scale: ConvNeXt's layer-scale branch under fp16 operands) */,
                                                     const float* __restrict__ dxb_rs = nullptr /* per-sample factor of the 16-bit COPY dxb only (dx stays): dxb = 16bit(dx * dxb_rs[row / dxb_rps]) -- the
                                                     gradient entering a residual branch under stochastic depth (timm DropPath: mask_b / keep_prob), whose GEMMs read dxb */, int dxb_rps = 1,
                                                     const bf16_t* __restrict__ dres16 = nullptr) {
  static_assert(!HALF || MAXJ == 1, "HALF: one 128-column slab");
  __shared__ float red[4][MAXJ * 256];
  float q8s = 1.0f, q8lim = 448.0f, q8am = 0.f;
  if (Q8) { q8s = q8.scale ? q8.scale[0] : 1.0f; q8lim = q8.fmt == 0 ? 448.0f : 57344.0f; }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int HS = HALF ? 2 : 1, RW = NR * HS;          // rows a wave holds per pass
  const int cl = HALF ? (lane & 31) : lane, sub = HALF ? (lane >> 5) : 0;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block; if (r1 > T) r1 = T;
  float ag[MAXJ][4], ab[MAXJ][4], ao[OCS ? MAXJ : 1][4];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; if (OCS) ao[j][e] = 0.f; }
  const float invC = 1.0f / (float)C;
  const float dsc = dy_scale ? dy_scale[0] : 1.0f;
  for (int rb = r0 + w * RW; rb < r1; rb += 4 * RW) {
    float rs[NR], s1[NR], s2[NR];
    float gk[NR][MAXJ][4], xh[NR][MAXJ][4], rv[NR][MAXJ][4];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int row = rb + k * HS + sub;
      const bool ok = row < r1;
      const float mu = ok ? mean[row] : 0.f;
      rs[k] = ok ? rstd[row] : 0.f;
      s1[k] = 0.f; s2[k] = 0.f;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = cl * 4 + j * 256;
#pragma unroll
        for (int e = 0; e < 4; ++e) { gk[k][j][e] = 0.f; xh[k][j][e] = 0.f; rv[k][j][e] = 0.f; }
        if (ok && c < C) {
          f32x4 xv = *(const f32x4*)(x + (long)row * ldx + c);
          f32x4 gm = *(const f32x4*)(gamma + c);
          float d[4];
          if (DY_BF16) {
            u32x2 u = *(const u32x2*)((const bf16_t*)dy + (long)row * lddy + c);
            d[0] = op_lo<OF>(u[0]); d[1] = op_hi<OF>(u[0]); d[2] = op_lo<OF>(u[1]); d[3] = op_hi<OF>(u[1]);
          } else {
            f32x4 u = *(const f32x4*)((const float*)dy + (long)row * lddy + c);
            d[0] = u[0]; d[1] = u[1]; d[2] = u[2]; d[3] = u[3];
          }
          if (dy_scale) { d[0] *= dsc; d[1] *= dsc; d[2] *= dsc; d[3] *= dsc; }
          if (DR16) {
            const u32x2 r2 = *(const u32x2*)(dres16 + (long)row * lddres + c);
            rv[k][j][0] = op_lo<OF>(r2[0]); rv[k][j][1] = op_hi<OF>(r2[0]); rv[k][j][2] = op_lo<OF>(r2[1]); rv[k][j][3] = op_hi<OF>(r2[1]);
          } else if (dres) {
            f32x4 r4 = *(const f32x4*)(dres + (long)row * lddres + c);
            rv[k][j][0] = r4[0]; rv[k][j][1] = r4[1]; rv[k][j][2] = r4[2]; rv[k][j][3] = r4[3];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float h = (xv[e] - mu) * rs[k];
            float g = d[e] * gm[e];
            xh[k][j][e] = h; gk[k][j][e] = g;
            s1[k] += g; s2[k] = fmaf(g, h, s2[k]);
            ag[j][e] = fmaf(d[e], h, ag[j][e]);
            ab[j][e] += d[e];
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
#pragma unroll
      for (int m = HALF ? 16 : 32; m >= 1; m >>= 1) { s1[k] += __shfl_xor(s1[k], m); s2[k] += __shfl_xor(s2[k], m); }
      s1[k] *= invC; s2[k] *= invC;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int row = rb + k * HS + sub;
      const bool ok = row < r1;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = cl * 4 + j * 256;
        if (ok && c < C) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs[k] * (gk[k][j][e] - s1[k] - xh[k][j][e] * s2[k]) + rv[k][j][e];
          if (dx) *(f32x4*)(dx + (long)row * lddx + c) = (f32x4){o[0], o[1], o[2], o[3]};
          if (dxb) {
            if (dxb_rs) { const float f = dxb_rs[row / dxb_rps]; o[0] *= f; o[1] *= f; o[2] *= f; o[3] *= f; }
            const u32x2 ob = {pack_op2<OF>(o[0], o[1]), pack_op2<OF>(o[2], o[3])};
            *(u32x2*)(dxb + (long)row * lddxb + c) = ob;
            if (OCS) { ao[j][0] += op_lo<OF>(ob[0]); ao[j][1] += op_hi<OF>(ob[0]); ao[j][2] += op_lo<OF>(ob[1]); ao[j][3] += op_hi<OF>(ob[1]); }
            if (Q8) *(unsigned*)(q8.out + (long)row * q8.ld + c) = ln_q8_pack4(ob, q8s, q8lim, q8.fmt, q8am);
          }
        }
      }
    }
  }
  if (Q8) ln_q8_publish(q8am, q8.amax, lane, &red[0][0]);      // (red is free until the combine below; the combine starts with its own barrier-separated phase)
  // combine the 4 waves' column partials: one 16 KB buffer, one phase per quantity (the kernel is HBM-bound: LDS per block decides how many blocks a CU holds)
#pragma unroll
  for (int ph = 0; ph < (OCS ? 3 : 2); ++ph) {
    if (ph) __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[w][j * 256 + lane * 4 + e] = ph == 0 ? ag[j][e] : (ph == 1 ? ab[j][e] : ao[OCS ? j : 0][e]);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
      if (HALF) v += (red[0][c + 128] + red[1][c + 128]) + (red[2][c + 128] + red[3][c + 128]);      // (lanes 32..63 = the odd rows, stored at columns 128 + c)
      // per-block partials side by side, [block][2C]: one reduction launch serves dgamma and dbeta when they are adjacent in the gradient buffer
      if (ph == 0) pgamma[(long)blockIdx.x * 2 * C + c] = v;
      else if (ph == 1) pbeta[(long)blockIdx.x * 2 * C + c] = v;
      else pout[(long)blockIdx.x * C + c] = v;
    }
  }
}

// out[c] = scale * sum_{s<S} in[s*ld + c]   (deterministic).  Block = 64 columns x 8 row groups; each thread strides
// over S/8 rows (coalesced 256-B row segments), the 4 groups are combined through LDS in a fixed order.
__global__ __launch_bounds__(512) void reduce_rows_f32_kernel(const float* __restrict__ in, long ld, int S, long n,
                                                              float* __restrict__ out, float scale) {
  __shared__ float red[8][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long c = (long)blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < n) {
    int k = rg;
    for (; k + 8 < S; k += 16) { s0 += in[(long)k * ld + c]; s1 += in[(long)(k + 8) * ld + c]; }
    if (k < S) s0 += in[(long)k * ld + c];
  }
  red[rg][cl] = s0 + s1;
  __syncthreads();
  if (rg == 0 && c < n)
    out[c] = (((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]))) * scale;
}

// Several such reductions in ONE launch (the six small ones a transformer block's backward leaves behind: two LayerNorm dgamma|dbeta pairs, four Linear
// bias gradients): ~10 us each as separate launches, i.e. at the launch-latency floor.  Jobs travel by value in the kernel arguments.
struct ReduceBatch { VdkReduceJob job[8]; int first_block[9]; int n; };
__global__ __launch_bounds__(512) void reduce_rows_batch_kernel(ReduceBatch b) {
  __shared__ float red[8][64];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
  const VdkReduceJob jb = b.job[j];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long c = (long)((int)blockIdx.x - b.first_block[j]) * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < jb.n) {
    int k = rg;
    for (; k + 8 < jb.S; k += 16) { s0 += jb.in[(long)k * jb.ld + c]; s1 += jb.in[(long)(k + 8) * jb.ld + c]; }
    if (k < jb.S) s0 += jb.in[(long)k * jb.ld + c];
  }
  red[rg][cl] = s0 + s1;
  __syncthreads();
  if (rg == 0 && c < jb.n)
    jb.out[c] = (((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]))) * jb.scale;
}

// The same for jobs of several hundred to a few thousand partial rows (the LayerNorm backward leaves one per workgroup: ~1000): 16 columns x 32 row groups per workgroup,
// four loads in flight per thread -- S = 1004 is 8 dependent steps instead of the 63 of the form above (28 us -> launch-latency floor).  first_block counts 16-column blocks.
__global__ __launch_bounds__(512) void reduce_rows_batch_tall_kernel(ReduceBatch b) {
  __shared__ float red[32][17];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
  const VdkReduceJob jb = b.job[j];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const long c = (long)((int)blockIdx.x - b.first_block[j]) * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < jb.n) {
    int k = rg;
    for (; k + 96 < jb.S; k += 128) {
      s0 += jb.in[(long)k * jb.ld + c]; s1 += jb.in[(long)(k + 32) * jb.ld + c]; s2 += jb.in[(long)(k + 64) * jb.ld + c]; s3 += jb.in[(long)(k + 96) * jb.ld + c];
    }
    for (; k < jb.S; k += 32) s0 += jb.in[(long)k * jb.ld + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < jb.n) {
    float t[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) t[g] = (red[4 * g][cl] + red[4 * g + 1][cl]) + (red[4 * g + 2][cl] + red[4 * g + 3][cl]);
    jb.out[c] = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) * jb.scale;
  }
}

// A TALL job (thousands of partial rows: the column-sum by-product of a GEMM over 1.6 M rows leaves 12 544 of them) is folded first: row g < 64 of the partial buffer
// becomes the sum of the rows g, g + 64, g + 128, ... (in place: block (columns, g) reads and writes only rows = g mod 64), 64 x n / 64 workgroups instead of the
// n / 64 of the batch kernel, whose threads would each walk S / 8 rows with two loads in flight (376 us for 25 MB).  The order of the sum is fixed: bit-reproducible.
#define RR_FOLD 64
__global__ __launch_bounds__(512) void reduce_rows_fold_kernel(float* __restrict__ buf, long ld, int S, long n) {
  __shared__ float red[8][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, g = blockIdx.y;
  const long c = (long)blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < n) {
    int k = g + RR_FOLD * rl;
    for (; k + 8 * RR_FOLD < S; k += 16 * RR_FOLD) { s0 += buf[(long)k * ld + c]; s1 += buf[(long)(k + 8 * RR_FOLD) * ld + c]; }
    if (k < S) s0 += buf[(long)k * ld + c];
  }
  red[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && c < n)
    buf[(long)g * ld + c] = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
}

// partial[split][c] = sum over this split's rows of in[r][c]  (bf16 input).  Block = 32 column chunks (16 B = 8 columns)
// x 8 row lanes; every load is 16 B and a wave reads 512 contiguous bytes of a row.
// Q8: the pass also writes the fp8 copy of the tensor it reads (and its amax) -- the attention backward's dqkv needs both its column sums (qkv.bias) and its e5m2 copy
// (operand of the next fp8 GEMM) in the engine's fp8 mode: one read instead of two.
template <bool Q8, int OF = 0>
__global__ __launch_bounds__(256) void colsum_bf16_partial_kernel(const bf16_t* __restrict__ in, long ld, int T, int N,
                                                                  int rows_per_split, float* __restrict__ partial, LnQ8 q8) {
  __shared__ float red[8][32][9];
  float q8s = 1.0f, q8lim = 448.0f, q8am = 0.f;
  if (Q8) { q8s = q8.scale ? q8.scale[0] : 1.0f; q8lim = q8.fmt == 0 ? 448.0f : 57344.0f; }
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cx) * 8;
  const int r0 = blockIdx.y * rows_per_split;
  int r1 = r0 + rows_per_split; if (r1 > T) r1 = T;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c < N) {
    int r = r0 + ry;
    for (; r + 24 < r1; r += 32) {                      // four independent 16-byte loads in flight per thread (the summation order per thread is unchanged)
      u32x4 u[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = *(const u32x4*)(in + (long)(r + 8 * k) * ld + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += op_lo<OF>(u[k][e]); acc[2 * e + 1] += op_hi<OF>(u[k][e]); }
        if (Q8) *(u32x2*)(q8.out + (long)(r + 8 * k) * q8.ld + c) = (u32x2){ln_q8_pack4((u32x2){u[k][0], u[k][1]}, q8s, q8lim, q8.fmt, q8am),
                                                                           ln_q8_pack4((u32x2){u[k][2], u[k][3]}, q8s, q8lim, q8.fmt, q8am)};
      }
    }
    for (; r < r1; r += 8) {
      u32x4 u = *(const u32x4*)(in + (long)r * ld + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += op_lo<OF>(u[e]); acc[2 * e + 1] += op_hi<OF>(u[e]); }
      if (Q8) *(u32x2*)(q8.out + (long)r * q8.ld + c) = (u32x2){ln_q8_pack4((u32x2){u[0], u[1]}, q8s, q8lim, q8.fmt, q8am), ln_q8_pack4((u32x2){u[2], u[3]}, q8s, q8lim, q8.fmt, q8am)};
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ry][cx][e] = acc[e];
  __syncthreads();
  if (ry == 0 && c < N) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][cx][e];
      partial[(long)blockIdx.y * N + c + e] = t;
    }
  }
  if (Q8) { __syncthreads(); ln_q8_publish(q8am, q8.amax, threadIdx.x & 63, &red[0][0][0]); }
}

// ------------------------------------------------------------------------------------------------
// softmax cross-entropy, label smoothing eps, optional mixup pair (yb, lam):
//   loss_r = lse - (1-eps) * (lam * x[ya] + (1-lam) * x[yb]) - eps * mean_c(x)
//   dlogits = gscale * (softmax - (1-eps) * (lam*1[ya] + (1-lam)*1[yb]) - eps/C)
// one workgroup per row.  dlogits bf16 [B, lddl] with columns C..lddl-1 zeroed (they pad the K dim of
// the head dgrad/wgrad GEMMs).
// OF: format of the 16-bit dlogits.  gscale_dev (optional, device scalar): multiplies gscale -- GradScaler's loss scale (train.py:205 `scaler.scale(loss).backward()`),
// read on the device so that a skipped step / a grown scale needs no host round trip.
template <int OF = 0>
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, long ldl, int C,
                                                         const long long* __restrict__ ya, const long long* __restrict__ yb,
                                                         float lam, float eps, float gscale, float* __restrict__ loss_rows,
                                                         bf16_t* __restrict__ dlogits, long lddl, float* __restrict__ dlogits_f32,
                                                         long lddf, const float* __restrict__ gscale_dev = nullptr) {
  __shared__ float red[4];
  if (gscale_dev) gscale *= gscale_dev[0];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long)row * ldl;
  float mx = -3.0e38f, sm = 0.f;
  for (int c = tid; c < C; c += 256) { float v = x[c]; mx = fmaxf(mx, v); sm += v; }
  mx = block_max<4>(mx, red);
  sm = block_sum<4>(sm, red);
  float se = 0.f;
  for (int c = tid; c < C; c += 256) se += expf(x[c] - mx);
  se = block_sum<4>(se, red);
  const float lse = mx + logf(se);
  const int a = (int)ya[row];
  const int b = yb ? (int)yb[row] : a;
  const float wa = yb ? lam : 1.0f, wb = yb ? (1.0f - lam) : 0.0f;
  if (tid == 0 && loss_rows)
    loss_rows[row] = lse - (1.0f - eps) * (wa * x[a] + wb * x[b]) - eps * (sm / (float)C);
  const float inv = 1.0f / se, epsc = eps / (float)C;
  for (int c = tid; c < (int)lddl || c < C; c += 256) {
    float g = 0.f;
    if (c < C) {
      g = expf(x[c] - mx) * inv - epsc;
      if (c == a) g -= (1.0f - eps) * wa;
      if (c == b) g -= (1.0f - eps) * wb;
      g *= gscale;
      if (dlogits_f32) dlogits_f32[(long)row * lddf + c] = g;
    }
    if (dlogits && c < (int)lddl) dlogits[(long)row * lddl + c] = f2op<OF>(g);
  }
}

// BCE-with-logits: loss_e = max(x,0) - x*t + log1p(exp(-|x|)); d = (sigmoid(x) - t) * gscale.
// focal_gamma > 0: FocalLoss(BCEWithLogits) of models/losses/loss.py:27-54: loss_e *= alpha_t * (1 - p_t)^gamma with
// p_t = t*p + (1-t)*(1-p), alpha_t = t*alpha + (1-t)*(1-alpha); the gradient differentiates the modulating factor too.
template <int OF = 0>
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ logits, long ldl, const float* __restrict__ tgt,
                                                         long ldt, int C, float gscale, float focal_gamma, float focal_alpha,
                                                         float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits, long lddl,
                                                         float* __restrict__ dlogits_f32, long lddf, const float* __restrict__ gscale_dev = nullptr) {
  __shared__ float red[4];
  if (gscale_dev) gscale *= gscale_dev[0];
  const int row = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int c = tid; c < (int)lddl || c < C; c += 256) {
    float g = 0.f;
    if (c < C) {
      const float x = logits[(long)row * ldl + c], t = tgt[(long)row * ldt + c];
      const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      const float p = 1.0f / (1.0f + expf(-x));
      if (focal_gamma > 0.f) {
        const float pt = t * p + (1.0f - t) * (1.0f - p);
        const float at = t * focal_alpha + (1.0f - t) * (1.0f - focal_alpha);
        const float om = fmaxf(1.0f - pt, 0.f);
        const float mod = powf(om, focal_gamma);
        s += bce * at * mod;
        const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
        const float dmod = om > 0.f ? -focal_gamma * powf(om, focal_gamma - 1.0f) * dpt : 0.f;
        g = at * ((p - t) * mod + bce * dmod) * gscale;
      } else {
        s += bce;
        g = (p - t) * gscale;
      }
      if (dlogits_f32) dlogits_f32[(long)row * lddf + c] = g;
    }
    if (dlogits && c < (int)lddl) dlogits[(long)row * lddl + c] = f2op<OF>(g);
  }
  s = block_sum<4>(s, red);
  if (tid == 0 && loss_rows) loss_rows[row] = s;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm1d over [B, F] (the TimmWrapper neck's last layer, models/faceX/backbone/timm_wrapper.py:37,46; eps 1e-5,
// momentum 0.1).  One thread per feature (coalesced across features), B <= a few thousand rows.
// train: batch mean / biased var normalise; running stats updated with the unbiased var.  eval: running stats.
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, long ldx, int B, int F, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float momentum, int training,
                                                       float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ y, long ldy,
                                                       float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  float mu, var;
  if (training) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += x[(long)b * ldx + f];
    mu = s / (float)B;
    float q = 0.f;
    for (int b = 0; b < B; ++b) { float d = x[(long)b * ldx + f] - mu; q = fmaf(d, d, q); }
    var = q / (float)B;
    if (rmean) rmean[f] = (1.0f - momentum) * rmean[f] + momentum * mu;
    if (rvar) rvar[f] = (1.0f - momentum) * rvar[f] + momentum * (B > 1 ? q / (float)(B - 1) : var);
  } else { mu = rmean[f]; var = rvar[f]; }
  const float is = 1.0f / sqrtf(var + eps);
  if (save_mean) save_mean[f] = mu;
  if (save_invstd) save_invstd[f] = is;
  const float g = gamma[f] * is, bb = beta[f];
  for (int b = 0; b < B; ++b) y[(long)b * ldy + f] = (x[(long)b * ldx + f] - mu) * g + bb;
}
// training-mode backward: dx = gamma * invstd / B * (B * dy - sum(dy) - xhat * sum(dy * xhat))
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx, int B, int F,
                                                       const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_invstd, float* __restrict__ dx, long lddx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const float mu = save_mean[f], is = save_invstd[f];
  float sg = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) { float d = dy[(long)b * lddy + f]; sb += d; sg = fmaf(d, (x[(long)b * ldx + f] - mu) * is, sg); }
  dgamma[f] = sg; dbeta[f] = sb;
  const float k = gamma[f] * is / (float)B;
  for (int b = 0; b < B; ++b) {
    const float xh = (x[(long)b * ldx + f] - mu) * is;
    dx[(long)b * lddx + f] = k * ((float)B * dy[(long)b * lddy + f] - sb - xh * sg);
  }
}

// Row-parallel variants for many samples (BatchNorm2d of the CNN neck on NHWC rows: B*H*W = 25 088 samples x 1024 channels at cfg3).
// A workgroup owns 32 columns; its 512 threads = 32 columns x 16 row groups stride over the rows, the column sums meet in LDS.  Same
// arithmetic as above (mean, then centred variance), deterministic, no workspace.
#define BNR_COLS 32
#define BNR_RG 16
__device__ __forceinline__ float bnr_colsum(float v, float* red /*[16][32]*/, int cl, int rg) {
  red[rg * BNR_COLS + cl] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < BNR_RG; ++i) t += red[i * BNR_COLS + cl];
  __syncthreads();
  return t;
}
__global__ __launch_bounds__(512) void bnrows_fwd_kernel(const float* __restrict__ x, long ldx, int B, int F, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float momentum, int training,
                                                         float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ y, long ldy,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  __shared__ float red[BNR_RG * BNR_COLS];
  const int cl = threadIdx.x & (BNR_COLS - 1), rg = threadIdx.x / BNR_COLS;
  const int f = blockIdx.x * BNR_COLS + cl;
  const bool ok = f < F;
  float mu = 0.f, var = 1.f;
  if (training) {
    float s = 0.f;
    if (ok) for (int b = rg; b < B; b += BNR_RG) s += x[(long)b * ldx + f];
    mu = bnr_colsum(s, red, cl, rg) / (float)B;
    float q = 0.f;
    if (ok) for (int b = rg; b < B; b += BNR_RG) { const float d = x[(long)b * ldx + f] - mu; q = fmaf(d, d, q); }
    q = bnr_colsum(q, red, cl, rg);
    var = q / (float)B;
    if (ok && rg == 0) {
      if (rmean) rmean[f] = (1.0f - momentum) * rmean[f] + momentum * mu;
      if (rvar) rvar[f] = (1.0f - momentum) * rvar[f] + momentum * (B > 1 ? q / (float)(B - 1) : var);
    }
  } else if (ok) { mu = rmean[f]; var = rvar[f]; }
  if (!ok) return;
  const float is = 1.0f / sqrtf(var + eps);
  if (rg == 0) { if (save_mean) save_mean[f] = mu; if (save_invstd) save_invstd[f] = is; }
  const float g = gamma[f] * is, bb = beta[f];
  for (int b = rg; b < B; b += BNR_RG) y[(long)b * ldy + f] = (x[(long)b * ldx + f] - mu) * g + bb;
}
__global__ __launch_bounds__(512) void bnrows_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx, int B, int F,
                                                         const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                         const float* __restrict__ save_invstd, float* __restrict__ dx, long lddx,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[BNR_RG * BNR_COLS];
  const int cl = threadIdx.x & (BNR_COLS - 1), rg = threadIdx.x / BNR_COLS;
  const int f = blockIdx.x * BNR_COLS + cl;
  const bool ok = f < F;
  const float mu = ok ? save_mean[f] : 0.f, is = ok ? save_invstd[f] : 0.f;
  float sg = 0.f, sb = 0.f;
  if (ok) for (int b = rg; b < B; b += BNR_RG) { const float d = dy[(long)b * lddy + f]; sb += d; sg = fmaf(d, (x[(long)b * ldx + f] - mu) * is, sg); }
  sb = bnr_colsum(sb, red, cl, rg);
  sg = bnr_colsum(sg, red, cl, rg);
  if (!ok) return;
  if (rg == 0) { dgamma[f] = sg; dbeta[f] = sb; }
  const float k = gamma[f] * is / (float)B;
  for (int b = rg; b < B; b += BNR_RG) {
    const float xh = (x[(long)b * ldx + f] - mu) * is;
    dx[(long)b * lddx + f] = k * ((float)B * dy[(long)b * lddy + f] - sb - xh * sg);
  }
}

extern "C" {

int vdk_layernorm_fwd(const float* x, int64_t ldx, int32_t T, int32_t C, const float* gamma, const float* beta, float eps,
                      void* y, int64_t ldy, int32_t y_dtype, float* mean, float* rstd, void* stream) {
  if (!x || !gamma || !beta || !y || T < 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3))
    return vdk_fail(VDK_EINVAL, "vdk_layernorm_fwd: bad argument (C, ldx, ldy % 4 == 0)");
  if (T == 0) return VDK_OK;
  if (C > 4096) return vdk_fail(VDK_EUNSUPPORTED, "vdk_layernorm_fwd: C <= 4096");
  dim3 grid((unsigned)((T + 3) / 4));
#define LNF(BF, MJ) do { if (f16) hipLaunchKernelGGL((ln_fwd_kernel<true, MJ, false, VDK_OPF_F16>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)T, (int)C, gamma, beta, eps, y, (long)ldy, mean, rstd); \
                          else hipLaunchKernelGGL((ln_fwd_kernel<BF, MJ>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)T, (int)C, gamma, beta, eps, y, (long)ldy, mean, rstd); } while (0)
  if (y_dtype != VDK_BF16 && y_dtype != VDK_F32 && y_dtype != VDK_F16) return vdk_fail(VDK_EINVAL, "vdk_layernorm_fwd: bad y_dtype");
  const bool bf = y_dtype == VDK_BF16, f16 = y_dtype == VDK_F16;
  if (C <= 128) {
    const dim3 g8((unsigned)((T + 7) / 8));
    if (f16) hipLaunchKernelGGL((ln_fwd_narrow_kernel<true, VDK_OPF_F16>), g8, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)T, (int)C, gamma, beta, eps, y, (long)ldy, mean, rstd);
    else if (bf) hipLaunchKernelGGL((ln_fwd_narrow_kernel<true>), g8, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)T, (int)C, gamma, beta, eps, y, (long)ldy, mean, rstd);
    else hipLaunchKernelGGL((ln_fwd_narrow_kernel<false>), g8, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)T, (int)C, gamma, beta, eps, y, (long)ldy, mean, rstd);
  }
  else if (C <= 256) { if (bf) LNF(true, 1); else LNF(false, 1); }
  else if (C <= 1024) { if (bf) LNF(true, 4); else LNF(false, 4); }
  else { if (bf) LNF(true, 16); else LNF(false, 16); }
#undef LNF
  return vdk_check_launch("vdk_layernorm_fwd");
}

// vdk_layernorm_fwd with bf16 y, plus the fp8 copy of y (out8 [T, ldo8] bytes) and its amax: see LnQ8.  128 < C <= 1024 (the ViT widths), ldo8 % 4 == 0
int vdk_layernorm_fwd_q8(const float* x, int64_t ldx, int32_t T, int32_t C, const float* gamma, const float* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd,
                         void* out8, int64_t ldo8, int32_t out_fmt, const float* out_scale, float* out_amax, void* stream) {
  if (!x || !gamma || !beta || !y || !out8 || T < 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3) || (ldo8 & 3) || (out_fmt != 0 && out_fmt != 1))
    return vdk_fail(VDK_EINVAL, "vdk_layernorm_fwd_q8: bad argument (C, ldx, ldy, ldo8 % 4 == 0)");
  if (C <= 128 || C > 1024) return vdk_fail(VDK_EUNSUPPORTED, "vdk_layernorm_fwd_q8: 128 < C <= 1024");
  if (T == 0) return VDK_OK;
  const LnQ8 q8 = {(unsigned char*)out8, (long)ldo8, out_scale, out_amax, (int)out_fmt};
  unsigned nblk = (unsigned)((T + 3) / 4); if (nblk > 1024) nblk = 1024;      // 4 blocks per CU: each wave walks over T / 4096 rows and publishes its amax once
  if (C <= 256)
    hipLaunchKernelGGL((ln_fwd_kernel<true, 1, true>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)T, (int)C, gamma, beta, eps, y, (long)ldy,
                       mean, rstd, q8);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<true, 4, true>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)T, (int)C, gamma, beta, eps, y, (long)ldy,
                       mean, rstd, q8);
  return vdk_check_launch("vdk_layernorm_fwd_q8");
}

int vdk_reduce_rows_f32(const float* in, int64_t ld, int32_t S, int64_t n, float* out, float scale, void* stream) {
  if (!in || !out || S < 0 || n < 0) return vdk_fail(VDK_EINVAL, "vdk_reduce_rows_f32: bad argument");
  if (n == 0) return VDK_OK;
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((unsigned)((n + 63) / 64)), dim3(512), 0, (hipStream_t)stream, in, (long)ld,
                     (int)S, (long)n, out, scale);
  return vdk_check_launch("vdk_reduce_rows_f32");
}

static inline int ln_bwd_blocks(int T, int C) {
  int nb = (T + 15) / 16;          // >= 16 rows per block (4 per wave); the kernel is HBM-bound and what streams is waves in flight: 25 088 rows (ViT-B bs 128, Swin-B stage 3)
                                   // as 64-row blocks were 392 blocks = 1.5 per CU of the 4 that fit (68 us against a 22 us byte floor)
  // never more blocks than are resident at once: C <= 768 runs 4 blocks per CU (<= 128 VGPRs), wider rows 3 (136 VGPRs) -- T / 64 = 1152 blocks of ViT-L/14 at 336 were 1.5
  // rounds on 768 slots, i.e. the time of two; one round of fatter blocks costs 1.5
  const int slots = C <= 768 ? 1024 : 768;
  if (nb > slots) nb = slots;
  if (nb < 1) nb = 1;
  const int rpb = (T + nb - 1) / nb;
  nb = (T + rpb - 1) / rpb;        // no empty blocks: their zero partial rows would only lengthen the reduction (25 088 rows: 1004 blocks of 25, not 1024)
  return nb;
}
int vdk_layernorm_bwd_workspace_bytes(int32_t T, int32_t C, size_t* bytes) {
  if (!bytes || T < 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd_workspace_bytes: bad argument");
  *bytes = (size_t)3 * ln_bwd_blocks(T, C) * C * 4;      // [blocks][dgamma | dbeta] + [blocks][column sums of the bf16 output] (in-library by-product)
  return VDK_OK;
}
// dy: bf16 or f32 [T, lddy]; x f32 rows (ldx); dres optional f32 residual-stream gradient added to dx;
// outputs dx (f32, optional), dxb (bf16 copy, optional), dgamma/dbeta [C] (overwritten).
static int ln_bwd_impl(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* mean,
                       const float* rstd, const float* gamma, const float* dres, int64_t lddres, int32_t T, int32_t C, float* dx,
                       int64_t lddx, void* dxb, int64_t lddxb, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                       void* stream_, VdkReduceJob* deferred, float* dxb_colsum = nullptr, VdkReduceJob* deferred2 = nullptr, const LnQ8* q8 = nullptr, int opf = 0,
                       const float* dy_scale = nullptr, const float* dxb_rs = nullptr, int dxb_rps = 1, const void* dres16 = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  if (dres16) {      // the 16-bit residual-gradient stream of the ViT engine's fp16 mode: C <= 768, 16-bit dy, the column-sum form (one instantiation)
    if (!(dy_dtype == VDK_F16) || !dxb_colsum || C > 1024 || C <= 512 || q8 || dy_scale || dxb_rs) return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd: dres16 serves the fp16 column-sum form at 512 < C <= 1024");
  }
  if (dy_dtype == VDK_F16) opf = VDK_OPF_F16;          // (an fp32 dy with an fp16 dxb: opf passed by the in-library caller)
  if (dy_dtype != VDK_BF16 && dy_dtype != VDK_F32 && dy_dtype != VDK_F16) return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd: bad dy_dtype");
  if (opf && dy_dtype == VDK_BF16) return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd: bf16 dy with an fp16 output copy");
  if (!dy || !x || !mean || !rstd || !gamma || !dgamma || !dbeta || T <= 0 || C <= 0 || (C & 3) || C > 4096)
    return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd: bad argument (C % 4 == 0, C <= 4096)");
  const int nb = ln_bwd_blocks(T, C);
  const bool ocs = dxb_colsum != nullptr;
  if (ocs && (!dxb || !deferred2 || C > 1024)) return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd: the output column sums need dxb, a job slot and C <= 1024");
  size_t need = (size_t)(ocs ? 3 : 2) * nb * C * 4;
  if (!ws || ws_bytes < need) return vdk_fail(VDK_EWORKSPACE, "vdk_layernorm_bwd: workspace too small");
  float* pg = (float*)ws; float* pb = pg + C;          // rows of 2C: [dgamma partial | dbeta partial]
  float* po = pg + (size_t)2 * nb * C;                 // rows of C: column sums of dxb
  const int rpb = (T + nb - 1) / nb;
  const bool bf = dy_dtype != VDK_F32;
#define LNB(MJ, BF, OC, NR, HF) do { if (opf) hipLaunchKernelGGL((ln_bwd_kernel<MJ, BF, OC, false, VDK_OPF_F16, NR, HF>), dim3((unsigned)nb), dim3(256), 0, stream, dy, (long)lddy, x, (long)ldx, \
                                           mean, rstd, gamma, dres, (long)lddres, (int)T, (int)C, rpb, dx, (long)lddx, (bf16_t*)dxb, (long)lddxb, pg, pb, po, LnQ8(), dy_scale, dxb_rs, dxb_rps); \
                             else hipLaunchKernelGGL((ln_bwd_kernel<MJ, BF, OC, false, VDK_OPF_BF16, NR, HF>), dim3((unsigned)nb), dim3(256), 0, stream, dy, (long)lddy, x, (long)ldx, \
                                           mean, rstd, gamma, dres, (long)lddres, (int)T, (int)C, rpb, dx, (long)lddx, (bf16_t*)dxb, (long)lddxb, pg, pb, po, LnQ8(), dy_scale, dxb_rs, dxb_rps); } while (0)
#define LNB2(MJ, NR, HF) do { if (ocs) { if (bf) LNB(MJ, true, true, NR, HF); else LNB(MJ, false, true, NR, HF); } \
                              else { if (bf) LNB(MJ, true, false, NR, HF); else LNB(MJ, false, false, NR, HF); } } while (0)
  // MAXJ = 3 (C <= 768, ViT-B) keeps the kernel at <= 128 VGPRs = 4 blocks per CU; the narrower rows of the Swin / ConvNeXt stages take two rows per wave and pass
  // (C <= 128: four, a row being half a wave)
  if (q8) {
    if (!ocs || !bf || opf || dy_scale || dxb_rs || !q8->out || (q8->ld & 3)) return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd: the fp8 copy needs bf16 dy, the column-sum form and ld % 4 == 0");
#define LNBQ(MJ) hipLaunchKernelGGL((ln_bwd_kernel<MJ, true, true, true>), dim3((unsigned)nb), dim3(256), 0, stream, dy, (long)lddy, x, (long)ldx, \
                                    mean, rstd, gamma, dres, (long)lddres, (int)T, (int)C, rpb, dx, (long)lddx, (bf16_t*)dxb, (long)lddxb, pg, pb, po, *q8)
    if (C <= 768) LNBQ(3); else LNBQ(4);
#undef LNBQ
  }
  else if (dres16) {
#define LNB16(MJ) hipLaunchKernelGGL((ln_bwd_kernel<MJ, true, true, false, VDK_OPF_F16, 1, false, true>), dim3((unsigned)nb), dim3(256), 0, stream, dy, (long)lddy, x, (long)ldx, mean, rstd, \
                       gamma, (const float*)nullptr, (long)lddres, (int)T, (int)C, rpb, dx, (long)lddx, (bf16_t*)dxb, (long)lddxb, pg, pb, po, LnQ8(), (const float*)nullptr,          \
                       (const float*)nullptr, 1, (const bf16_t*)dres16)
    if (C <= 768) LNB16(3); else LNB16(4);
#undef LNB16
  }
  else if (C <= 128) LNB2(1, 2, true);
  else if (C <= 256) LNB2(1, 2, false);
  else if (C <= 512) LNB2(2, 2, false);
  else if (C <= 768) LNB2(3, 1, false);
  else if (C <= 1024) LNB2(4, 1, false);
  else { if (bf) LNB(16, true, false, 1, false); else LNB(16, false, false, 1, false); }
#undef LNB2
#undef LNB
  if (ocs) *deferred2 = VdkReduceJob{po, (long)C, nb, (long)C, dxb_colsum, 1.0f};
  if (deferred && dbeta == dgamma + C) {      // the caller batches the partial-sum reduction with others (vdk_reduce_rows_batch)
    *deferred = VdkReduceJob{pg, (long)(2 * C), nb, (long)(2 * C), dgamma, 1.0f};
    return vdk_check_launch("vdk_layernorm_bwd");
  }
  if (deferred) deferred->in = nullptr;
  if (dbeta == dgamma + C) {        // norm.weight / norm.bias of the flat gradient buffer (C % 64 == 0): one launch
    hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(512), 0, stream, (const float*)pg, (long)(2 * C), nb,
                       (long)(2 * C), dgamma, 1.0f);
  } else {
    hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((unsigned)((C + 63) / 64)), dim3(512), 0, stream, (const float*)pg, (long)(2 * C), nb,
                       (long)C, dgamma, 1.0f);
    hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((unsigned)((C + 63) / 64)), dim3(512), 0, stream, (const float*)pb, (long)(2 * C), nb,
                       (long)C, dbeta, 1.0f);
  }
  return vdk_check_launch("vdk_layernorm_bwd");
}
int vdk_layernorm_bwd(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* mean,
                      const float* rstd, const float* gamma, const float* dres, int64_t lddres, int32_t T, int32_t C, float* dx,
                      int64_t lddx, void* dxb, int64_t lddxb, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                      void* stream_) {
  return ln_bwd_impl(dy, lddy, dy_dtype, x, ldx, mean, rstd, gamma, dres, lddres, T, C, dx, lddx, dxb, lddxb, dgamma, dbeta, ws, ws_bytes, stream_, nullptr);
}

// row splits of the column-sum pass: enough workgroups to stream at HBM rate whatever the width -- a [1.6 M, 128] gradient (ConvNeXt stage 0) is ONE column block, and 128
// splits of it were 128 workgroups on 256 CUs: 1 TB/s.  About 1024 workgroups in all (four 16-byte loads in flight per thread), at least 64 rows each.
static inline int colsum_splits(int T, int N) {
  const int bx = (N / 8 + 31) / 32;
  int target = 1024 / (bx > 0 ? bx : 1); if (target < 128) target = 128;
  int s = (T + 63) / 64; if (s > target) s = target; if (s < 1) s = 1;
  return s;
}
int vdk_colsum_bf16_workspace_bytes(int32_t T, int32_t N, size_t* bytes) {
  if (!bytes || T < 0 || N <= 0) return vdk_fail(VDK_EINVAL, "vdk_colsum_bf16_workspace_bytes: bad argument");
  *bytes = (size_t)colsum_splits(T, N) * N * 4;
  return VDK_OK;
}
// out[c] = sum_r in[r][c]  (bias gradient of a Linear: column sum of dY), N % 8 == 0, ld % 8 == 0
int vdk_colsum_bf16(const void* in, int64_t ld, int32_t T, int32_t N, float* out, void* ws, size_t ws_bytes, void* stream_) {
  return vdk_colsum_16(in, ld, T, N, out, ws, ws_bytes, VDK_OPF_BF16, stream_);
}
}  // extern "C"
int vdk_colsum_16(const void* in, int64_t ld, int32_t T, int32_t N, float* out, void* ws, size_t ws_bytes, int opf, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || !out || T <= 0 || N <= 0 || (N & 7) || (ld & 7)) return vdk_fail(VDK_EINVAL, "vdk_colsum_bf16: bad argument (N, ld % 8 == 0)");
  const int S = colsum_splits(T, N);
  if (!ws || ws_bytes < (size_t)S * N * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_colsum_bf16: workspace too small");
  const int rps = (T + S - 1) / S;
  if (opf) hipLaunchKernelGGL((colsum_bf16_partial_kernel<false, VDK_OPF_F16>), dim3((unsigned)((N / 8 + 31) / 32), (unsigned)S), dim3(256), 0, stream,
                              (const bf16_t*)in, (long)ld, (int)T, (int)N, rps, (float*)ws, LnQ8());
  else
  hipLaunchKernelGGL(colsum_bf16_partial_kernel<false>, dim3((unsigned)((N / 8 + 31) / 32), (unsigned)S), dim3(256), 0, stream,
                     (const bf16_t*)in, (long)ld, (int)T, (int)N, rps, (float*)ws, LnQ8());
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((unsigned)((N + 63) / 64)), dim3(512), 0, stream, (const float*)ws, (long)N, S,
                     (long)N, out, 1.0f);
  return vdk_check_launch("vdk_colsum_bf16");
}
extern "C" {

int vdk_batchnorm1d_fwd(const float* x, int64_t ldx, int32_t B, int32_t F, const float* gamma, const float* beta, float eps, float momentum,
                        int32_t training, float* running_mean, float* running_var, float* y, int64_t ldy, float* save_mean, float* save_invstd,
                        void* stream) {
  if (!x || !gamma || !beta || !y || B <= 0 || F <= 0 || (!training && (!running_mean || !running_var)))
    return vdk_fail(VDK_EINVAL, "vdk_batchnorm1d_fwd: bad argument");
  if (B >= 256)   // many samples (BatchNorm2d on NHWC rows, large batches): row-parallel kernel
    hipLaunchKernelGGL(bnrows_fwd_kernel, dim3((unsigned)((F + BNR_COLS - 1) / BNR_COLS)), dim3(512), 0, (hipStream_t)stream, x, (long)ldx, (int)B, (int)F, gamma,
                       beta, eps, momentum, (int)training, running_mean, running_var, y, (long)ldy, save_mean, save_invstd);
  else
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (int)B, (int)F, gamma, beta, eps,
                       momentum, (int)training, running_mean, running_var, y, (long)ldy, save_mean, save_invstd);
  return vdk_check_launch("vdk_batchnorm1d_fwd");
}
int vdk_batchnorm1d_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int32_t B, int32_t F, const float* gamma, const float* save_mean,
                        const float* save_invstd, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* stream) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || B <= 0 || F <= 0)
    return vdk_fail(VDK_EINVAL, "vdk_batchnorm1d_bwd: bad argument");
  if (B >= 256)
    hipLaunchKernelGGL(bnrows_bwd_kernel, dim3((unsigned)((F + BNR_COLS - 1) / BNR_COLS)), dim3(512), 0, (hipStream_t)stream, dy, (long)lddy, x, (long)ldx, (int)B,
                       (int)F, gamma, save_mean, save_invstd, dx, (long)lddx, dgamma, dbeta);
  else
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, (long)lddy, x, (long)ldx, (int)B, (int)F,
                       gamma, save_mean, save_invstd, dx, (long)lddx, dgamma, dbeta);
  return vdk_check_launch("vdk_batchnorm1d_bwd");
}

int vdk_softmax_ce_amp(const float* logits, int64_t ldl, int32_t B, int32_t C, const int64_t* ya, const int64_t* yb, float lam,
                       float label_smoothing, float grad_scale, const float* loss_scale, float* loss_rows, void* dlogits16, int64_t lddl, int32_t dl_dtype,
                       float* dlogits_f32, int64_t lddf, void* stream) {
  if (!logits || !ya || B <= 0 || C <= 0 || (dlogits16 && lddl < C) || (dl_dtype != VDK_BF16 && dl_dtype != VDK_F16)) return vdk_fail(VDK_EINVAL, "vdk_softmax_ce: bad argument");
  if (dl_dtype == VDK_F16)
    hipLaunchKernelGGL(softmax_ce_kernel<VDK_OPF_F16>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl, (int)C, (const long long*)ya, (const long long*)yb, lam,
                       label_smoothing, grad_scale, loss_rows, (bf16_t*)dlogits16, (long)(dlogits16 ? lddl : 0), dlogits_f32, (long)lddf, loss_scale);
  else
    hipLaunchKernelGGL(softmax_ce_kernel<0>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl, (int)C, (const long long*)ya, (const long long*)yb, lam,
                       label_smoothing, grad_scale, loss_rows, (bf16_t*)dlogits16, (long)(dlogits16 ? lddl : 0), dlogits_f32, (long)lddf, loss_scale);
  return vdk_check_launch("vdk_softmax_ce");
}
int vdk_softmax_ce(const float* logits, int64_t ldl, int32_t B, int32_t C, const int64_t* ya, const int64_t* yb, float lam,
                   float label_smoothing, float grad_scale, float* loss_rows, void* dlogits_bf16, int64_t lddl,
                   float* dlogits_f32, int64_t lddf, void* stream) {
  return vdk_softmax_ce_amp(logits, ldl, B, C, ya, yb, lam, label_smoothing, grad_scale, nullptr, loss_rows, dlogits_bf16, lddl, VDK_BF16, dlogits_f32, lddf, stream);
}

int vdk_bce_logits_amp(const float* logits, int64_t ldl, const float* targets, int64_t ldt, int32_t B, int32_t C, float grad_scale, const float* loss_scale,
                       float focal_gamma, float focal_alpha, float* loss_rows, void* dlogits16, int64_t lddl, int32_t dl_dtype, float* dlogits_f32, int64_t lddf,
                       void* stream) {
  if (!logits || !targets || B <= 0 || C <= 0 || (dlogits16 && lddl < C) || (dl_dtype != VDK_BF16 && dl_dtype != VDK_F16)) return vdk_fail(VDK_EINVAL, "vdk_bce_logits: bad argument");
  if (dl_dtype == VDK_F16)
    hipLaunchKernelGGL(bce_logits_kernel<VDK_OPF_F16>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl, targets, (long)ldt, (int)C, grad_scale, focal_gamma,
                       focal_alpha, loss_rows, (bf16_t*)dlogits16, (long)(dlogits16 ? lddl : 0), dlogits_f32, (long)lddf, loss_scale);
  else
    hipLaunchKernelGGL(bce_logits_kernel<0>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl, targets, (long)ldt, (int)C, grad_scale, focal_gamma,
                       focal_alpha, loss_rows, (bf16_t*)dlogits16, (long)(dlogits16 ? lddl : 0), dlogits_f32, (long)lddf, loss_scale);
  return vdk_check_launch("vdk_bce_logits");
}
int vdk_bce_logits(const float* logits, int64_t ldl, const float* targets, int64_t ldt, int32_t B, int32_t C, float grad_scale,
                   float focal_gamma, float focal_alpha, float* loss_rows, void* dlogits_bf16, int64_t lddl, float* dlogits_f32, int64_t lddf,
                   void* stream) {
  return vdk_bce_logits_amp(logits, ldl, targets, ldt, B, C, grad_scale, nullptr, focal_gamma, focal_alpha, loss_rows, dlogits_bf16, lddl, VDK_BF16, dlogits_f32, lddf, stream);
}

}  // extern "C"

// ---- in-library helpers (C++ linkage, declared in vdk_host.h) ---------------------------------------------------------------------------------
// LayerNorm backward whose dgamma | dbeta partial-sum reduction is left to the caller: *job describes it (job->in == NULL if it was done here after all)
int vdk_layernorm_bwd_deferred(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                               const float* dres, int64_t lddres, int32_t T, int32_t C, float* dx, int64_t lddx, void* dxb, int64_t lddxb, float* dgamma, float* dbeta,
                               void* ws, size_t ws_bytes, void* stream, VdkReduceJob* job, float* dxb_colsum, VdkReduceJob* job2, const LnQ8* dxb_q8, int opf, const float* dy_scale, const float* dxb_rs, int dxb_rps,
                               const void* dres16) {
  if (dxb_rs && (!dxb || dxb_rps <= 0)) return vdk_fail(VDK_EINVAL, "vdk_layernorm_bwd: dxb_rs needs dxb and dxb_rps > 0");
  return ln_bwd_impl(dy, lddy, dy_dtype, x, ldx, mean, rstd, gamma, dres, lddres, T, C, dx, lddx, dxb, lddxb, dgamma, dbeta, ws, ws_bytes, stream, job, dxb_colsum, job2, dxb_q8, opf, dy_scale,
                     dxb_rs, dxb_rps, dres16);
}
// vdk_colsum_bf16 whose final reduction over the row splits is left to the caller (*job describes it)
int vdk_colsum_bf16_deferred(const void* in, int64_t ld, int32_t T, int32_t N, float* out, void* ws, size_t ws_bytes, void* stream_, VdkReduceJob* job, const LnQ8* q8, int opf) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || !out || !job || T <= 0 || N <= 0 || (N & 7) || (ld & 7) || (q8 && (!q8->out || (q8->ld & 7)))) return vdk_fail(VDK_EINVAL, "vdk_colsum_bf16: bad argument (N, ld % 8 == 0)");
  const int S = colsum_splits(T, N);
  if (!ws || ws_bytes < (size_t)S * N * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_colsum_bf16: workspace too small");
  const int rps = (T + S - 1) / S;
  const dim3 grid((unsigned)((N / 8 + 31) / 32), (unsigned)S);
  if (q8 && opf) return vdk_fail(VDK_EINVAL, "vdk_colsum_bf16: the fp8 copy goes with bf16 tensors");
  if (opf) hipLaunchKernelGGL((colsum_bf16_partial_kernel<false, VDK_OPF_F16>), grid, dim3(256), 0, stream, (const bf16_t*)in, (long)ld, (int)T, (int)N, rps, (float*)ws, LnQ8());
  else if (q8) hipLaunchKernelGGL(colsum_bf16_partial_kernel<true>, grid, dim3(256), 0, stream, (const bf16_t*)in, (long)ld, (int)T, (int)N, rps, (float*)ws, *q8);
  else hipLaunchKernelGGL(colsum_bf16_partial_kernel<false>, grid, dim3(256), 0, stream, (const bf16_t*)in, (long)ld, (int)T, (int)N, rps, (float*)ws, LnQ8());
  *job = VdkReduceJob{(float*)ws, (long)N, S, (long)N, out, 1.0f};
  return vdk_check_launch("vdk_colsum_bf16");
}
int vdk_reduce_rows_batch(const VdkReduceJob* jobs, int n, void* stream) {
  if (n < 0 || (n > 0 && !jobs)) return vdk_fail(VDK_EINVAL, "vdk_reduce_rows_batch: bad argument");
  for (int i0 = 0; i0 < n; i0 += 8) {
    ReduceBatch b;
    b.n = 0;
    int maxS = 0;
    for (int i = i0; i < n && i < i0 + 8; ++i) if (jobs[i].in && jobs[i].n > 0 && jobs[i].S > maxS) maxS = jobs[i].S;
    const bool tall = maxS >= 4 * RR_FOLD && maxS < 32 * RR_FOLD;      // 256 .. 2047 partial rows: the 32-row-group form; beyond: fold first (below)
    const int cw = tall ? 16 : 64;
    int blocks = 0;
    for (int i = i0; i < n && i < i0 + 8; ++i) {
      if (!jobs[i].in || jobs[i].n <= 0) continue;
      b.first_block[b.n] = blocks;
      VdkReduceJob jb = jobs[i];
      if (!tall && jb.S >= 16 * RR_FOLD) {      // very tall: fold the partial rows (scratch of the caller) onto the first 64, in place
        hipLaunchKernelGGL(reduce_rows_fold_kernel, dim3((unsigned)((jb.n + 63) / 64), RR_FOLD), dim3(512), 0, (hipStream_t)stream, (float*)jb.in, (long)jb.ld, (int)jb.S, (long)jb.n);
        jb.S = RR_FOLD;
      }
      b.job[b.n++] = jb;
      blocks += (int)((jobs[i].n + cw - 1) / cw);
    }
    b.first_block[b.n] = blocks;
    if (blocks > 0) {
      if (tall) hipLaunchKernelGGL(reduce_rows_batch_tall_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, b);
      else hipLaunchKernelGGL(reduce_rows_batch_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, b);
    }
  }
  return vdk_check_launch("vdk_reduce_rows_batch");
}
/* tests: one job through vdk_reduce_rows_batch (the in-library batch reduction; `buf` is scratch: a tall job is folded in place) */
extern "C" int vdk_debug_reduce_rows_job(float* buf, int64_t ld, int32_t S, int64_t n, float* out, float scale, void* stream) {
  if (!buf || !out || S <= 0 || n <= 0 || ld < n) return vdk_fail(VDK_EINVAL, "vdk_debug_reduce_rows_job: bad argument");
  const VdkReduceJob jb = {buf, (long)ld, (int)S, (long)n, out, scale};
  return vdk_reduce_rows_batch(&jb, 1, stream);
}


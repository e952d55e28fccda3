// margin_head.hip — K11: margin-softmax heads of the faceX / CBIR training path, fused with the cross-entropy.
//   ArcFace     models/faceX/head/arcface.py:20-36     (normalise W columns + feature rows, cos = f^ W^, clamp, margin at target, x scale)
//   CircleLoss  models/faceX/head/circleloss.py:21-43  (detached alpha_p / alpha_n re-weighting)
//   MV-Softmax  models/faceX/head/mv_softmax.py:25-44  (hard-negative re-weighting, AM or arc margin on the target)
// followed by nn.CrossEntropyLoss (engine/procedure/train.py:196 `criterion(self.model(images, labels), labels)`).
//
// The reference materialises >= 6 B x C fp32 temporaries (kernel_norm, cos, sin, cos_m, where, index, output, log-softmax:
// 2 GB each at B=512, C=1M).  Here: one fp32 cos matrix (GEMM output) and one bf16 d(cos) matrix; margin, scale, softmax,
// loss and the Jacobian of the margin are evaluated per row on the fly.  The three GEMMs (cos = f^ W^, dW^ = f^T dcos,
// df^ = dcos W^T) run on the bf16 MFMA kernels of gemm.hip straight from the layouts the tensors already have
// (W is [feat_dim, num_class] row-major: B operand of the TN kernel forward, B operand of the NT kernel backward).
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_margin.h"

// ---- normalisations ------------------------------------------------------------------------------------------------
// W f32 [D, ldw] (C valid columns) -> inv[c] = 1 / max(||W[:, c]||, eps), and W^ as THREE bf16 planes stacked along the
// contraction dim, Wb [3D, ldb]: rows [0,D) = hi(W^), [D,2D) = hi(W^), [2D,3D) = lo(W^) with lo = bf16(W^ - hi).  Paired with the
// feature planes (hi, lo, hi) the cos GEMM accumulates hi*hi + lo*hi + hi*lo: fp32-class accuracy (~2^-16) on the bf16 MFMA
// pipe.  Columns C..Cp-1 are zero.  The backward GEMMs use the first plane only.
// OF: format of the planes -- bf16, or fp16 (VDK_OPF_F16: hi = fp16(v), lo = fp16(v - hi); the same three products, the operands of an fp16 head)
template <int OF = 0>
__global__ __launch_bounds__(256) void colnorm_fwd_kernel(const float* __restrict__ W, long ldw, int D, int C, int Cp, float eps,
                                                          float* __restrict__ inv, bf16_t* __restrict__ Wb, long ldb, int planes) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Cp) return;
  float s = 0.f;
  if (c < C) for (int d = 0; d < D; ++d) { float v = W[(long)d * ldw + c]; s = fmaf(v, v, s); }
  const float iv = c < C ? 1.0f / fmaxf(sqrtf(s), eps) : 0.f;
  if (c < C && inv) inv[c] = iv;
  for (int d = 0; d < D; ++d) {
    const float v = c < C ? W[(long)d * ldw + c] * iv : 0.f;
    const bf16_t h = f2op<OF>(v);
    const bf16_t l = f2op<OF>(v - op2f<OF>(h));
    Wb[(long)d * ldb + c] = h;
    if (planes == 3) { Wb[(long)(D + d) * ldb + c] = h; Wb[(long)(2 * D + d) * ldb + c] = l; }
  }
}
// dW[:, c] = inv[c] * (dW^[:, c] - W^[:, c] * <W^[:, c], dW^[:, c]>),  W^ = W * inv
__global__ __launch_bounds__(256) void colnorm_bwd_kernel(const float* __restrict__ W, long ldw, const float* __restrict__ inv,
                                                          const float* __restrict__ dWh, long ldg, int D, int C, float* __restrict__ dW,
                                                          long ldo) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float iv = inv[c];
  float dot = 0.f;
  for (int d = 0; d < D; ++d) dot = fmaf(W[(long)d * ldw + c] * iv, dWh[(long)d * ldg + c], dot);
  for (int d = 0; d < D; ++d) dW[(long)d * ldo + c] = iv * (dWh[(long)d * ldg + c] - W[(long)d * ldw + c] * iv * dot);
}
// Tiled forms (D <= 16 * 32, 16-byte aligned rows): a 512-thread block owns 64 columns x all D rows as 32 row groups x 16 column quads; every thread keeps
// its D / 32 rows x 4 columns in registers, so W (and dW^) are read from HBM exactly once with 16-byte loads, many in flight (the thread-per-column loops above
// issue one dependent 4-byte load at a time: 1.6 / 1.9 ms at C = 10^6 where 5 / 6 GB of traffic cost 1.0 / 1.2 ms).  planes: 3 = (hi, hi, lo), 1 = hi only.
template <int RPT, int OF = 0>
__global__ __launch_bounds__(512) void colnorm_fwd_tiled_kernel(const float* __restrict__ W, long ldw, int D, int C, int Cp, float eps, float* __restrict__ inv,
                                                                bf16_t* __restrict__ Wb, long ldb, int planes) {
  __shared__ float red[32][65];
  __shared__ float tot[64];
  const int tid = threadIdx.x, q = tid & 15, rg = tid >> 4;
  const int c0 = blockIdx.x * 64 + q * 4;
  f32x4 v[RPT];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int d = rg + 32 * i;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (d < D && c0 < C) {
      const float* src = W + (long)d * ldw + c0;
      if (c0 + 3 < C) x = *(const f32x4*)src;
      else { x[0] = src[0]; if (c0 + 1 < C) x[1] = src[1]; if (c0 + 2 < C) x[2] = src[2]; }
    }
    v[i] = x;
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = fmaf(x[e], x[e], s[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rg][q * 4 + e] = s[e];
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][tid];
    const int c = blockIdx.x * 64 + tid;
    const float iv = c < C ? 1.0f / fmaxf(sqrtf(t), eps) : 0.f;
    tot[tid] = iv;
    if (c < C && inv) inv[c] = iv;
  }
  __syncthreads();
  if (c0 >= Cp) return;
  float iv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) iv[e] = tot[q * 4 + e];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int d = rg + 32 * i;
    if (d >= D) continue;
    float n[4]; bf16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { n[e] = v[i][e] * iv[e]; h[e] = f2op<OF>(n[e]); l[e] = f2op<OF>(n[e] - op2f<OF>(h[e])); }
    const u32x2 hv = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    *(u32x2*)(Wb + (long)d * ldb + c0) = hv;
    if (planes == 3) {
      *(u32x2*)(Wb + (long)(D + d) * ldb + c0) = hv;
      *(u32x2*)(Wb + (long)(2 * D + d) * ldb + c0) = (u32x2){(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
    }
  }
}
template <int RPT>
__global__ __launch_bounds__(512) void colnorm_bwd_tiled_kernel(const float* __restrict__ W, long ldw, const float* __restrict__ inv, const float* __restrict__ dWh,
                                                                long ldg, int D, int C, float* __restrict__ dW, long ldo) {
  __shared__ float red[32][65];
  __shared__ float tot[64];
  const int tid = threadIdx.x, q = tid & 15, rg = tid >> 4;
  const int c0 = blockIdx.x * 64 + q * 4;          // C % 4 == 0 here (launcher): a quad is all valid or all beyond C
  const bool ok = c0 < C;
  f32x4 iv = {0.f, 0.f, 0.f, 0.f};
  if (ok) iv = *(const f32x4*)(inv + c0);
  f32x4 wv[RPT], gv[RPT];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int d = rg + 32 * i;
    f32x4 x = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
    if (d < D && ok) { x = *(const f32x4*)(W + (long)d * ldw + c0); g = *(const f32x4*)(dWh + (long)d * ldg + c0); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[e] *= iv[e]; s[e] = fmaf(x[e], g[e], s[e]); }
    wv[i] = x; gv[i] = g;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rg][q * 4 + e] = s[e];
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][tid];
    tot[tid] = t;
  }
  __syncthreads();
  if (!ok) return;
  float dot[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) dot[e] = tot[q * 4 + e];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int d = rg + 32 * i;
    if (d >= D) continue;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = iv[e] * (gv[i][e] - wv[i][e] * dot[e]);
    *(f32x4*)(dW + (long)d * ldo + c0) = o;
  }
}
// f f32 [B, D] -> f^ as f32 [B, D], bf16 [Bp, D] (hi plane, backward operand) and the transposed split planes
// fbt [3D, Bp] = (hi, lo, hi) for the cos GEMM; inv[B]; rows B..Bp-1 zero.  one wave per row
template <int OF = 0>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ f, int B, int Bp, int D, float eps, float* __restrict__ fh,
                                                          bf16_t* __restrict__ fb, bf16_t* __restrict__ fbt, float* __restrict__ inv, int planes) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Bp) return;
  float s = 0.f;
  if (row < B) for (int d = lane; d < D; d += 64) { float v = f[(long)row * D + d]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  const float iv = row < B ? 1.0f / fmaxf(sqrtf(s), eps) : 0.f;
  if (lane == 0 && row < B) inv[row] = iv;
  for (int d = lane; d < D; d += 64) {
    const float v = row < B ? f[(long)row * D + d] * iv : 0.f;
    if (row < B) fh[(long)row * D + d] = v;
    const bf16_t h = f2op<OF>(v);
    const bf16_t l = f2op<OF>(v - op2f<OF>(h));
    fb[(long)row * D + d] = h;
    fbt[(long)d * Bp + row] = h;
    if (planes == 3) { fbt[(long)(D + d) * Bp + row] = l; fbt[(long)(2 * D + d) * Bp + row] = h; }
  }
}
// df = inv * (df^ - f^ <f^, df^>)
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ fh, const float* __restrict__ inv, const float* __restrict__ dfh,
                                                          long lddfh, int B, int D, float* __restrict__ df) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot = fmaf(fh[(long)row * D + d], dfh[(long)row * lddfh + d], dot);
  dot = wave_sum(dot);
  const float iv = inv[row];
  for (int d = lane; d < D; d += 64) df[(long)row * D + d] = iv * (dfh[(long)row * lddfh + d] - fh[(long)row * D + d] * dot);
}

// ---- margin + CE per row --------------------------------------------------------------------------------------------
// cos f32 [B, ldc]; one workgroup per row.  Outputs (all optional): logits f32 [B, ldl], loss_rows [B],
// dcos bf16 [Bp?, lddc] = gscale * (softmax - target) * jac (columns C..lddc-1 zeroed).
__global__ __launch_bounds__(256) void margin_ce_kernel(MarginP P, const float* __restrict__ cosv, long ldc, int C, const long long* __restrict__ y,
                                                        float label_smoothing, float gscale, float* __restrict__ logits, long ldl,
                                                        float* __restrict__ loss_rows, bf16_t* __restrict__ dcos, long lddc, float* __restrict__ dcos32,
                                                        int opf = 0 /* format of dcos */, const float* __restrict__ lscale = nullptr /* device loss scale multiplied into gscale */) {
  __shared__ float red[4];
  if (lscale) gscale *= lscale[0];
  const int row = blockIdx.x, tid = threadIdx.x;
  margin_row_params(P, row);
  const float* cr = cosv + (long)row * ldc;
  const int yt = (int)y[row];
  const RowCtx R = margin_row_ctx(P, cr[yt]);
  float mx = -3.0e38f, sm = 0.f;
  for (int c = tid; c < C; c += 256) {
    float lg, jc; margin_eval(P, R, cr[c], c == yt, lg, jc);
    if (logits) logits[(long)row * ldl + c] = lg;
    mx = fmaxf(mx, lg); sm += lg;
  }
  mx = block_max<4>(mx, red);
  if (!loss_rows && !dcos && !dcos32) return;
  sm = block_sum<4>(sm, red);
  float se = 0.f;
  for (int c = tid; c < C; c += 256) { float lg, jc; margin_eval(P, R, cr[c], c == yt, lg, jc); se += expf(lg - mx); }
  se = block_sum<4>(se, red);
  const float lse = mx + logf(se);
  if (tid == 0 && loss_rows) {
    float lg, jc; margin_eval(P, R, cr[yt], true, lg, jc);
    loss_rows[row] = lse - (1.0f - label_smoothing) * lg - label_smoothing * (sm / (float)C);
  }
  if (!dcos && !dcos32) return;
  const float inv = 1.0f / se, epsc = label_smoothing / (float)C;
  for (int c = tid; c < (int)lddc; c += 256) {
    float g = 0.f;
    if (c < C) {
      float lg, jc; margin_eval(P, R, cr[c], c == yt, lg, jc);
      g = expf(lg - mx) * inv - epsc;
      if (c == yt) g -= (1.0f - label_smoothing);
      g *= gscale * jc;
    }
    if (dcos32) dcos32[(long)row * lddc + c] = g;      // the fp32-class training mode keeps the logit gradient unrounded
    else dcos[(long)row * lddc + c] = opf ? f2op<VDK_OPF_F16>(g) : f2bf(g);
  }
}
// Training form (no logits output) for wide heads: the same arithmetic with 16-byte loads, four of them in flight per thread (a 4-byte strided loop keeps
// ~1 KB in flight per workgroup: 1 TB/s at C = 10^6), and an online softmax so that cos is read twice instead of three times.
#ifndef MCE_U
#define MCE_U 4   // (8 in flight measured the same: 107.8 ms per cfg3 step either way -- the kernel is bound by its arithmetic, not by bytes in flight)
#endif
// v_exp_f32 on x * log2(e): 2 instructions against libm expf's ~12 (range reduction + polynomial); relative error ~2^-22 on |x| < 90, i.e. inside the rounding of the
// softmax sums.  The narrow-head kernel above (golden-vector comparisons) keeps expf.
__device__ __forceinline__ float vexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// ARC (ArcFace without per-row margins -- cfg3's head): every entry but the row's target is s * clamp(cos), so the margin function is evaluated ONCE per row (the target's
// logit and jacobian) and an entry costs a clamp, a multiply and a select instead of the generic evaluation's branches; the same expressions, bit-identical results.
// OF: format of dcos (bf16 | fp16).  lscale: GradScaler's loss scale on the device, multiplied into gscale -- with fp16 the softmax gradient of a 10^6-way head
// (p ~ 1e-6 times s / B) is far below fp16's normal range unscaled; the loss value itself is never scaled.
template <bool ARC, int OF = 0>
__global__ __launch_bounds__(256) void margin_ce_vec_kernel(MarginP P, const float* __restrict__ cosv, long ldc, int C, const long long* __restrict__ y,
                                                            float label_smoothing, float gscale, float* __restrict__ loss_rows, bf16_t* __restrict__ dcos,
                                                            long lddc, const float* __restrict__ lscale = nullptr) {
  __shared__ float red[4];
  if (lscale) gscale *= lscale[0];
  const int row = blockIdx.x, tid = threadIdx.x;
  margin_row_params(P, row);
  const float* cr = cosv + (long)row * ldc;
  const int yt = (int)y[row];
  const RowCtx R = margin_row_ctx(P, cr[yt]);
  float lg_t = 0.f, jc_t = 0.f;
  if (ARC) margin_eval(P, R, cr[yt], true, lg_t, jc_t);
  auto ev = [&](float cv, bool tg, float& lg, float& jc) {
    if (ARC) {
      const float c = fminf(fmaxf(cv, -1.0f), 1.0f);
      lg = tg ? lg_t : P.s * c;
      jc = tg ? jc_t : (c == cv ? P.s : 0.0f);          // s * (1 inside the clamp's range, 0 outside)
    } else margin_eval(P, R, cv, tg, lg, jc);
  };
  const int C4 = C & ~3;
  float m = -3.0e38f, se = 0.f, sm = 0.f;
  auto visit = [&](float cv, int c) {
    float lg, jc; ev(cv, c == yt, lg, jc);
    sm += lg;
    if (lg > m) { se *= vexp(m - lg); m = lg; }
    se += vexp(lg - m);
  };
  // four logits per running-maximum update: 5 exponentials per 4 entries instead of 8 (the kernel is VALU-bound, not HBM-bound: ~40 lane-ops per entry and pass)
  auto visit4 = [&](const f32x4& cv, int c) {
    float lg[4], jc;
#pragma unroll
    for (int e = 0; e < 4; ++e) ev(cv[e], c + e == yt, lg[e], jc);
    sm += (lg[0] + lg[1]) + (lg[2] + lg[3]);
    const float m4 = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    if (m4 > m) { se *= vexp(m - m4); m = m4; }
    se += (vexp(lg[0] - m) + vexp(lg[1] - m)) + (vexp(lg[2] - m) + vexp(lg[3] - m));
  };
  for (int base = tid * 4; base < C4; base += 256 * 4 * MCE_U) {
    f32x4 v[MCE_U];
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) { const int c = base + u * 1024; if (c < C4) v[u] = *(const f32x4*)(cr + c); }
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) {
      const int c = base + u * 1024;
      if (c < C4) visit4(v[u], c);
    }
  }
  for (int c = C4 + tid; c < C; c += 256) visit(cr[c], c);
  const float mx = block_max<4>(m, red);
  se = block_sum<4>(se * vexp(m - mx), red);
  sm = block_sum<4>(sm, red);
  if (tid == 0 && loss_rows) {
    float lg, jc; margin_eval(P, R, cr[yt], true, lg, jc);
    loss_rows[row] = mx + logf(se) - (1.0f - label_smoothing) * lg - label_smoothing * (sm / (float)C);
  }
  if (!dcos) return;
  const float inv = 1.0f / se, epsc = label_smoothing / (float)C;
  auto grad = [&](float cv, int c) -> float {
    float lg, jc; ev(cv, c == yt, lg, jc);
    float g = vexp(lg - mx) * inv - epsc;
    if (c == yt) g -= (1.0f - label_smoothing);
    return g * gscale * jc;
  };
  bf16_t* dr = dcos + (long)row * lddc;
  for (int base = tid * 4; base < C4; base += 256 * 4 * MCE_U) {
    f32x4 v[MCE_U];
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) { const int c = base + u * 1024; if (c < C4) v[u] = *(const f32x4*)(cr + c); }
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) {
      const int c = base + u * 1024;
      if (c < C4) *(u32x2*)(dr + c) = (u32x2){pack_op2<OF>(grad(v[u][0], c), grad(v[u][1], c + 1)), pack_op2<OF>(grad(v[u][2], c + 2), grad(v[u][3], c + 3))};
    }
  }
  for (int c = C4 + tid; c < (int)lddc; c += 256) dr[c] = f2op<OF>(c < C ? grad(cr[c], c) : 0.f);
}
// ---- class-sharded head (one shard of the weight per GPU: SURVEY 8(e), "class-sharded head") -----------------------------------------------------
// The softmax of a row spans all shards, so the fused kernel above splits into three local passes around two small all-reduces:
//   margin_target_cos:  gt[b] = cos[b, y_b - c_base] if this shard owns the target column else 0        (SUM over shards = the target cosine)
//   margin_stats:       stats[b] = (max_c logit, sum_c exp(logit - max), sum_c logit, target logit or 0)  (MAX / rescaled SUM over shards = the global LSE)
//   margin_grad:        dcos = gscale * (exp(logit - M) / S - eps / C_total - (1 - eps) [c is the target]) * jac     with the GLOBAL M, S
// cos is this shard's [B, Cloc] block (global column = c_base + c); MV-Softmax needs the global target cosine gt for every column's margin.
__global__ __launch_bounds__(256) void margin_target_cos_kernel(const float* __restrict__ cosv, long ldc, int B, int Cloc, long c_base, const long long* __restrict__ y,
                                                                float* __restrict__ gt) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const long t = (long)y[b] - c_base;
  gt[b] = (t >= 0 && t < Cloc) ? cosv[(long)b * ldc + t] : 0.f;
}
__global__ __launch_bounds__(256) void margin_stats_kernel(MarginP P, const float* __restrict__ cosv, long ldc, int Cloc, long c_base, const long long* __restrict__ y,
                                                           const float* __restrict__ gt, float* __restrict__ stats) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  margin_row_params(P, row);
  const float* cr = cosv + (long)row * ldc;
  const long yt = (long)y[row] - c_base;                 // local target column (may lie outside this shard)
  const RowCtx R = margin_row_ctx(P, gt[row]);
  float m = -3.0e38f, se = 0.f, sm = 0.f, tl = 0.f;
  auto visit = [&](float cv, int c) {
    float lg, jc; margin_eval(P, R, cv, c == yt, lg, jc);
    sm += lg;
    if (c == yt) tl = lg;
    if (lg > m) { se *= expf(m - lg); m = lg; }
    se += expf(lg - m);
  };
  auto visit4 = [&](const f32x4& cv, int c) {
    float lg[4], jc;
#pragma unroll
    for (int e = 0; e < 4; ++e) { margin_eval(P, R, cv[e], c + e == yt, lg[e], jc); if (c + e == yt) tl = lg[e]; }
    sm += (lg[0] + lg[1]) + (lg[2] + lg[3]);
    const float m4 = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    if (m4 > m) { se *= expf(m - m4); m = m4; }
    se += (expf(lg[0] - m) + expf(lg[1] - m)) + (expf(lg[2] - m) + expf(lg[3] - m));
  };
  // 16-byte loads, four in flight per thread (see margin_ce_vec_kernel); scalar loop for unaligned blocks and the tail
  const int C4 = (((ldc & 3) == 0) && (((size_t)cosv & 15) == 0)) ? (Cloc & ~3) : 0;
  for (int base = tid * 4; base < C4; base += 256 * 4 * MCE_U) {
    f32x4 v[MCE_U];
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) { const int c = base + u * 1024; if (c < C4) v[u] = *(const f32x4*)(cr + c); }
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) {
      const int c = base + u * 1024;
      if (c < C4) visit4(v[u], c);
    }
  }
  for (int c = C4 + tid; c < Cloc; c += 256) visit(cr[c], c);
  const float mx = block_max<4>(m, red);
  se = block_sum<4>(se * expf(m - mx), red);
  sm = block_sum<4>(sm, red);
  tl = block_sum<4>(tl, red);
  if (tid == 0) { float* o = stats + (long)row * 4; o[0] = mx; o[1] = se; o[2] = sm; o[3] = tl; }
}
__global__ __launch_bounds__(256) void margin_grad_kernel(MarginP P, const float* __restrict__ cosv, long ldc, int Cloc, long c_base, long C_total,
                                                          const long long* __restrict__ y, const float* __restrict__ gt, const float* __restrict__ gmax,
                                                          const float* __restrict__ gsum, float label_smoothing, float gscale, bf16_t* __restrict__ dcos, long lddc) {
  const int row = blockIdx.x, tid = threadIdx.x;
  margin_row_params(P, row);
  const float* cr = cosv + (long)row * ldc;
  const long yt = (long)y[row] - c_base;
  const RowCtx R = margin_row_ctx(P, gt[row]);
  const float mx = gmax[row], inv = 1.0f / gsum[row], epsc = label_smoothing / (float)C_total;
  auto grad = [&](float cv, int c) -> float {
    float lg, jc; margin_eval(P, R, cv, c == yt, lg, jc);
    float g = expf(lg - mx) * inv - epsc;
    if (c == yt) g -= (1.0f - label_smoothing);
    return g * gscale * jc;
  };
  bf16_t* dr = dcos + (long)row * lddc;
  const bool al = ((ldc & 3) == 0) && (((size_t)cosv & 15) == 0) && ((lddc & 3) == 0) && (((size_t)dcos & 7) == 0);
  const int C4 = al ? (Cloc & ~3) : 0;
  for (int base = tid * 4; base < C4; base += 256 * 4 * MCE_U) {
    f32x4 v[MCE_U];
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) { const int c = base + u * 1024; if (c < C4) v[u] = *(const f32x4*)(cr + c); }
#pragma unroll
    for (int u = 0; u < MCE_U; ++u) {
      const int c = base + u * 1024;
      if (c < C4) *(u32x2*)(dr + c) = (u32x2){pack_bf2(grad(v[u][0], c), grad(v[u][1], c + 1)), pack_bf2(grad(v[u][2], c + 2), grad(v[u][3], c + 3))};
    }
  }
  for (int c = C4 + tid; c < (int)lddc; c += 256) dr[c] = f2bf(c < Cloc ? grad(cr[c], c) : 0.f);
}

// backward of the logits-returning form: dcos = dlogits * jac (bf16, padded columns zeroed)
__global__ __launch_bounds__(256) void margin_bwd_kernel(MarginP P, const float* __restrict__ cosv, long ldc, int C, const long long* __restrict__ y,
                                                         const float* __restrict__ dlogits, long lddl, bf16_t* __restrict__ dcos, long lddc) {
  const int row = blockIdx.x, tid = threadIdx.x;
  margin_row_params(P, row);
  const float* cr = cosv + (long)row * ldc;
  const int yt = (int)y[row];
  const RowCtx R = margin_row_ctx(P, cr[yt]);
  for (int c = tid; c < (int)lddc; c += 256) {
    float g = 0.f;
    if (c < C) { float lg, jc; margin_eval(P, R, cr[c], c == yt, lg, jc); g = dlogits[(long)row * lddl + c] * jc; }
    dcos[(long)row * lddc + c] = f2bf(g);
  }
}

// ---- the fused form's small kernels (the cos GEMM applies the head in its epilogue: vdk_margin_cos_pass in gemm.hip) ---------------------------------------------
// row statistics from the per-slice partials of pass 1: rowstat[b] = (max, 1 / sum exp(logit - max)), loss_rows[b] = lse - (1 - eps) logit_target - eps mean(logit)
__global__ __launch_bounds__(256) void margin_rowstat_kernel(const float* __restrict__ stats, long nslice, const float* __restrict__ tlogit, int C, float label_smoothing,
                                                             float* __restrict__ rowstat, float* __restrict__ loss_rows) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const f32x4* sr = (const f32x4*)stats + (long)row * nslice;
  float m = -3.0e38f, se = 0.f, sl = 0.f;
  for (long i = tid; i < nslice; i += 256) {
    const f32x4 v = sr[i];
    sl += v[2];
    if (v[0] > m) { se *= vdk_vexp(m - v[0]); m = v[0]; }
    se += v[1] * vdk_vexp(v[0] - m);
  }
  const float mx = block_max<4>(m, red);
  se = block_sum<4>(se * vdk_vexp(m - mx), red);
  sl = block_sum<4>(sl, red);
  if (tid == 0) {
    rowstat[2 * row] = mx; rowstat[2 * row + 1] = 1.0f / se;
    if (loss_rows) loss_rows[row] = mx + logf(se) - (1.0f - label_smoothing) * tlogit[row] - label_smoothing * (sl / (float)C);
  }
}
// gt[b] = sum_k fbt[k][b] * wb[k][y_b]: the target cosine from the very operands of the cos GEMM (bf16 planes, fp32 accumulation), so that MV-Softmax's thresholds see
// the same number the GEMM's tile holds for that column (up to summation order); one wave per row
__global__ __launch_bounds__(256) void margin_target_cos_direct_kernel(const bf16_t* __restrict__ fbt, long ld_f, const bf16_t* __restrict__ wb, long ld_w, int K, int B,
                                                                       const long long* __restrict__ y, float* __restrict__ gt) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const long yt = y[row];
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(bf2f(fbt[(long)k * ld_f + row]), bf2f(wb[(long)k * ld_w + yt]), s);
  s = wave_sum(s);
  if (lane == 0) gt[row] = s;
}

extern "C" {

int vdk_colnorm_fwd_dt(const float* W, int64_t ldw, int32_t D, int32_t C, int32_t Cp, float eps, float* inv, void* Wb, int64_t ldb, int32_t planes, int32_t dtype, void* stream) {
  if (!W || !Wb || D <= 0 || C <= 0 || Cp < C || (planes != 1 && planes != 3) || (dtype != VDK_BF16 && dtype != VDK_F16))
    return vdk_fail(VDK_EINVAL, "vdk_colnorm_fwd: bad argument (planes = 1 or 3, dtype VDK_BF16 | VDK_F16)");
  const bool tiled = D <= 512 && (ldw % 4 == 0) && (ldb % 4 == 0) && (Cp % 4 == 0) && (((uintptr_t)W | (uintptr_t)Wb) % 16 == 0);
  const bool f16 = dtype == VDK_F16;
  if (tiled) {
    const dim3 grid((unsigned)((Cp + 63) / 64));
#define CNF(R) do { if (f16) hipLaunchKernelGGL((colnorm_fwd_tiled_kernel<R, VDK_OPF_F16>), grid, dim3(512), 0, (hipStream_t)stream, W, (long)ldw, (int)D, (int)C, (int)Cp, eps, inv, (bf16_t*)Wb, (long)ldb, (int)planes); \
                    else hipLaunchKernelGGL((colnorm_fwd_tiled_kernel<R, 0>), grid, dim3(512), 0, (hipStream_t)stream, W, (long)ldw, (int)D, (int)C, (int)Cp, eps, inv, (bf16_t*)Wb, (long)ldb, (int)planes); } while (0)
    if (D <= 128) CNF(4); else if (D <= 256) CNF(8); else CNF(16);
#undef CNF
  } else if (f16) {
    hipLaunchKernelGGL(colnorm_fwd_kernel<VDK_OPF_F16>, dim3((unsigned)((Cp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, (long)ldw, (int)D, (int)C, (int)Cp, eps,
                       inv, (bf16_t*)Wb, (long)ldb, (int)planes);
  } else {
    hipLaunchKernelGGL(colnorm_fwd_kernel<0>, dim3((unsigned)((Cp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, (long)ldw, (int)D, (int)C, (int)Cp, eps,
                       inv, (bf16_t*)Wb, (long)ldb, (int)planes);
  }
  return vdk_check_launch("vdk_colnorm_fwd");
}
int vdk_colnorm_fwd(const float* W, int64_t ldw, int32_t D, int32_t C, int32_t Cp, float eps, float* inv, void* Wb, int64_t ldb, int32_t planes, void* stream) {
  return vdk_colnorm_fwd_dt(W, ldw, D, C, Cp, eps, inv, Wb, ldb, planes, VDK_BF16, stream);
}
int vdk_colnorm_bwd(const float* W, int64_t ldw, const float* inv, const float* dWh, int64_t ldg, int32_t D, int32_t C, float* dW, int64_t ldo,
                    void* stream) {
  if (!W || !inv || !dWh || !dW || D <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_colnorm_bwd: bad argument");
  const bool tiled = D <= 512 && (C % 4 == 0) && (ldw % 4 == 0) && (ldg % 4 == 0) && (ldo % 4 == 0) &&
                     (((uintptr_t)W | (uintptr_t)dWh | (uintptr_t)dW | (uintptr_t)inv) % 16 == 0);
  if (tiled) {
    const dim3 grid((unsigned)((C + 63) / 64));
#define CNB(R) hipLaunchKernelGGL((colnorm_bwd_tiled_kernel<R>), grid, dim3(512), 0, (hipStream_t)stream, W, (long)ldw, inv, dWh, (long)ldg, (int)D, (int)C, dW, (long)ldo)
    if (D <= 128) CNB(4); else if (D <= 256) CNB(8); else CNB(16);
#undef CNB
  } else {
    hipLaunchKernelGGL(colnorm_bwd_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, (long)ldw, inv, dWh, (long)ldg, (int)D,
                       (int)C, dW, (long)ldo);
  }
  return vdk_check_launch("vdk_colnorm_bwd");
}
int vdk_rownorm_fwd_dt(const float* f, int32_t B, int32_t Bp, int32_t D, float eps, float* fh, void* fb, void* fbt, float* inv, int32_t planes, int32_t dtype, void* stream) {
  if (!f || !fh || !fb || !fbt || !inv || B <= 0 || Bp < B || D <= 0 || (planes != 1 && planes != 3) || (dtype != VDK_BF16 && dtype != VDK_F16))
    return vdk_fail(VDK_EINVAL, "vdk_rownorm_fwd: bad argument");
  if (dtype == VDK_F16)
    hipLaunchKernelGGL(rownorm_fwd_kernel<VDK_OPF_F16>, dim3((unsigned)((Bp + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f, (int)B, (int)Bp, (int)D, eps, fh, (bf16_t*)fb,
                       (bf16_t*)fbt, inv, (int)planes);
  else
    hipLaunchKernelGGL(rownorm_fwd_kernel<0>, dim3((unsigned)((Bp + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f, (int)B, (int)Bp, (int)D, eps, fh, (bf16_t*)fb,
                       (bf16_t*)fbt, inv, (int)planes);
  return vdk_check_launch("vdk_rownorm_fwd");
}
int vdk_rownorm_fwd(const float* f, int32_t B, int32_t Bp, int32_t D, float eps, float* fh, void* fb, void* fbt, float* inv, int32_t planes, void* stream) {
  return vdk_rownorm_fwd_dt(f, B, Bp, D, eps, fh, fb, fbt, inv, planes, VDK_BF16, stream);
}
int vdk_rownorm_bwd(const float* fh, const float* inv, const float* dfh, int64_t lddfh, int32_t B, int32_t D, float* df, void* stream) {
  if (!fh || !inv || !dfh || !df || B <= 0 || D <= 0) return vdk_fail(VDK_EINVAL, "vdk_rownorm_bwd: bad argument");
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, fh, inv, dfh, (long)lddfh, (int)B, (int)D, df);
  return vdk_check_launch("vdk_rownorm_bwd");
}
int vdk_margin_rowstat(const float* stats, int64_t nslice, const float* tlogit, int32_t B, int32_t C, float label_smoothing, float* rowstat, float* loss_rows, void* stream) {
  if (!stats || !tlogit || !rowstat || B <= 0 || C <= 0 || nslice <= 0) return vdk_fail(VDK_EINVAL, "vdk_margin_rowstat: bad argument");
  hipLaunchKernelGGL(margin_rowstat_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, stats, (long)nslice, tlogit, (int)C, label_smoothing, rowstat, loss_rows);
  return vdk_check_launch("vdk_margin_rowstat");
}
int vdk_margin_target_cos_direct(const void* fbt, int64_t ld_f, const void* wb, int64_t ld_w, int32_t K, int32_t B, const int64_t* labels, float* gt, void* stream) {
  if (!fbt || !wb || !labels || !gt || B <= 0 || K <= 0) return vdk_fail(VDK_EINVAL, "vdk_margin_target_cos_direct: bad argument");
  hipLaunchKernelGGL(margin_target_cos_direct_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)fbt, (long)ld_f, (const bf16_t*)wb, (long)ld_w,
                     (int)K, (int)B, (const long long*)labels, gt);
  return vdk_check_launch("vdk_margin_target_cos_direct");
}
int vdk_margin_ce_amp(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, float label_smoothing,
                      float grad_scale, const float* loss_scale, float* logits, int64_t ldl, float* loss_rows, void* dcos_bf16, int64_t lddc, int32_t dc_dtype, void* stream) {
  MarginP P; int rc = fill_params(h, &P); if (rc) return rc;
  if (!cosv || !labels || B <= 0 || C <= 0 || (dcos_bf16 && lddc < C) || (dc_dtype != VDK_BF16 && dc_dtype != VDK_F16)) return vdk_fail(VDK_EINVAL, "vdk_margin_ce: bad argument");
  const bool f16 = dc_dtype == VDK_F16;
  const bool vec = !logits && (loss_rows || dcos_bf16) && C >= 4096 && (ldc & 3) == 0 && ((size_t)cosv & 15) == 0 &&
                   (!dcos_bf16 || ((lddc & 3) == 0 && ((size_t)dcos_bf16 & 7) == 0));
  const char* genv = getenv("VDK_MARGIN_GENERIC");      // A/B and tests: =1 runs the generic evaluation also for ArcFace (read per call)
  const bool generic_env = genv && atoi(genv) == 1;
#define MCV(ARC, OF) hipLaunchKernelGGL((margin_ce_vec_kernel<ARC, OF>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, P, cosv, (long)ldc, (int)C, (const long long*)labels, \
                                        label_smoothing, grad_scale, loss_rows, (bf16_t*)dcos_bf16, (long)lddc, loss_scale)
  if (vec && P.mode == VDK_HEAD_ARCFACE && !P.row_margin && !generic_env) { if (f16) MCV(true, VDK_OPF_F16); else MCV(true, 0); }
  else if (vec) { if (f16) MCV(false, VDK_OPF_F16); else MCV(false, 0); }
  else
    hipLaunchKernelGGL(margin_ce_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, P, cosv, (long)ldc, (int)C, (const long long*)labels,
                       label_smoothing, grad_scale, logits, (long)ldl, loss_rows, (bf16_t*)dcos_bf16, (long)lddc, (float*)nullptr, f16 ? 1 : 0, loss_scale);
#undef MCV
  return vdk_check_launch("vdk_margin_ce");
}
int vdk_margin_ce(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, float label_smoothing,
                  float grad_scale, float* logits, int64_t ldl, float* loss_rows, void* dcos_bf16, int64_t lddc, void* stream) {
  return vdk_margin_ce_amp(h, cosv, ldc, B, C, labels, label_smoothing, grad_scale, nullptr, logits, ldl, loss_rows, dcos_bf16, lddc, VDK_BF16, stream);
}
/* vdk_margin_ce with the logit gradient d(loss)/d(cos) left in fp32 [B, lddc] (columns C .. lddc - 1 zeroed): the head of the fp32-class training mode */
int vdk_margin_ce_f32(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, float label_smoothing, float grad_scale,
                      float* loss_rows, float* dcos_f32, int64_t lddc, void* stream) {
  MarginP P; int rc = fill_params(h, &P); if (rc) return rc;
  if (!cosv || !labels || !dcos_f32 || B <= 0 || C <= 0 || lddc < C) return vdk_fail(VDK_EINVAL, "vdk_margin_ce_f32: bad argument");
  hipLaunchKernelGGL(margin_ce_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, P, cosv, (long)ldc, (int)C, (const long long*)labels, label_smoothing, grad_scale,
                     (float*)nullptr, 0L, loss_rows, (bf16_t*)nullptr, (long)lddc, dcos_f32);
  return vdk_check_launch("vdk_margin_ce_f32");
}
int vdk_margin_target_cos(const float* cosv, int64_t ldc, int32_t B, int32_t Cloc, int64_t c_base, const int64_t* labels, float* gt, void* stream) {
  if (!cosv || !labels || !gt || B <= 0 || Cloc <= 0) return vdk_fail(VDK_EINVAL, "vdk_margin_target_cos: bad argument");
  hipLaunchKernelGGL(margin_target_cos_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cosv, (long)ldc, (int)B, (int)Cloc, (long)c_base,
                     (const long long*)labels, gt);
  return vdk_check_launch("vdk_margin_target_cos");
}
int vdk_margin_stats(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t Cloc, int64_t c_base, const int64_t* labels, const float* gt,
                     float* stats, void* stream) {
  MarginP P; int rc = fill_params(h, &P); if (rc) return rc;
  if (!cosv || !labels || !gt || !stats || B <= 0 || Cloc <= 0) return vdk_fail(VDK_EINVAL, "vdk_margin_stats: bad argument");
  hipLaunchKernelGGL(margin_stats_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, P, cosv, (long)ldc, (int)Cloc, (long)c_base, (const long long*)labels, gt,
                     stats);
  return vdk_check_launch("vdk_margin_stats");
}
int vdk_margin_grad(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t Cloc, int64_t c_base, int64_t C_total, const int64_t* labels,
                    const float* gt, const float* gmax, const float* gsum, float label_smoothing, float grad_scale, void* dcos_bf16, int64_t lddc, void* stream) {
  MarginP P; int rc = fill_params(h, &P); if (rc) return rc;
  if (!cosv || !labels || !gt || !gmax || !gsum || !dcos_bf16 || B <= 0 || Cloc <= 0 || C_total < Cloc || lddc < Cloc)
    return vdk_fail(VDK_EINVAL, "vdk_margin_grad: bad argument");
  hipLaunchKernelGGL(margin_grad_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, P, cosv, (long)ldc, (int)Cloc, (long)c_base, (long)C_total,
                     (const long long*)labels, gt, gmax, gsum, label_smoothing, grad_scale, (bf16_t*)dcos_bf16, (long)lddc);
  return vdk_check_launch("vdk_margin_grad");
}
int vdk_margin_bwd(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, const float* dlogits,
                   int64_t lddl, void* dcos_bf16, int64_t lddc, void* stream) {
  MarginP P; int rc = fill_params(h, &P); if (rc) return rc;
  if (!cosv || !labels || !dlogits || !dcos_bf16 || B <= 0 || C <= 0 || lddc < C) return vdk_fail(VDK_EINVAL, "vdk_margin_bwd: bad argument");
  hipLaunchKernelGGL(margin_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, P, cosv, (long)ldc, (int)C, (const long long*)labels, dlogits,
                     (long)lddl, (bf16_t*)dcos_bf16, (long)lddc);
  return vdk_check_launch("vdk_margin_bwd");
}

}  // extern "C"

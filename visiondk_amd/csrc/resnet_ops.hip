// resnet_ops.hip — what a BatchNorm-based CNN (timm ResNet BasicBlock family: `timm-resnet18`, BASELINE.json configs[0], built through
// /root/reference models/classifier/classify_model.py:49-54) needs besides the implicit-GEMM convolution of csrc/gemm.hip (VdkConvGeom):
//   * weight layouts for the implicit GEMM: [Co][(ky,kx,ci)] forward, [Ci][(ky,kx,co)] input gradient, and the gradient back to [Co][Ci][KH][KW];
//   * NCHW f32 image -> NHWC bf16 with the channel count padded to a multiple of 8 (so the 7x7 stem is an ordinary implicit conv);
//   * explicit bf16 im2col, used ONLY for the weight gradient (contraction over pixels: TN GEMM dY^T . col);
//   * BatchNorm2d on NHWC rows with many samples and few channels (64 ... 512): slice-parallel statistics, fused normalise + residual + ReLU -> bf16,
//     and the matching backward (ReLU mask from the saved output, dgamma / dbeta, bf16 dY for the conv GEMMs, f32 shortcut gradient);
//   * MaxPool2d(3, 2, 1) forward / backward with torch's first-maximum tie rule, global average pool forward / backward.
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"

// The 16-bit format of every activation / weight-operand tensor these kernels read or write: bfloat16 (default) or IEEE half (VdkResNetConfig.operand_dtype = VDK_F16: the
// reference's autocast dtype on a GPU, engine/procedure/train.py:118).  A calling thread selects it with vdk_resnet_ops_format(); the kernels take it as an argument.
static thread_local int t_rn_opf = VDK_OPF_BF16;
__device__ __forceinline__ bf16_t rn_to16(float v, int opf) { return opf ? f2op<VDK_OPF_F16>(v) : f2bf(v); }
__device__ __forceinline__ float rn_from16(bf16_t h, int opf) { return opf ? op2f<VDK_OPF_F16>(h) : bf2f(h); }
__device__ __forceinline__ unsigned rn_pack2(float a, float b, int opf) { return opf ? pack_h2(a, b) : pack_bf2(a, b); }
__device__ __forceinline__ float rn_lo(unsigned w, int opf) { return opf ? h_lo(w) : bf_lo(w); }
__device__ __forceinline__ float rn_hi(unsigned w, int opf) { return opf ? h_hi(w) : bf_hi(w); }

// ------------------------------------------------------------------------------------ weight layouts
// w f32 [Co][Ci][KH][KW] -> wf bf16 [Co][KH*KW*Cip] (k = (ky*KW + kx)*Cip + ci, ci >= Ci zero) and wd bf16 [Cip][KH*KW*Co] (k = (ky*KW + kx)*Co + co)
__global__ __launch_bounds__(256) void conv_weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, int Co, int Ci, int Cip,
                                                               int KH, int KW, int opf) {
  const long n = (long)Co * Cip * KH * KW;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ci = (int)(i % Cip);
  long r = i / Cip;
  const int t = (int)(r % (KH * KW)), co = (int)(r / (KH * KW));
  const float v = ci < Ci ? w[((long)co * Ci + ci) * KH * KW + t] : 0.f;
  const bf16_t b = rn_to16(v, opf);
  wf[i] = b;
  if (wd) wd[((long)ci * KH * KW + t) * Co + co] = b;
}
// dwp f32 [Co][KH*KW*Cip] -> dw f32 [Co][Ci][KH][KW]
__global__ __launch_bounds__(256) void conv_wgrad_unpermute_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int Cip, int KH, int KW) {
  const long n = (long)Co * Ci * KH * KW;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % (KH * KW));
  long r = i / (KH * KW);
  const int ci = (int)(r % Ci), co = (int)(r / Ci);
  dw[i] = dwp[((long)co * KH * KW + t) * Cip + ci];
}
// x f32 [B][C][H][W] -> out bf16 [B][H][W][Cp] (channels >= C zero)
__global__ __launch_bounds__(256) void nchw_to_nhwc_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B, int C, int H, int W, int Cp, int opf) {
  const long n = (long)B * H * W * Cp;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % Cp);
  const long p = i / Cp;
  const int xx = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
  out[i] = rn_to16(c < C ? x[(((long)b * C + c) * H + y) * W + xx] : 0.f, opf);
}
// explicit im2col for the weight gradient: col bf16 [B*OH*OW][KH*KW*C] (k = (ky*KW + kx)*C + c) from in bf16 [B][H][W][C]; 16-byte chunks
__global__ __launch_bounds__(256) void im2col_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ col, int B, int H, int W, int C, int OH, int OW, int KH,
                                                          int KW, int stride, int pad) {
  const int c8n = C / 8;
  const long n = (long)B * OH * OW * KH * KW * c8n;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c8 = (int)(i % c8n);
  long r = i / c8n;
  const int t = (int)(r % (KH * KW));
  const long m = r / (KH * KW);
  const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((long)OW * OH));
  const int ky = t / KW, kx = t % KW, y = oy * stride + ky - pad, x = ox * stride + kx - pad;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (y >= 0 && y < H && x >= 0 && x < W) v = *(const u32x4*)(in + (((long)b * H + y) * W + x) * C + c8 * 8);
  *(u32x4*)(col + (m * KH * KW + t) * C + c8 * 8) = v;
}

// ------------------------------------------------------------------------------------ BatchNorm2d on NHWC rows
// statistics: grid (S slices, ceil(C / 64)); 512 threads = 64 columns x 8 row groups; a slice owns rows [s*rps, (s+1)*rps)
// MODE 0: (sum x, sum x^2);  MODE 1 (backward): g = dout * (relu ? out > 0 : 1): (sum g, sum g * xhat)
template <int MODE>
__global__ __launch_bounds__(512) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dout, const bf16_t* __restrict__ outb,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd, long R, int C, long rps,
                                                         float* __restrict__ part /*[S][2][C]*/, int opf) {
  __shared__ float red[2][8][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.y * 64 + cl;
  const long r0 = (long)blockIdx.x * rps;
  long r1 = r0 + rps; if (r1 > R) r1 = R;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    const float mu = MODE ? mean[c] : 0.f, is = MODE ? invstd[c] : 0.f;
    for (long r = r0 + rg; r < r1; r += 8) {
      if (MODE == 0) { const float v = x[r * C + c]; a0 += v; a1 = fmaf(v, v, a1); }
      else {
        float g = dout[r * C + c];
        if (outb && !(rn_from16(outb[r * C + c], opf) > 0.f)) g = 0.f;
        a0 += g; a1 = fmaf(g, (x[r * C + c] - mu) * is, a1);
      }
    }
  }
  red[0][rg][cl] = a0; red[1][rg][cl] = a1;
  __syncthreads();
  if (rg < 2 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[rg][i][cl];
    part[((long)blockIdx.x * 2 + rg) * C + c] = t;
  }
}
// combine the slice partials into ONE vector sums[2C + 1] = (sum_0[C], sum_1[C], count): what a SyncBatchNorm all-reduces across ranks
// 16 columns x 16 slice groups per workgroup (a thread per column walking all S partials serially took 40 - 260 us per call: 20 % of a ResNet-18 step
// at batch 32); fp64 accumulation, fixed combination order
__global__ __launch_bounds__(256) void bn_combine_kernel(const float* __restrict__ part, int S, int C, float count, float* __restrict__ sums) {
  __shared__ double red[16][17];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + col;
  if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * C] = count;
  double s = 0.0;
  if (i < 2 * C) {
    const int which = i / C, c = i - which * C;
    for (int k = grp; k < S; k += 16) s += part[((long)k * 2 + which) * C + c];
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && i < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][col];
    sums[i] = (float)t;
  }
}
// forward finalize from sums (sum x, sum x^2, count): mean / biased var -> invstd, running statistics (unbiased var); eval mode: running statistics
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* __restrict__ sums, int C, float eps, float momentum, int training, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                              float* __restrict__ save_count) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float mu, var;
  if (training) {
    const double R = (double)sums[2 * C];
    const double m = (double)sums[c] / R;
    double v = (double)sums[C + c] / R - m * m; if (v < 0.0) v = 0.0;
    mu = (float)m; var = (float)v;
    if (rmean) rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mu;
    if (rvar) rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (R > 1.0 ? (float)(v * R / (R - 1.0)) : var);
    if (c == 0 && save_count) save_count[0] = (float)R;
  } else {
    mu = rmean[c]; var = rvar[c];
    if (c == 0 && save_count) save_count[0] = 0.f;      // "normalised with the running statistics": the backward then has no batch-statistics terms (bn_bwd_apply_kernel)
  }
  save_mean[c] = mu; save_invstd[c] = 1.0f / sqrtf(var + eps);
}
// backward: weight / bias gradients from the LOCAL sums (data-parallel training all-reduces gradients later, like torch's SyncBatchNorm)
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = sums[c]; dgamma[c] = sums[C + c];
}
// y = gamma * (x - mean) * invstd + beta [+ res] [relu] -> bf16 (and / or f32); 4 channels per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long n4, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ res_f32,
                                                       const bf16_t* __restrict__ res_bf16, int relu, bf16_t* __restrict__ outb, float* __restrict__ outf, int opf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  f32x4 v = *(const f32x4*)(x + i * 4);
  const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c), mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (v[e] - mu[e]) * (g[e] * is[e]) + b[e];
  if (res_f32) { const f32x4 r = *(const f32x4*)(res_f32 + i * 4); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
  if (res_bf16) { const u32x2 r = *(const u32x2*)(res_bf16 + i * 4); v[0] += rn_lo(r[0], opf); v[1] += rn_hi(r[0], opf); v[2] += rn_lo(r[1], opf); v[3] += rn_hi(r[1], opf); }
  if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
  if (outb) *(u32x2*)(outb + i * 4) = (u32x2){rn_pack2(v[0], v[1], opf), rn_pack2(v[2], v[3], opf)};
  if (outf) *(f32x4*)(outf + i * 4) = v;
}
// g = dout * mask;  dy = gamma * invstd / R * (R g - sum g - xhat * sum(g xhat)) -> bf16;  dres = g (f32, optional: the shortcut's gradient)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dout, const bf16_t* __restrict__ outb, long n4, int C,
                                                           const float* __restrict__ count, const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ sums, bf16_t* __restrict__ dyb, float* __restrict__ dres, float* __restrict__ dxf, int opf,
                                                           const float* __restrict__ fwd_count /* the forward's saved sample count (NULL: unknown = batch statistics); 0 = it ran on
                                                                                                  the running statistics */) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const f32x4 xv = *(const f32x4*)(x + i * 4);
  f32x4 g = *(const f32x4*)(dout + i * 4);
  if (outb) {
    const u32x2 o = *(const u32x2*)(outb + i * 4);
    if (!(rn_lo(o[0], opf) > 0.f)) g[0] = 0.f;
    if (!(rn_hi(o[0], opf) > 0.f)) g[1] = 0.f;
    if (!(rn_lo(o[1], opf) > 0.f)) g[2] = 0.f;
    if (!(rn_hi(o[1], opf) > 0.f)) g[3] = 0.f;
  }
  const f32x4 ga = *(const f32x4*)(gamma + c), mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
  const f32x4 sb = *(const f32x4*)(sums + c), sg = *(const f32x4*)(sums + C + c);
  const float Rf = count[0];   // samples behind the statistics: local rows, or all ranks' rows under SyncBatchNorm
  const bool batch_stats = !fwd_count || fwd_count[0] > 0.f;
  float d[4];
  if (batch_stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - mu[e]) * is[e];
      d[e] = ga[e] * is[e] / Rf * (Rf * g[e] - sb[e] - xh * sg[e]);
    }
  } else {      // mean and invstd are constants of the pass: dy = gamma * invstd * g (dgamma / dbeta are the same sums either way)
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = ga[e] * is[e] * g[e];
  }
  if (dyb) *(u32x2*)(dyb + i * 4) = (u32x2){rn_pack2(d[0], d[1], opf), rn_pack2(d[2], d[3], opf)};
  if (dxf) *(f32x4*)(dxf + i * 4) = (f32x4){d[0], d[1], d[2], d[3]};
  if (dres) *(f32x4*)(dres + i * 4) = g;
}

// ------------------------------------------------------------------------------------ pooling
// MaxPool2d(3, stride 2, padding 1) on NHWC bf16 (out-of-range taps are -inf, like torch)
// argmax (optional): the window position 0..8 in (ky, kx) scan order of the FIRST maximum, one byte per output element -- what the backward needs
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, unsigned char* __restrict__ argmax, int B, int H,
                                                             int W, int C, int OH, int OW, int opf) {
  const long n = (long)B * OH * OW * C;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long p = i / C;
  const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), b = (int)(p / ((long)OW * OH));
  float m = -3.0e38f;
  int km = 0;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int y = oy * 2 + ky - 1, x = ox * 2 + kx - 1;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        const float v = rn_from16(in[(((long)b * H + y) * W + x) * C + c], opf);
        if (v > m) { m = v; km = ky * 3 + kx; }
      }
    }
  out[i] = rn_to16(m, opf);
  if (argmax) argmax[i] = (unsigned char)km;
}
// din[b,y,x,c] = sum over the (up to 4) windows containing (y,x) in which it is the FIRST maximum in (ky, kx) scan order (torch's argmax rule) of dout.
// With the forward's argmax map: one byte and (on a hit) one dout per window; without it the 9 window entries are re-read per window (36 loads per element:
// 733 us for 32 x 112 x 112 x 64 against 60 us with the map).
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const bf16_t* __restrict__ in, const unsigned char* __restrict__ argmax, const float* __restrict__ dout,
                                                             float* __restrict__ din, int B, int H, int W, int C, int OH, int OW, int opf) {
  const long n = (long)B * H * W * C;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long p = i / C;
  const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
  const float v = argmax ? 0.f : rn_from16(in[i], opf);
  float acc = 0.f;
  for (int oy = (y + 1 - 2 + 1) / 2; oy <= (y + 1) / 2; ++oy) {     // windows with oy*2 - 1 <= y <= oy*2 + 1
    if (oy < 0 || oy >= OH) continue;
    for (int ox = (x + 1 - 2 + 1) / 2; ox <= (x + 1) / 2; ++ox) {
      if (ox < 0 || ox >= OW) continue;
      const int myk = (y - (oy * 2 - 1)) * 3 + (x - (ox * 2 - 1));
      const long o = (((long)b * OH + oy) * OW + ox) * C + c;
      bool win = true;
      if (argmax) win = argmax[o] == myk;
      else
        for (int k = 0; k < 9 && win; ++k) {
          const int yy = oy * 2 - 1 + k / 3, xx = ox * 2 - 1 + k % 3;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W || k == myk) continue;
          const float q = rn_from16(in[(((long)b * H + yy) * W + xx) * C + c], opf);
          if (q > v || (q == v && k < myk)) win = false;
        }
      if (win) acc += dout[o];
    }
  }
  din[i] = acc;
}
// global average pool: out bf16 [Bp][C] (rows >= B zero) = mean over HW of in bf16 [B][HW][C]
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int Bp, int HW, int C, int opf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)Bp * C) return;
  const int c = (int)(i % C), b = (int)(i / C);
  float s = 0.f;
  if (b < B) for (int p = 0; p < HW; ++p) s += rn_from16(in[((long)b * HW + p) * C + c], opf);
  out[i] = rn_to16(s / (float)HW, opf);
}
// dout f32 [B*HW][C] = dfeat[b][c] / HW  (dfeat bf16 [B][ld])
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const bf16_t* __restrict__ dfeat, long ld, float* __restrict__ dout, int B, int HW, int C, int opf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * HW * C) return;
  const int c = (int)(i % C);
  const int b = (int)(i / ((long)HW * C));
  dout[i] = rn_from16(dfeat[(long)b * ld + c], opf) / (float)HW;
}

static inline int bn_slices(long R, int C) {
  long s = 1024 / ((C + 63) / 64);
  if (s > (R + 255) / 256) s = (R + 255) / 256;
  return (int)(s < 1 ? 1 : s);
}

extern "C" {

// the calling thread's 16-bit format for the functions of this file: VDK_BF16 (default) | VDK_F16
int vdk_resnet_ops_format(int32_t dtype) {
  if (dtype != VDK_BF16 && dtype != VDK_F16) return vdk_fail(VDK_EINVAL, "vdk_resnet_ops_format: VDK_BF16 or VDK_F16");
  t_rn_opf = dtype == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16;
  return VDK_OK;
}

int vdk_conv_weight_prep(const float* w, void* wf, void* wd, int32_t Co, int32_t Ci, int32_t Cip, int32_t KH, int32_t KW, void* stream) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  if (!w || !wf || Co <= 0 || Ci <= 0 || Cip < Ci || (Cip & 7) || KH <= 0 || KW <= 0) return vdk_fail(VDK_EINVAL, "vdk_conv_weight_prep: bad argument (Cip % 8 == 0)");
  const long n = (long)Co * Cip * KH * KW;
  hipLaunchKernelGGL(conv_weight_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wf, (bf16_t*)wd, (int)Co, (int)Ci, (int)Cip,
                     (int)KH, (int)KW, opf_);
  return vdk_check_launch("vdk_conv_weight_prep");
}
int vdk_conv_wgrad_unpermute(const float* dwp, float* dw, int32_t Co, int32_t Ci, int32_t Cip, int32_t KH, int32_t KW, void* stream) {
  if (!dwp || !dw || Co <= 0 || Ci <= 0 || Cip < Ci || KH <= 0 || KW <= 0) return vdk_fail(VDK_EINVAL, "vdk_conv_wgrad_unpermute: bad argument");
  const long n = (long)Co * Ci * KH * KW;
  hipLaunchKernelGGL(conv_wgrad_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dwp, dw, (int)Co, (int)Ci, (int)Cip, (int)KH, (int)KW);
  return vdk_check_launch("vdk_conv_wgrad_unpermute");
}
int vdk_nchw_to_nhwc_bf16(const float* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp, void* stream) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) return vdk_fail(VDK_EINVAL, "vdk_nchw_to_nhwc_bf16: bad argument");
  const long n = (long)B * H * W * Cp;
  hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out, (int)B, (int)C, (int)H, (int)W, (int)Cp, opf_);
  return vdk_check_launch("vdk_nchw_to_nhwc_bf16");
}
int vdk_im2col_bf16(const void* in, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                    void* stream) {
  if (!in || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    return vdk_fail(VDK_EINVAL, "vdk_im2col_bf16: bad argument (C % 8 == 0)");
  const long n = (long)B * OH * OW * KH * KW * (C / 8);
  hipLaunchKernelGGL(im2col_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)col, (int)B, (int)H, (int)W, (int)C,
                     (int)OH, (int)OW, (int)KH, (int)KW, (int)stride, (int)pad);
  return vdk_check_launch("vdk_im2col_bf16");
}

int vdk_bn_rows_workspace_bytes(int64_t R, int32_t C, size_t* bytes) {
  if (!bytes || R <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_bn_rows_workspace_bytes: bad argument");
  *bytes = ((size_t)bn_slices(R, C) * 2 * C + 2 * (size_t)C + 64) * 4;
  return VDK_OK;
}
/* BatchNorm2d (+ residual, + ReLU) on NHWC rows: x f32 [R, C] -> out_bf16 and / or out_f32; training != 0: batch statistics (saved) and running update.
 * save_stats: f32 [2C + 1] = (mean, invstd, sample count).  sync != NULL (SyncBatchNorm, the reference's opt-in at engine/vision_engine.py:224-225): called on
 * the host with the device vector (sum x, sum x^2, count) [2C + 1] after its producer is enqueued; the callee enqueues a SUM all-reduce over the ranks. */
int vdk_bn_act_fwd(const float* x, int64_t R, int32_t C, const float* gamma, const float* beta, float eps, float momentum, int32_t training, float* running_mean,
                   float* running_var, const float* res_f32, const void* res_bf16, int32_t relu, void* out_bf16, float* out_f32, float* save_mean, float* save_invstd,
                   void* ws, size_t ws_bytes, vdk_stat_sync_fn sync, void* user, void* stream_) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !gamma || !beta || !save_mean || !save_invstd || (!out_bf16 && !out_f32) || R <= 0 || C <= 0 || (C & 3) || (!training && (!running_mean || !running_var)))
    return vdk_fail(VDK_EINVAL, "vdk_bn_act_fwd: bad argument (C % 4 == 0)");
  const int S = bn_slices(R, C);
  if (!ws || ws_bytes < ((size_t)S * 2 * C + 2 * (size_t)C + 64) * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_bn_act_fwd: workspace too small");
  float* part = (float*)ws; float* sums = part + (size_t)S * 2 * C;
  const long rps = (R + S - 1) / S;
  if (training) {
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3((unsigned)S, (unsigned)((C + 63) / 64)), dim3(512), 0, stream, x, (const float*)nullptr, (const bf16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (long)R, (int)C, rps, part, opf_);
    hipLaunchKernelGGL(bn_combine_kernel, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, stream, (const float*)part, S, (int)C, (float)R, sums);
    if (sync) sync(user, sums, 2 * (int64_t)C + 1);
  }
  // the sample count is kept right behind invstd when the two save vectors are adjacent (engines allocate [2C + 1])
  float* save_count = (save_invstd == save_mean + C) ? save_invstd + C : nullptr;
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, (const float*)sums, (int)C, eps, momentum, (int)training, running_mean,
                     running_var, save_mean, save_invstd, save_count);
  const long n4 = R * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, n4, (int)C, gamma, beta, (const float*)save_mean,
                     (const float*)save_invstd, res_f32, (const bf16_t*)res_bf16, (int)relu, (bf16_t*)out_bf16, out_f32, opf_);
  return vdk_check_launch("vdk_bn_act_fwd");
}
/* backward of the above (training mode): dout f32 = gradient of the block output; out_bf16 = that output (ReLU mask; NULL = no ReLU).  sync != NULL: the vector
 * (sum g, sum g x^, local count) [2C + 1] is all-reduced before dy is formed (dgamma / dbeta stay local, like torch.nn.SyncBatchNorm). */
int vdk_bn_act_bwd(const float* x, const float* dout, const void* out_bf16, int64_t R, int32_t C, const float* gamma, const float* save_mean, const float* save_invstd,
                   void* dy_bf16, float* dres, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, vdk_stat_sync_fn sync, void* user, void* stream_) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !dout || !gamma || !save_mean || !save_invstd || !dy_bf16 || !dgamma || !dbeta || R <= 0 || C <= 0 || (C & 3))
    return vdk_fail(VDK_EINVAL, "vdk_bn_act_bwd: bad argument (C % 4 == 0)");
  const int S = bn_slices(R, C);
  if (!ws || ws_bytes < ((size_t)S * 2 * C + 2 * (size_t)C + 64) * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_bn_act_bwd: workspace too small");
  float* part = (float*)ws; float* sums = part + (size_t)S * 2 * C;
  const long rps = (R + S - 1) / S;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3((unsigned)S, (unsigned)((C + 63) / 64)), dim3(512), 0, stream, x, dout, (const bf16_t*)out_bf16, save_mean, save_invstd,
                     (long)R, (int)C, rps, part, opf_);
  hipLaunchKernelGGL(bn_combine_kernel, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, stream, (const float*)part, S, (int)C, (float)R, sums);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, (const float*)sums, (int)C, dgamma, dbeta);
  if (sync) sync(user, sums, 2 * (int64_t)C + 1);
  const long n4 = R * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, dout, (const bf16_t*)out_bf16, n4, (int)C, (const float*)(sums + 2 * C),
                     gamma, save_mean, save_invstd, (const float*)sums, (bf16_t*)dy_bf16, dres, (float*)nullptr, opf_,
                     (const float*)((save_invstd == save_mean + C) ? save_invstd + C : nullptr));
  return vdk_check_launch("vdk_bn_act_bwd");
}
// The same backward with the input gradient in fp32 and no activation mask: the BatchNorm2d / BatchNorm1d of the embedding neck
// (models/faceX/backbone/timm_wrapper.py:30-38) under SyncBatchNorm (engine/vision_engine.py:224-225 converts EVERY BatchNorm of the model).
int vdk_bn_rows_bwd(const float* x, const float* dout, int64_t R, int32_t C, const float* gamma, const float* save_mean, const float* save_invstd, float* dx,
                    float* dgamma, float* dbeta, void* ws, size_t ws_bytes, vdk_stat_sync_fn sync, void* user, void* stream_) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !dout || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || R <= 0 || C <= 0 || (C & 3))
    return vdk_fail(VDK_EINVAL, "vdk_bn_rows_bwd: bad argument (C % 4 == 0)");
  const int S = bn_slices(R, C);
  if (!ws || ws_bytes < ((size_t)S * 2 * C + 2 * (size_t)C + 64) * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_bn_rows_bwd: workspace too small");
  float* part = (float*)ws; float* sums = part + (size_t)S * 2 * C;
  const long rps = (R + S - 1) / S;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3((unsigned)S, (unsigned)((C + 63) / 64)), dim3(512), 0, stream, x, dout, (const bf16_t*)nullptr, save_mean, save_invstd,
                     (long)R, (int)C, rps, part, opf_);
  hipLaunchKernelGGL(bn_combine_kernel, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, stream, (const float*)part, S, (int)C, (float)R, sums);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, (const float*)sums, (int)C, dgamma, dbeta);
  if (sync) sync(user, sums, 2 * (int64_t)C + 1);
  const long n4 = R * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, dout, (const bf16_t*)nullptr, n4, (int)C, (const float*)(sums + 2 * C),
                     gamma, save_mean, save_invstd, (const float*)sums, (bf16_t*)nullptr, (float*)nullptr, dx, opf_, (const float*)nullptr);
  return vdk_check_launch("vdk_bn_rows_bwd");
}

int vdk_maxpool3s2_fwd(const void* in, void* out, uint8_t* argmax, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_maxpool3s2_fwd: bad argument");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long n = (long)B * OH * OW * C;
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, (unsigned char*)argmax, (int)B, (int)H,
                     (int)W, (int)C, OH, OW, opf_);
  return vdk_check_launch("vdk_maxpool3s2_fwd");
}
int vdk_maxpool3s2_bwd(const void* in, const uint8_t* argmax, const float* dout, float* din, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  if ((!in && !argmax) || !dout || !din || B <= 0 || H <= 0 || W <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_maxpool3s2_bwd: bad argument");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long n = (long)B * H * W * C;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (const unsigned char*)argmax, dout, din, (int)B, (int)H,
                     (int)W, (int)C, OH, OW, opf_);
  return vdk_check_launch("vdk_maxpool3s2_bwd");
}
int vdk_avgpool_fwd(const void* in, void* out, int32_t B, int32_t Bp, int32_t HW, int32_t C, void* stream) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  if (!in || !out || B <= 0 || Bp < B || HW <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_avgpool_fwd: bad argument");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((unsigned)(((long)Bp * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, (int)B, (int)Bp, (int)HW,
                     (int)C, opf_);
  return vdk_check_launch("vdk_avgpool_fwd");
}
int vdk_avgpool_bwd(const void* dfeat, int64_t ld, float* dout, int32_t B, int32_t HW, int32_t C, void* stream) {
  const int opf_ = t_rn_opf;      // (read on the calling thread: the emulator evaluates launch arguments on its worker threads)
  if (!dfeat || !dout || B <= 0 || HW <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_avgpool_bwd: bad argument");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((unsigned)(((long)B * HW * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dfeat, (long)ld, dout, (int)B, (int)HW,
                     (int)C, opf_);
  return vdk_check_launch("vdk_avgpool_bwd");
}

}  // extern "C"

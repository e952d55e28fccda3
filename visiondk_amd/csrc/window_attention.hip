// window_attention.hip -- the attention step of timm's Swin Transformer (WindowAttention inside SwinTransformerBlock): the default backbone of both shipped
// configs of the reference (`timm-swin_base_patch4_window7_224`: configs/classification/pet.yaml:25, configs/faceX/cbir.yaml:26; built by timm.create_model in
// models/classifier/classify_model.py:49-54 and models/faceX/backbone/timm_wrapper.py:16-21).  Per (window, head):
//     S = (q * hd^-0.5) k^T + bias[head] (+ mask[window mod nW]);   P = softmax(S);   o = P v            N = 49 tokens (7 x 7), head dim 32
// qkv: bf16 [W * N, 3 C] rows in (window, token) order, q | k | v thirds, head h at columns h * 32; bias f32 [H, N, N] (the relative-position table gathered once per
// step by the host side); mask f32 [nW, N, N] (0 / -100 of the shifted windows) or NULL.  Arithmetic as under the reference's autocast: bf16 operands, fp32 products and
// sums, softmax in fp32, P rounded to bf16 once as the left operand of P v, o rounded to bf16.
//
// Structure: 65 536 tiny problems per layer at batch 256 (49 x 49 x 32): ONE WAVE per (window, head), lane = query row, the window's K / V rows as fp32 in the wave's
// own LDS (read as broadcasts), the lane's 49 scores in registers.  No workgroup barriers.  This is the first, plain-VALU form (2 x 49 x 32 FMAs per lane and item);
// the MFMA form (a 64 x 64 x 32 tile per item) is the obvious next step once the family is profiled.
// Backward: the same mapping recomputes P from the saved row log-sum-exp; dQ is lane-local; dK / dV contract over the queries, i.e. over lanes: P and dS go through the
// wave's LDS as bf16 [N][N] and lane j then owns key j.  d(bias) = sum of dS over every window of a head: each wave walks the windows of ONE head and keeps the sum of
// its dS rows in registers; one partial [N, N] per wave, reduced by vdk_reduce_rows_f32 in a fixed order (no atomics: bit-reproducible).
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"

#define WA_N 49
#define WA_HD 32
#define WA_LOG2E 1.4426950408889634f

// a lane's row of 32 bf16 -> 32 floats
__device__ __forceinline__ void wa_load_row(const bf16_t* __restrict__ p, float (&r)[WA_HD]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const u32x4 v = *(const u32x4*)(p + 8 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[8 * c + 2 * e] = bf_lo(v[e]); r[8 * c + 2 * e + 1] = bf_hi(v[e]); }
  }
}

__global__ __launch_bounds__(256) void window_attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o, long ldo, float* __restrict__ lse,
                                                              const float* __restrict__ bias, const float* __restrict__ mask, int nW, long items, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[4][WA_N * WA_HD];
  __shared__ __attribute__((aligned(16))) float Vs[4][WA_N * WA_HD];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = H * WA_HD;
  const bool row = lane < WA_N;
  for (long item = (long)blockIdx.x * 4 + w; item < items; item += (long)gridDim.x * 4) {
    const long win = item / H; const int h = (int)(item - win * H);
    const bf16_t* base = qkv + (win * WA_N + (row ? lane : 0)) * ld + h * WA_HD;
    float q[WA_HD], t[WA_HD];
    wa_load_row(base, q);
    VDK_WAVE_LDS_SYNC();                                  // the previous item's readers are done
    wa_load_row(base + C, t);
    if (row) {
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) *(f32x4*)(&Ks[w][lane * WA_HD + d]) = (f32x4){t[d], t[d + 1], t[d + 2], t[d + 3]};
    }
    wa_load_row(base + 2 * C, t);
    if (row) {
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) *(f32x4*)(&Vs[w][lane * WA_HD + d]) = (f32x4){t[d], t[d + 1], t[d + 2], t[d + 3]};
    }
    VDK_WAVE_LDS_SYNC();
    const float* brow = bias + ((long)h * WA_N + (row ? lane : 0)) * WA_N;
    const float* mrow = mask ? mask + ((win % nW) * WA_N + (row ? lane : 0)) * WA_N : nullptr;
    float s[WA_N];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < WA_N; ++j) {
      vdk_f32x2 a2 = {0.f, 0.f};                           // two partial sums on packed FMAs (v_pk_fma_f32: the kernel is bound by its FMA count)
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) {
        const f32x4 kv = *(const f32x4*)(&Ks[w][j * WA_HD + d]);
        a2 = vdk_fma2((vdk_f32x2){q[d], q[d + 1]}, (vdk_f32x2){kv[0], kv[1]}, a2);
        a2 = vdk_fma2((vdk_f32x2){q[d + 2], q[d + 3]}, (vdk_f32x2){kv[2], kv[3]}, a2);
      }
      float a = a2[0] + a2[1];
      a = a * scale + brow[j];
      if (mrow) a += mrow[j];
      s[j] = a;
      mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < WA_N; ++j) { s[j] = fast_exp2((s[j] - mx) * WA_LOG2E); sum += s[j]; }
    const float inv = 1.0f / sum;
    vdk_f32x2 acc2[WA_HD / 2];
#pragma unroll
    for (int d = 0; d < WA_HD / 2; ++d) acc2[d] = (vdk_f32x2){0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WA_N; ++j) {
      const float p = bf2f(f2bf(s[j] * inv));              // P rounded once, as the left operand of P v
      const vdk_f32x2 p2 = {p, p};
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) {
        const f32x4 vv = *(const f32x4*)(&Vs[w][j * WA_HD + d]);
        acc2[d / 2] = vdk_fma2(p2, (vdk_f32x2){vv[0], vv[1]}, acc2[d / 2]);
        acc2[d / 2 + 1] = vdk_fma2(p2, (vdk_f32x2){vv[2], vv[3]}, acc2[d / 2 + 1]);
      }
    }
    if (row) {
      bf16_t* orow = o + (win * WA_N + lane) * ldo + h * WA_HD;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *(u32x4*)(orow + 8 * c) = (u32x4){pack_bf2(acc2[4 * c][0], acc2[4 * c][1]), pack_bf2(acc2[4 * c + 1][0], acc2[4 * c + 1][1]), pack_bf2(acc2[4 * c + 2][0], acc2[4 * c + 2][1]),
                                          pack_bf2(acc2[4 * c + 3][0], acc2[4 * c + 3][1])};
      if (lse) lse[(win * H + h) * WA_N + lane] = mx + logf(sum);
    }
  }
}

// one wave walks the windows win = slot, slot + nslot, ... of ONE head (h = wave index mod H); dbias_part: f32 [gridDim.x * 2][N * N] (rows of head h: wave ids = h mod H).
// Two waves per workgroup (32 KB of LDS each: K / V rows, P and dS, the wave's running d(bias) sum); the j / i loops stay rolled -- fully unrolled (the first form: the
// bias sums in 49 registers) the kernel needed 5.5 KB of scratch per lane and ran 20x slower than the forward.
#define WA_BW 2
__global__ __launch_bounds__(64 * WA_BW) void window_attn_bwd_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                                     long ldo, const float* __restrict__ lse, const float* __restrict__ bias, const float* __restrict__ mask,
                                                                     int nW, long nwin, int H, float scale, bf16_t* __restrict__ dqkv, long ldd,
                                                                     float* __restrict__ dbias_part) {
  __shared__ __attribute__((aligned(16))) float Ks[WA_BW][WA_N * WA_HD];      // K rows, later Q rows
  __shared__ __attribute__((aligned(16))) float Vs[WA_BW][WA_N * WA_HD];      // V rows, later dO rows
  __shared__ __attribute__((aligned(16))) bf16_t Ps[WA_BW][WA_N * 52];        // P  [query][key], row pitch 52
  __shared__ __attribute__((aligned(16))) bf16_t Ds[WA_BW][WA_N * 52];        // dS [query][key]
  __shared__ float Bs[WA_BW][WA_N * WA_N];                                    // this wave's sum of dS over its windows: lane i owns row i (pitch 49: conflict-free)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = H * WA_HD;
  const bool row = lane < WA_N;
  const long wid = (long)blockIdx.x * WA_BW + w, nwave = (long)gridDim.x * WA_BW;
  const int h = (int)(wid % H);
  const long slot = wid / H, nslot = nwave / H;            // (the launcher makes the wave count a multiple of H)
  const int li = row ? lane : 0;
  if (row) for (int j = 0; j < WA_N; ++j) Bs[w][lane * WA_N + j] = 0.f;
  const float* brow = bias + ((long)h * WA_N + li) * WA_N;
  for (long win = slot; win < nwin; win += nslot) {
    const long r0 = win * WA_N + li;
    const bf16_t* base = qkv + r0 * ld + h * WA_HD;
    float q[WA_HD], g[WA_HD], t[WA_HD];
    wa_load_row(base, q);
    wa_load_row(dout + r0 * ldo + h * WA_HD, g);
    VDK_WAVE_LDS_SYNC();
    wa_load_row(base + C, t);
    if (row) {
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) *(f32x4*)(&Ks[w][lane * WA_HD + d]) = (f32x4){t[d], t[d + 1], t[d + 2], t[d + 3]};
    }
    wa_load_row(base + 2 * C, t);
    if (row) {
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) *(f32x4*)(&Vs[w][lane * WA_HD + d]) = (f32x4){t[d], t[d + 1], t[d + 2], t[d + 3]};
    }
    // D = rowsum(dO * O) on the rounded tensors
    wa_load_row(o + r0 * ldo + h * WA_HD, t);
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < WA_HD; ++d) D = fmaf(g[d], t[d], D);
    VDK_WAVE_LDS_SYNC();
    const float* mrow = mask ? mask + ((win % nW) * WA_N + li) * WA_N : nullptr;
    const float l = row ? lse[(win * H + h) * WA_N + lane] : 0.f;
    vdk_f32x2 dq2[WA_HD / 2];
#pragma unroll
    for (int d = 0; d < WA_HD / 2; ++d) dq2[d] = (vdk_f32x2){0.f, 0.f};
#pragma unroll 1
    for (int j = 0; j < WA_N; ++j) {
      vdk_f32x2 a2 = {0.f, 0.f}, dp2 = {0.f, 0.f};
      f32x4 kv[WA_HD / 4];
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) {
        kv[d / 4] = *(const f32x4*)(&Ks[w][j * WA_HD + d]);
        const f32x4 vv = *(const f32x4*)(&Vs[w][j * WA_HD + d]);
        a2 = vdk_fma2((vdk_f32x2){q[d], q[d + 1]}, (vdk_f32x2){kv[d / 4][0], kv[d / 4][1]}, a2);
        a2 = vdk_fma2((vdk_f32x2){q[d + 2], q[d + 3]}, (vdk_f32x2){kv[d / 4][2], kv[d / 4][3]}, a2);
        dp2 = vdk_fma2((vdk_f32x2){g[d], g[d + 1]}, (vdk_f32x2){vv[0], vv[1]}, dp2);
        dp2 = vdk_fma2((vdk_f32x2){g[d + 2], g[d + 3]}, (vdk_f32x2){vv[2], vv[3]}, dp2);
      }
      float a = a2[0] + a2[1];
      const float dp = dp2[0] + dp2[1];
      a = a * scale + brow[j];
      if (mrow) a += mrow[j];
      const float p = fast_exp2((a - l) * WA_LOG2E);
      const float ds = p * (dp - D);                       // d(loss)/dS: also the bias gradient of this (query, key)
      const bf16_t pb = f2bf(p), dsb = f2bf(ds);            // the operands of dV = P^T dO and dQ / dK = dS K / dS^T Q
      if (row) { Bs[w][lane * WA_N + j] += ds; Ps[w][lane * 52 + j] = pb; Ds[w][lane * 52 + j] = dsb; }
      const float dsr = bf2f(dsb);
      const vdk_f32x2 ds2 = {dsr, dsr};
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) {
        dq2[d / 2] = vdk_fma2(ds2, (vdk_f32x2){kv[d / 4][0], kv[d / 4][1]}, dq2[d / 2]);
        dq2[d / 2 + 1] = vdk_fma2(ds2, (vdk_f32x2){kv[d / 4][2], kv[d / 4][3]}, dq2[d / 2 + 1]);
      }
    }
    bf16_t* drow = dqkv + r0 * ldd + h * WA_HD;
    if (row) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *(u32x4*)(drow + 8 * c) = (u32x4){pack_bf2(dq2[4 * c][0] * scale, dq2[4 * c][1] * scale), pack_bf2(dq2[4 * c + 1][0] * scale, dq2[4 * c + 1][1] * scale),
                                          pack_bf2(dq2[4 * c + 2][0] * scale, dq2[4 * c + 2][1] * scale), pack_bf2(dq2[4 * c + 3][0] * scale, dq2[4 * c + 3][1] * scale)};
    }
    // lane j now owns KEY j: the K / V tiles are replaced by the Q / dO rows (every lane has finished reading them)
    VDK_WAVE_LDS_SYNC();
    if (row) {
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) {
        *(f32x4*)(&Ks[w][lane * WA_HD + d]) = (f32x4){q[d], q[d + 1], q[d + 2], q[d + 3]};
        *(f32x4*)(&Vs[w][lane * WA_HD + d]) = (f32x4){g[d], g[d + 1], g[d + 2], g[d + 3]};
      }
    }
    VDK_WAVE_LDS_SYNC();
    vdk_f32x2 dk2[WA_HD / 2], dv2[WA_HD / 2];
#pragma unroll
    for (int d = 0; d < WA_HD / 2; ++d) { dk2[d] = (vdk_f32x2){0.f, 0.f}; dv2[d] = (vdk_f32x2){0.f, 0.f}; }
#pragma unroll 1
    for (int i = 0; i < WA_N; ++i) {
      const float p = bf2f(Ps[w][i * 52 + li]), ds = bf2f(Ds[w][i * 52 + li]);
      const vdk_f32x2 pp = {p, p}, dd = {ds, ds};
#pragma unroll
      for (int d = 0; d < WA_HD; d += 4) {
        const f32x4 qv = *(const f32x4*)(&Ks[w][i * WA_HD + d]);
        const f32x4 gv = *(const f32x4*)(&Vs[w][i * WA_HD + d]);
        dk2[d / 2] = vdk_fma2(dd, (vdk_f32x2){qv[0], qv[1]}, dk2[d / 2]); dk2[d / 2 + 1] = vdk_fma2(dd, (vdk_f32x2){qv[2], qv[3]}, dk2[d / 2 + 1]);
        dv2[d / 2] = vdk_fma2(pp, (vdk_f32x2){gv[0], gv[1]}, dv2[d / 2]); dv2[d / 2 + 1] = vdk_fma2(pp, (vdk_f32x2){gv[2], gv[3]}, dv2[d / 2 + 1]);
      }
    }
    if (row) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *(u32x4*)(drow + C + 8 * c) = (u32x4){pack_bf2(dk2[4 * c][0] * scale, dk2[4 * c][1] * scale), pack_bf2(dk2[4 * c + 1][0] * scale, dk2[4 * c + 1][1] * scale),
                                              pack_bf2(dk2[4 * c + 2][0] * scale, dk2[4 * c + 2][1] * scale), pack_bf2(dk2[4 * c + 3][0] * scale, dk2[4 * c + 3][1] * scale)};
        *(u32x4*)(drow + 2 * C + 8 * c) = (u32x4){pack_bf2(dv2[4 * c][0], dv2[4 * c][1]), pack_bf2(dv2[4 * c + 1][0], dv2[4 * c + 1][1]), pack_bf2(dv2[4 * c + 2][0], dv2[4 * c + 2][1]),
                                                  pack_bf2(dv2[4 * c + 3][0], dv2[4 * c + 3][1])};
      }
    }
  }
  VDK_WAVE_LDS_SYNC();
  if (row) {
    float* dst = dbias_part + (wid * WA_N + lane) * WA_N;
    for (int j = 0; j < WA_N; ++j) dst[j] = Bs[w][lane * WA_N + j];
  }
}

extern "C" {

int vdk_reduce_rows_f32(const float*, int64_t, int32_t, int64_t, float*, float, void*);

static int wa_check(const void* qkv, int64_t ld, int64_t windows, int32_t H, int32_t N, int32_t hd, const float* bias, const float* mask, int32_t nW, const char* who) {
  if (!qkv || !bias || windows <= 0 || H <= 0 || (ld & 7) || ld < 3 * H * hd) return vdk_fail(VDK_EINVAL, who);
  if (N != WA_N || hd != WA_HD) return vdk_fail(VDK_EUNSUPPORTED, "window attention: 7 x 7 windows (49 tokens) with head dim 32 (every timm swin_*_window7_224)");
  if (mask && (nW <= 0 || windows % nW)) return vdk_fail(VDK_EINVAL, who);
  return VDK_OK;
}

/* timm WindowAttention core (Swin): qkv bf16 [windows * 49, ld] (q | k | v thirds of 3 * H * 32 columns) -> o bf16 [windows * 49, ldo]; lse f32 [windows, H, 49] (may be NULL
 * for inference); bias f32 [H, 49, 49]; mask f32 [nW, 49, 49] or NULL (window w takes mask[w mod nW]) */
int vdk_window_attention_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, const float* bias, const float* mask, int32_t nW, int64_t windows, int32_t H, int32_t N,
                             int32_t hd, float scale, void* stream) {
  int rc = wa_check(qkv, ld, windows, H, N, hd, bias, mask, nW, "vdk_window_attention_fwd: bad argument");
  if (rc) return rc;
  if (!o || (ldo & 7) || ldo < H * hd) return vdk_fail(VDK_EINVAL, "vdk_window_attention_fwd: bad argument");
  const long items = (long)windows * H;
  long grid = (items + 3) / 4; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(window_attn_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ld, (bf16_t*)o, (long)ldo, lse, bias, mask, (int)nW,
                     items, (int)H, scale);
  return vdk_check_launch("vdk_window_attention_fwd");
}

int vdk_window_attention_bwd_workspace_bytes(int64_t windows, int32_t H, size_t* bytes) {
  if (!bytes || windows <= 0 || H <= 0) return vdk_fail(VDK_EINVAL, "vdk_window_attention_bwd_workspace_bytes: bad argument");
  long waves = 4096 / H * H; if (waves > windows * H) waves = windows * H; if (waves < H) waves = H;
  waves = (waves + WA_BW * H - 1) / (WA_BW * H) * (WA_BW * H);           // whole workgroups, whole head groups
  *bytes = (size_t)waves * WA_N * WA_N * 4;
  return VDK_OK;
}
/* backward: dqkv bf16 [windows * 49, ldd] (dq | dk | dv), dbias f32 [H, 49, 49] (summed over every window; overwritten).  ws: vdk_window_attention_bwd_workspace_bytes */
int vdk_window_attention_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, const float* bias, const float* mask, int32_t nW,
                             int64_t windows, int32_t H, int32_t N, int32_t hd, float scale, void* dqkv, int64_t ldd, float* dbias, void* ws, size_t ws_bytes, void* stream) {
  int rc = wa_check(qkv, ld, windows, H, N, hd, bias, mask, nW, "vdk_window_attention_bwd: bad argument");
  if (rc) return rc;
  if (!o || !dout || !lse || !dqkv || !dbias || (ldo & 7) || (ldd & 7) || ldd < 3 * H * hd) return vdk_fail(VDK_EINVAL, "vdk_window_attention_bwd: bad argument");
  size_t need = 0; vdk_window_attention_bwd_workspace_bytes(windows, H, &need);
  if (!ws || ws_bytes < need) return vdk_fail(VDK_EWORKSPACE, "vdk_window_attention_bwd: workspace too small");
  const long waves = (long)(need / ((size_t)WA_N * WA_N * 4));
  hipLaunchKernelGGL(window_attn_bwd_kernel, dim3((unsigned)(waves / WA_BW)), dim3(64 * WA_BW), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ld, (const bf16_t*)o, (const bf16_t*)dout,
                     (long)ldo, lse, bias, mask, (int)nW, (long)windows, (int)H, scale, (bf16_t*)dqkv, (long)ldd, (float*)ws);
  // partial row r belongs to head r mod H: a group of H consecutive rows IS one [H, N, N] tensor, and the sum over the groups (in group order) is d(bias)
  const long nn = (long)WA_N * WA_N;
  rc = vdk_reduce_rows_f32((const float*)ws, (int64_t)H * nn, (int32_t)(waves / H), (int64_t)H * nn, dbias, 1.0f, stream);
  if (rc) return rc;
  return vdk_check_launch("vdk_window_attention_bwd");
}

}  // extern "C"

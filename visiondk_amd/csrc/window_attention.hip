// window_attention.hip -- the attention step of timm's Swin Transformer (WindowAttention inside SwinTransformerBlock): the default backbone of both shipped
// configs of the reference (`timm-swin_base_patch4_window7_224`: configs/classification/pet.yaml:25, configs/faceX/cbir.yaml:26; built by timm.create_model in
// models/classifier/classify_model.py:49-54 and models/faceX/backbone/timm_wrapper.py:16-21).  Per (window, head):
//     S = (q * hd^-0.5) k^T + bias[head] (+ mask[window mod nW]);   P = softmax(S);   o = P v            N = 49 tokens (7 x 7), head dim 32
// qkv: bf16 [W * N, 3 C] rows in (window, token) order or reached through a row index, q | k | v thirds, head h at columns h * 32; bias f32 [H, N, N] (the relative-position table gathered once per
// step by the host side); mask f32 [nW, N, N] (0 / -100 of the shifted windows) or NULL.  Arithmetic as under the reference's autocast: bf16 operands, fp32 products and
// sums, softmax in fp32, P rounded to bf16 once as the left operand of P v, o rounded to bf16.
//
// Structure: 65 536 tiny problems per layer at batch 256 (49 x 49 x 32), one wave each, no workgroup barriers; see the kernel comment below.  (The first form of this
// file ran them on plain VALU FMAs with lane = query: 48.7 ms per swin_base step at batch 128 against 40.7 ms with the MFMA form, same box.)
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_attn_tile.h"

#define WA_N 49
#define WA_HD 32
#define WA_LOG2E 1.4426950408889634f

#define WA_BW 2                                             // waves per workgroup of the backward
#ifndef WA_BWD_MINW
#define WA_BWD_MINW 1                                       // waves per SIMD the backward is compiled for
#endif

// One wave per (window, head); the 49 tokens are padded to 64 = two 32-row MFMA tiles (v_mfma_f32_32x32x16_bf16, contraction over the head dim 32 = 2 steps).
// Every product is computed TRANSPOSED so that the softmax axis lies on registers and the query on the lane:
//     S^T[key][q] = K Q^T    (A = K rows, B = Q rows: 16-byte row fragments of the LDS tiles the operands are DMA'd into)     C layout: lane = q, registers = keys
//     O^T[d][q]   = V^T P^T  (A = V through the transposing LDS read, B = P^T: the C-layout registers packed to bf16)   C layout: lane = q, registers = d
// bias + mask arrive pre-arranged in that C layout (wa_prep_bias_kernel: one f32x4 per 4 registers, -inf on the padded keys), so the padding costs no compares.
// Backward, per (window, head): S^T and dP^T = V dO^T as above; dS^T = P^T (dP^T - D); dQ^T = K^T dS^T straight from the registers; dV^T = dO^T P and dK^T = Q^T dS contract
// over the QUERY, which lies on lanes: P (then dS) goes once through a [64][64] bf16 LDS tile and comes back through the transposing read with the key on the lane.
// 40 MFMAs and ~100 LDS instructions per item against 12.5 k FMAs per lane-row of the VALU form.  d(bias): the wave's dS^T sum over its windows stays in registers in the
// C layout; partials are reduced in a fixed order (vdk_reduce_rows_f32) and un-permuted by wa_unprep_dbias_kernel.
#define WA_FRAG 4096                                        // floats of one [64 q][64 keys] tile in fragment order: [qt][kt][lane][16]

// element (qt, kt, lane, r) of the fragment order <-> query 32 qt + (lane & 31), key 32 kt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__global__ __launch_bounds__(256) void wa_prep_bias_kernel(const float* __restrict__ bias, const float* __restrict__ mask, int nWm, int H, float* __restrict__ bm) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nWm * H * WA_FRAG) return;
  const int r = (int)(i & 15), lane = (int)((i >> 4) & 63), kt = (int)((i >> 10) & 1), qt = (int)((i >> 11) & 1);
  const long wh = i >> 12; const int h = (int)(wh % H); const long wm = wh / H;
  const int q = 32 * qt + (lane & 31), key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  float v = key < WA_N ? 0.f : -INFINITY;
  if (q < WA_N && key < WA_N) { v = bias[((long)h * WA_N + q) * WA_N + key]; if (mask) v += mask[(wm * WA_N + q) * WA_N + key]; }
  bm[i] = v;
}
// bias[h][q][key] = table[index(q, key)][h] (+ the shift mask) straight from the relative-position table [169][H], in the attention kernels' fragment order (timm gathers
// it per forward); the backward reads the prepared tile again.
// Every block's tile in ONE launch (the Swin engine's forward; one launch per block was 24 x 5 us at the launch floor); jobs travel in the kernel arguments
struct WaPrepBatch { VdkWaPrepJob job[32]; long first[33]; int n; };
__global__ __launch_bounds__(256) void wa_prep_table_batch_kernel(WaPrepBatch b) {
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= b.first[b.n]) return;
  int k = 0;
  while (k + 1 < b.n && gi >= b.first[k + 1]) ++k;
  const VdkWaPrepJob jb = b.job[k];
  const long i = gi - b.first[k];
  const int H = jb.H;
  const int r = (int)(i & 15), lane = (int)((i >> 4) & 63), kt = (int)((i >> 10) & 1), qt = (int)((i >> 11) & 1);
  const long wh = i >> 12; const int h = (int)(wh % H); const long wm = wh / H;
  const int q = 32 * qt + (lane & 31), key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  float v = key < WA_N ? 0.f : -INFINITY;
  if (q < WA_N && key < WA_N) {
    const int rel = (q / 7 - key / 7 + 6) * 13 + (q % 7 - key % 7 + 6);
    v = jb.table[rel * H + h];
    if (jb.mask) v += jb.mask[(wm * WA_N + q) * WA_N + key];
  }
  jb.bm[i] = v;
}
__global__ __launch_bounds__(256) void wa_unprep_dbias_kernel(const float* __restrict__ red, int H, float* __restrict__ dbias) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)H * WA_N * WA_N) return;
  const int key = (int)(i % WA_N); const long hq = i / WA_N; const int q = (int)(hq % WA_N); const long h = hq / WA_N;
  const int kk = key & 31, hi = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3), lane = (q & 31) + 32 * hi;
  dbias[i] = red[h * WA_FRAG + ((((q >> 5) * 2 + (key >> 5)) * 64 + lane) << 4) + r];
}

// Row traffic.  A lane moves 16 bytes of row 16 i + (lane >> 2), i = 0..3 (4 lanes per 64-byte head slice of a token row, 16 rows per instruction): the operands come
// in by LDS-DMA into [64][32] bf16 tiles (64-byte rows) and the outputs leave the same way round (wa_flush_rows).  Tokens >= 49 read token 48 (finite filler).
// rowidx (optional): token j of window w lives in tensor row rowidx[w * 49 + j] -- the (shifted) window partition as an index instead of gather copies either side of the
// attention.  (Row fragments straight from global memory are 16 bytes per lane from 32 different rows per instruction; this form and the staged stores below together
// measured 36.24 -> 35.96 ms per swin_base step on one box.  The kernels move a layer's qkv / dO / o / dqkv once and run within 1.8x of that byte floor, DESIGN.md.)
__device__ __forceinline__ void wa_rows(const int* __restrict__ rowidx, long win, int lane, long (&rr)[4]) {
  long j[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = 16 * i + (lane >> 2);
    j[i] = win * WA_N + (t < WA_N ? t : WA_N - 1);
  }
  if (rowidx) {      // ONE branch around four independent loads (a select per element made four load-wait pairs: four serial round trips per window)
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = rowidx[j[i]];
#pragma unroll
    for (int i = 0; i < 4; ++i) rr[i] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) rr[i] = j[i];
  }
}
__device__ __forceinline__ void wa_dma_rows(unsigned char* tile, const bf16_t* __restrict__ base, long ld, const long (&rr)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16_t* g = base + rr[i] * ld + 8 * (lane & 3);
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(g), VDK_LDS_PTR(tile + i * 1024), 16, 0, 0);
  }
}
// MFMA row fragments of such a tile: lane (row 32 t + l31, hi) -> the 16 bytes at columns 16 ks + 8 hi
__device__ __forceinline__ void wa_row_frags(const unsigned char* tile, int l31, int hi, s16x8 (&f)[2][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) f[t][ks] = *(const s16x8*)(tile + (32 * t + l31) * 64 + 32 * ks + 16 * hi);
}
// transposed fragment of a 64-byte-row tile: lane = column l31, slots 0..3 = rows t1 + 4 hi + {0..3}, slots 4..7 = the same + 8 (the contraction order of a C-layout
// tile packed by as_pack_b).  Rows r .. r+3 cover all 64 banks once, the two hi halves take the two passes a 512-byte read needs anyway: no swizzle required.
__device__ __forceinline__ s16x8 wa_tr32(const unsigned char* tile, int t1, int lane) {
  const int s = lane & 15, chalf = (lane >> 4) & 1, hi = lane >> 5;
  const unsigned char* p1 = tile + (t1 + 4 * hi + (s >> 2)) * 64 + 32 * chalf + 8 * (s & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1));
  s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1 + 8 * 64));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return r;
}
// C-layout tile (kt, qt) as bf16 -> the [64 q][64 keys] LDS tile in the AS layout (128-byte rows, 16-byte chunk ^ as_f(row)); f = the tile's two packed B fragments
__device__ __forceinline__ void wa_put_pt(unsigned char* tile, const s16x8 (&f)[2], int kt, int qt, int l31, int hi) {
  const int row = 32 * qt + l31;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const u32x4 u = *(const u32x4*)&f[s];
    *(u32x2*)(tile + row * AS_ROW + (((4 * kt + 2 * s) ^ as_f(row)) << 4) + 8 * hi) = (u32x2){u[0], u[1]};
    *(u32x2*)(tile + row * AS_ROW + (((4 * kt + 2 * s + 1) ^ as_f(row)) << 4) + 8 * hi) = (u32x2){u[2], u[3]};
  }
}
// Output rows.  A C-layout tile of a transposed product has lane = token row, registers = head-dim columns: stored straight from the registers that is 8 bytes per lane
// into 32 different rows per instruction.  The two tiles of an output go through a [64][32] bf16 staging tile in LDS instead (80-byte pitch: the 8-byte writes of 16
// consecutive rows fall on distinct bank pairs) and leave as 16-byte stores, 4 lanes per 64-byte row.
#define WA_ST_PITCH 80
template <int OF = 0>
__device__ __forceinline__ void wa_stage_t(unsigned char* st, const f32x16& x, float mul, int t, int l31, int hi) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *(u32x2*)(st + (32 * t + l31) * WA_ST_PITCH + (8 * g + 4 * hi) * 2) = (u32x2){pack_op2<OF>(x[4 * g] * mul, x[4 * g + 1] * mul), pack_op2<OF>(x[4 * g + 2] * mul, x[4 * g + 3] * mul)};
}
// rows 0..48 of the staging tile -> the tensor rows rr of wa_rows
__device__ __forceinline__ void wa_flush_rows(const unsigned char* st, const long (&rr)[4], bf16_t* __restrict__ base, long ld, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * i + (lane >> 2);
    if (row < WA_N) *(u32x4*)(base + rr[i] * ld + 8 * (lane & 3)) = *(const u32x4*)(st + row * WA_ST_PITCH + 16 * (lane & 3));
  }
}
template <int OF = 0>
__device__ __forceinline__ float wa_dot8(const s16x8& a, const s16x8& b) {
  const u32x4 ua = *(const u32x4*)&a, ub = *(const u32x4*)&b;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { s = fmaf(op_lo<OF>(ua[e]), op_lo<OF>(ub[e]), s); s = fmaf(op_hi<OF>(ua[e]), op_hi<OF>(ub[e]), s); }
  return s;
}

template <int OF /* operand format of qkv / o: VDK_OPF_BF16 | VDK_OPF_F16 */>
__global__ __launch_bounds__(256) void window_attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o, long ldo, float* __restrict__ lse,
                                                                   const float* __restrict__ bm, int nWm, long items, int H, float scale,
                                                                   const int* __restrict__ rowidx) {
  __shared__ __attribute__((aligned(16))) unsigned char Qt[4][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Kt[4][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[4][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char St[4][64 * WA_ST_PITCH];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = H * WA_HD;
  // (one global round trip per item: the next item's row indices and this item's bias tile travel with the operand DMAs, see the backward)
  const long item0 = (long)blockIdx.x * 4 + w, istep = (long)gridDim.x * 4;
  long rr_next[4];
  wa_rows(rowidx, (item0 < items ? item0 : 0) / H, lane, rr_next);
  for (long item = item0; item < items; item += istep) {
    const long win = item / H; const int h = (int)(item - win * H);
    const bf16_t* base = qkv + h * WA_HD;
    long rr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rr[i] = rr_next[i];
    VDK_WAVE_LDS_SYNC();                                  // the previous item's LDS readers are done
    wa_dma_rows(Qt[w], base, ld, rr, lane);
    wa_dma_rows(Kt[w], base + C, ld, rr, lane);
    wa_dma_rows(Vt[w], base + 2 * C, ld, rr, lane);
    f32x4 bq[2][2][4];                                    // [qt][kt][4 registers each]
    {
      const float* bmp0 = bm + ((win % nWm) * H + h) * WA_FRAG + lane * 16;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int g = 0; g < 4; ++g) bq[qt][kt][g] = *(const f32x4*)(bmp0 + (qt * 2 + kt) * 1024 + 4 * g);
    }
    { const long nx = item + istep; wa_rows(rowidx, (nx < items ? nx : item) / H, lane, rr_next); }
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the wave's DMAs have landed
    VDK_WAVE_LDS_SYNC();
    s16x8 qf[2][2], kf[2][2];
    wa_row_frags(Qt[w], l31, hi, qf);
    wa_row_frags(Kt[w], l31, hi, kf);
    s16x8 pf[2][2][2];                                    // [kt][qt][k-step]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16 sa[2];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        sa[kt] = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) sa[kt] = vdk_mfma32<OF>(kf[kt][ks], qf[qt][ks], sa[kt]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = bq[qt][kt][g];
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float a = fmaf(sa[kt][4 * g + e], scale, b[e]); sa[kt][4 * g + e] = a; mx = fmaxf(mx, a); }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = fast_exp2((sa[kt][r] - mx) * WA_LOG2E); sa[kt][r] = e; sum += e; }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[kt][r] *= inv;
        as_pack_b<OF>(sa[kt], pf[kt][qt]);                     // P rounded to bf16 once, as the operand of P v
      }
      if (lse && hi == 0 && 32 * qt + l31 < WA_N) lse[(win * H + h) * WA_N + 32 * qt + l31] = mx + logf(sum);
    }
    s16x8 vt[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) vt[kt][s] = wa_tr32(Vt[w], 32 * kt + 16 * s, lane);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16 oa = as_zero16();
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) oa = vdk_mfma32<OF>(vt[kt][s], pf[kt][qt][s], oa);
      wa_stage_t<OF>(St[w], oa, 1.0f, qt, l31, hi);
    }
    VDK_WAVE_LDS_SYNC();
    wa_flush_rows(St[w], rr, o + h * WA_HD, ldo, lane);
  }
}

// one wave walks the windows win = slot, slot + nslot, ... of ONE head (h = wave index mod H); dbias_part: f32 [waves][WA_FRAG] in fragment order
template <int OF>
__global__ __launch_bounds__(64 * WA_BW, WA_BWD_MINW) void window_attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                                          long ldo, const float* __restrict__ lse, const float* __restrict__ bm, int nWm, long nwin, int H,
                                                                          float scale, bf16_t* __restrict__ dqkv, long ldd, float* __restrict__ dbias_part,
                                                                          const int* __restrict__ rowidx) {
  __shared__ __attribute__((aligned(16))) unsigned char Kt[WA_BW][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Qt[WA_BW][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Gt[WA_BW][64 * 64];     // dO rows
  __shared__ __attribute__((aligned(16))) unsigned char Pt[WA_BW][64 * AS_ROW]; // P [q][key], then dS [q][key]
  __shared__ __attribute__((aligned(16))) unsigned char Vt[WA_BW][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char St[WA_BW][64 * WA_ST_PITCH];   // output staging
  __shared__ float Dl[WA_BW][64];                                                     // rowsum(dO * O)
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = H * WA_HD;
  const long wid = (long)blockIdx.x * WA_BW + w, nwave = (long)gridDim.x * WA_BW;
  const int h = (int)(wid % H);
  const long slot = wid / H, nslot = nwave / H;            // (the launcher makes the wave count a multiple of H)
  f32x16 dB[2][2];                                         // [kt][qt]
#pragma unroll
  for (int a = 0; a < 2; ++a) { dB[a][0] = as_zero16(); dB[a][1] = as_zero16(); }
  // One global-memory round trip per window: the row indices of the NEXT window are fetched with this window's operands (an index load in front of the DMAs was a second,
  // serial round trip), and the bias tile's 64 values per lane are loaded with them too instead of tile by tile in front of each softmax (four more).  The wave has its
  // SIMD to itself -- one wave per SIMD by LDS -- so the registers are there (round 4: 88 -> see DESIGN.md, stage 3 of swin_base at batch 128).
  long rr_next[4];
  wa_rows(rowidx, slot < nwin ? slot : 0, lane, rr_next);
  for (long win = slot; win < nwin; win += nslot) {
    const bf16_t* base = qkv + h * WA_HD;
    long rr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rr[i] = rr_next[i];
    VDK_WAVE_LDS_SYNC();                                  // the previous window's readers are done
    wa_dma_rows(Qt[w], base, ld, rr, lane);
    wa_dma_rows(Kt[w], base + C, ld, rr, lane);
    wa_dma_rows(Vt[w], base + 2 * C, ld, rr, lane);
    wa_dma_rows(Gt[w], dout + h * WA_HD, ldo, rr, lane);
    u32x4 orow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) orow[i] = *(const u32x4*)(o + h * WA_HD + rr[i] * ldo + 8 * (lane & 3));
    float l[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) l[qt] = 32 * qt + l31 < WA_N ? lse[(win * H + h) * WA_N + 32 * qt + l31] : INFINITY;     // padded queries: P = exp(-inf) = 0
    f32x4 bq[2][2][4];                                    // bias (+ mask) of this (window mod nWm, head) in the C layout: [qt][kt][4 registers each]
    {
      const float* bmp0 = bm + ((win % nWm) * H + h) * WA_FRAG + lane * 16;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int g = 0; g < 4; ++g) bq[qt][kt][g] = *(const f32x4*)(bmp0 + (qt * 2 + kt) * 1024 + 4 * g);
    }
    { const long nx = win + nslot; wa_rows(rowidx, nx < nwin ? nx : win, lane, rr_next); }
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the wave's DMAs have landed
    VDK_WAVE_LDS_SYNC();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                          // D = rowsum(dO * O) on the rounded tensors: 8 products per lane, 4 lanes per row
      const s16x8 gq = *(const s16x8*)(Gt[w] + i * 1024 + lane * 16);
      float d = wa_dot8<OF>(gq, *(const s16x8*)&orow[i]);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      if ((lane & 3) == 0) Dl[w][16 * i + (lane >> 2)] = d;
    }
    s16x8 qf[2][2], kf[2][2], vf[2][2], gf[2][2];
    wa_row_frags(Qt[w], l31, hi, qf);
    wa_row_frags(Kt[w], l31, hi, kf);
    wa_row_frags(Vt[w], l31, hi, vf);
    wa_row_frags(Gt[w], l31, hi, gf);
    VDK_WAVE_LDS_SYNC();
    const float D[2] = {Dl[w][l31], Dl[w][32 + l31]};
    s16x8 dsf[2][2][2];                                   // dS^T as B fragments [kt][qt][k-step]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x16 sa = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) sa = vdk_mfma32<OF>(kf[kt][ks], qf[qt][ks], sa);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dp = vdk_mfma32<OF>(vf[kt][ks], gf[qt][ks], dp);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = bq[qt][kt][g];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float p = fast_exp2((fmaf(sa[r], scale, b[e]) - l[qt]) * WA_LOG2E);
            const float ds = p * (dp[r] - D[qt]);          // d(loss)/dS: also the bias gradient of this (query, key)
            dB[kt][qt][r] += ds;
            sa[r] = p; dp[r] = ds;
          }
        }
        s16x8 pfr[2];
        as_pack_b<OF>(sa, pfr);
        wa_put_pt(Pt[w], pfr, kt, qt, l31, hi);
        as_pack_b<OF>(dp, dsf[kt][qt]);
      }
    VDK_WAVE_LDS_SYNC();
    bf16_t* dbase = dqkv + h * WA_HD;
    // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
    {
      s16x8 ktr[2][2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) ktr[kt][s] = wa_tr32(Kt[w], 32 * kt + 16 * s, lane);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x16 acc = as_zero16();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s = 0; s < 2; ++s) acc = vdk_mfma32<OF>(ktr[kt][s], dsf[kt][qt][s], acc);
        wa_stage_t<OF>(St[w], acc, scale, qt, l31, hi);
      }
      VDK_WAVE_LDS_SYNC();
      wa_flush_rows(St[w], rr, dbase, ldd, lane);
    }
    // dV^T[d][key] = sum_q dO^T[d][q] P[q][key]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16 acc = as_zero16();
#pragma unroll
      for (int qs = 0; qs < 4; ++qs)
        acc = vdk_mfma32<OF>(wa_tr32(Gt[w], 16 * qs, lane), as_tr_frag(Pt[w], 16 * qs, 32 * kt, lane), acc);
      if (kt == 0) VDK_WAVE_LDS_SYNC();                    // dQ has left the staging tile
      wa_stage_t<OF>(St[w], acc, 1.0f, kt, l31, hi);
    }
    VDK_WAVE_LDS_SYNC();                                  // P has been read: the tile takes dS
    wa_flush_rows(St[w], rr, dbase + 2 * C, ldd, lane);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) wa_put_pt(Pt[w], dsf[kt][qt], kt, qt, l31, hi);
    VDK_WAVE_LDS_SYNC();
    // dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16 acc = as_zero16();
#pragma unroll
      for (int qs = 0; qs < 4; ++qs)
        acc = vdk_mfma32<OF>(wa_tr32(Qt[w], 16 * qs, lane), as_tr_frag(Pt[w], 16 * qs, 32 * kt, lane), acc);
      if (kt == 0) VDK_WAVE_LDS_SYNC();                    // dV has left the staging tile
      wa_stage_t<OF>(St[w], acc, scale, kt, l31, hi);
    }
    VDK_WAVE_LDS_SYNC();
    wa_flush_rows(St[w], rr, dbase + C, ldd, lane);
  }
  float* dst = dbias_part + wid * WA_FRAG + lane * 16;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(dst + (qt * 2 + kt) * 1024 + 4 * g) = (f32x4){dB[kt][qt][4 * g], dB[kt][qt][4 * g + 1], dB[kt][qt][4 * g + 2], dB[kt][qt][4 * g + 3]};
}

// The same backward at TWO waves per SIMD (round 4).  The form above holds one window's whole working set at once -- 29.9 KB of LDS per wave and 256 + 34 registers: one
// wave per SIMD, nothing to hide a DMA or an LDS round trip behind, 3-4x the byte floor (swin_base stage 3 at batch 128: 88 us for 206 MB).  Here the key tiles go one at
// a time: for kt in {0, 1}: S^T, dP^T of the 32 keys -> P half tile [64 q][32 keys] (4 KB) -> dV^T of those keys -> dS over P's bytes -> dK^T of those keys, and dQ^T
// accumulates over kt in registers.  V is dead once its fragments are loaded: its tile becomes the output staging (49 rows x 80 B).  20 KB of LDS per wave, <= 256
// registers: 8 waves per CU.  Outputs leave per 32-key half.  Same arithmetic and summation order per element as the form above (bit-equal dq / dk / dv / d(bias)).
__device__ __forceinline__ void wa_put_ph(unsigned char* tile, const s16x8 (&f)[2], int qt, int l31, int hi) {      // C-layout (kt, qt) -> rows 32 qt + l31 of [64 q][32 keys]
  const int row = 32 * qt + l31;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const u32x4 u = *(const u32x4*)&f[s];
    *(u32x2*)(tile + row * 64 + 32 * s + 8 * hi) = (u32x2){u[0], u[1]};
    *(u32x2*)(tile + row * 64 + 32 * s + 16 + 8 * hi) = (u32x2){u[2], u[3]};
  }
}

// d(relative_position_bias_table)[r][h] = sum of d(bias)[h][pos] over the positions pos of the 49 x 49 map that read table entry r (uses[r][0..U), -1 = none), in list
// order: a gather per output, no atomics (timm gathers the table with relative_position_index; its backward is an index_add)
__global__ __launch_bounds__(256) void wa_table_grad_kernel(const float* __restrict__ dbias, const int* __restrict__ uses, int R, int U, int H, int NN, float* __restrict__ dtable) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * H) return;
  const int r = i / H, h = i - r * H;
  float s = 0.f;
  for (int u = 0; u < U; ++u) { const int pos = uses[r * U + u]; if (pos >= 0) s += dbias[(long)h * NN + pos]; }
  dtable[i] = s;
}

// ... and its gradient straight from the reduced d(bias) in fragment order: one wave per table entry, lane = one of the <= 49 positions that read it, a wave sum per head
__global__ __launch_bounds__(256) void wa_table_grad_frag_kernel(const float* __restrict__ red, const int* __restrict__ uses, int R, int U, int H, float* __restrict__ dtable) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const int pos = lane < U ? uses[r * U + lane] : -1;
  int off = -1;
  if (pos >= 0) {
    const int q = pos / WA_N, key = pos - q * WA_N;
    const int kk = key & 31, hi = (kk >> 2) & 1, rr = (kk & 3) + 4 * (kk >> 3), ln = (q & 31) + 32 * hi;
    off = ((((q >> 5) * 2 + (key >> 5)) * 64 + ln) << 4) + rr;
  }
  for (int h = 0; h < H; ++h) {
    const float v = wave_sum(off >= 0 ? red[(long)h * WA_FRAG + off] : 0.f);
    if (lane == 0) dtable[r * H + h] = v;
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
static size_t wa_bm_bytes(int32_t nW, int32_t H) { return (size_t)(nW > 0 ? nW : 1) * H * WA_FRAG * 4; }
#define WA_LAUNCH_BWD(OPF, GRID, ST, ...) do { if (OPF) hipLaunchKernelGGL(window_attn_bwd_mfma_kernel<VDK_OPF_F16>, GRID, dim3(64 * WA_BW), 0, ST, __VA_ARGS__); \
                                               else hipLaunchKernelGGL(window_attn_bwd_mfma_kernel<VDK_OPF_BF16>, GRID, dim3(64 * WA_BW), 0, ST, __VA_ARGS__); } while (0)
static long wa_bwd_waves(int64_t windows, int32_t H) {
  long waves = 4096 / H * H; if (waves > windows * H) waves = windows * H; if (waves < H) waves = H;
  return (waves + WA_BW * H - 1) / (WA_BW * H) * (WA_BW * H);             // whole workgroups, whole head groups
}
// bias (+ mask) -> the kernels' fragment order
static void wa_prep(const float* bias, const float* mask, int32_t nW, int32_t H, float* bm, hipStream_t st) {
  const int nWm = mask ? nW : 1;
  const long n = (long)nWm * H * WA_FRAG;
  hipLaunchKernelGGL(wa_prep_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bias, mask, nWm, (int)H, bm);
}

/* workspace of vdk_window_attention_fwd: the bias (+ mask) in the kernel's fragment order, (nW or 1) * H * 16 KB */
int vdk_window_attention_fwd_workspace_bytes(int32_t nW, int32_t H, size_t* bytes) {
  if (!bytes || nW < 0 || H <= 0) return vdk_fail(VDK_EINVAL, "vdk_window_attention_fwd_workspace_bytes: bad argument");
  *bytes = wa_bm_bytes(nW, H);
  return VDK_OK;
}
/* timm WindowAttention core (Swin): qkv bf16 [windows * 49, ld] (q | k | v thirds of 3 * H * 32 columns) -> o bf16 [windows * 49, ldo]; lse f32 [windows, H, 49] (may be NULL
 * for inference); bias f32 [H, 49, 49]; mask f32 [nW, 49, 49] or NULL (window w takes mask[w mod nW]); rowidx int32 [windows * 49] or NULL: token j of window w is row
 * rowidx[w * 49 + j] of qkv / o / dout / dqkv (a permutation: the cyclic shift + window partition of timm's block as an index, so the tensors stay in image order) */
int vdk_window_attention_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, const float* bias, const float* mask, int32_t nW, int64_t windows, int32_t H, int32_t N,
                             int32_t hd, float scale, const int32_t* rowidx, void* ws, size_t ws_bytes, void* stream) {
  int rc = wa_check(qkv, ld, windows, H, N, hd, bias, mask, nW, "vdk_window_attention_fwd: bad argument");
  if (rc) return rc;
  if (!o || (ldo & 7) || ldo < H * hd) return vdk_fail(VDK_EINVAL, "vdk_window_attention_fwd: bad argument");
  if (!ws || ws_bytes < wa_bm_bytes(mask ? nW : 0, H)) return vdk_fail(VDK_EWORKSPACE, "vdk_window_attention_fwd: workspace too small");
  const long items = (long)windows * H;
  long grid = (items + 3) / 4; if (grid > 4096) grid = 4096;
  wa_prep(bias, mask, nW, H, (float*)ws, (hipStream_t)stream);
  hipLaunchKernelGGL(window_attn_fwd_mfma_kernel<VDK_OPF_BF16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ld, (bf16_t*)o, (long)ldo, lse, (const float*)ws,
                     mask ? (int)nW : 1, items, (int)H, scale, (const int*)rowidx);
  return vdk_check_launch("vdk_window_attention_fwd");
}

/* workspace of vdk_window_attention_bwd: the prepared bias, one d(bias) partial per wave and their sum, all in fragment order */
int vdk_window_attention_bwd_workspace_bytes(int64_t windows, int32_t nW, int32_t H, size_t* bytes) {
  if (!bytes || windows <= 0 || nW < 0 || H <= 0) return vdk_fail(VDK_EINVAL, "vdk_window_attention_bwd_workspace_bytes: bad argument");
  *bytes = wa_bm_bytes(nW, H) + (size_t)(wa_bwd_waves(windows, H) + H) * WA_FRAG * 4;
  return VDK_OK;
}
/* backward: dqkv bf16 [windows * 49, ldd] (dq | dk | dv), dbias f32 [H, 49, 49] (summed over every window; overwritten).  ws: vdk_window_attention_bwd_workspace_bytes */
int vdk_window_attention_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, const float* bias, const float* mask, int32_t nW,
                             int64_t windows, int32_t H, int32_t N, int32_t hd, float scale, const int32_t* rowidx, void* dqkv, int64_t ldd, float* dbias, void* ws, size_t ws_bytes,
                             void* stream) {
  int rc = wa_check(qkv, ld, windows, H, N, hd, bias, mask, nW, "vdk_window_attention_bwd: bad argument");
  if (rc) return rc;
  if (!o || !dout || !lse || !dqkv || !dbias || (ldo & 7) || (ldd & 7) || ldd < 3 * H * hd) return vdk_fail(VDK_EINVAL, "vdk_window_attention_bwd: bad argument");
  size_t need = 0; vdk_window_attention_bwd_workspace_bytes(windows, mask ? nW : 0, H, &need);
  if (!ws || ws_bytes < need) return vdk_fail(VDK_EWORKSPACE, "vdk_window_attention_bwd: workspace too small");
  const long waves = wa_bwd_waves(windows, H);
  float* bm = (float*)ws;
  float* part = bm + wa_bm_bytes(mask ? nW : 0, H) / 4;
  float* red = part + waves * WA_FRAG;
  hipStream_t st = (hipStream_t)stream;
  wa_prep(bias, mask, nW, H, bm, st);
  WA_LAUNCH_BWD(0, dim3((unsigned)(waves / WA_BW)), st, (const bf16_t*)qkv, (long)ld, (const bf16_t*)o, (const bf16_t*)dout,
                (long)ldo, lse, (const float*)bm, mask ? (int)nW : 1, (long)windows, (int)H, scale, (bf16_t*)dqkv, (long)ldd, part, (const int*)rowidx);
  // partial row r belongs to head r mod H: a group of H consecutive rows IS one [H, 64 x 64] tensor, and the sum over the groups (in group order) is d(bias)
  rc = vdk_reduce_rows_f32(part, (int64_t)H * WA_FRAG, (int32_t)(waves / H), (int64_t)H * WA_FRAG, red, 1.0f, stream);
  if (rc) return rc;
  const long n = (long)H * WA_N * WA_N;
  hipLaunchKernelGGL(wa_unprep_dbias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)red, (int)H, dbias);
  return vdk_check_launch("vdk_window_attention_bwd");
}

/* backward of the bias gather bias[h][q][k] = table[index[q][k]][h]: dtable f32 [R, H] from dbias f32 [H, NN]; uses int32 [R, U]: the positions q * N + k whose index is r
 * (-1 padded), summed in list order */
int vdk_relpos_bias_table_grad(const float* dbias, const int32_t* uses, int32_t R, int32_t U, int32_t H, int32_t NN, float* dtable, void* stream) {
  if (!dbias || !uses || !dtable || R <= 0 || U <= 0 || H <= 0 || NN <= 0) return vdk_fail(VDK_EINVAL, "vdk_relpos_bias_table_grad: bad argument");
  hipLaunchKernelGGL(wa_table_grad_kernel, dim3((unsigned)((R * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dbias, (const int*)uses, (int)R, (int)U, (int)H, (int)NN, dtable);
  return vdk_check_launch("vdk_relpos_bias_table_grad");
}

}  // extern "C"

// ---- in-library forms for the Swin engine (csrc/swin_engine.hip): the bias tile is prepared ONCE per block and step, straight from the relative-position table, and kept
// for the backward; d(table) comes from the reduced fragment-order d(bias) in one launch (no un-permute, no [H, 49, 49] tensor in between)
size_t vdk_wa_bm_bytes(int32_t nW, int32_t H) { return wa_bm_bytes(nW, H); }
size_t vdk_wa_bwd_scratch_bytes(int64_t windows, int32_t H) { return (size_t)(wa_bwd_waves(windows, H) + H) * WA_FRAG * 4; }
int vdk_wa_prep_table_batch(const VdkWaPrepJob* jobs, int n, void* stream) {
  if (n < 0 || (n > 0 && !jobs)) return vdk_fail(VDK_EINVAL, "vdk_wa_prep_table_batch: bad argument");
  for (int i0 = 0; i0 < n; i0 += 32) {
    WaPrepBatch b; b.n = 0; long tot = 0;
    for (int i = i0; i < n && i < i0 + 32; ++i) {
      b.first[b.n] = tot;
      b.job[b.n++] = jobs[i];
      tot += (long)(jobs[i].mask ? jobs[i].nW : 1) * jobs[i].H * WA_FRAG;
    }
    b.first[b.n] = tot;
    if (tot > 0) hipLaunchKernelGGL(wa_prep_table_batch_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b);
  }
  return vdk_check_launch("vdk_wa_prep_table_batch");
}
int vdk_wa_fwd_bm(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, const float* bm, int32_t nWm, int64_t windows, int32_t H, float scale, const int32_t* rowidx, int opf,
                  void* stream) {
  const long items = (long)windows * H;
  long grid = (items + 3) / 4; if (grid > 4096) grid = 4096;
  if (opf) hipLaunchKernelGGL(window_attn_fwd_mfma_kernel<VDK_OPF_F16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ld, (bf16_t*)o, (long)ldo, lse, bm, (int)nWm,
                              items, (int)H, scale, (const int*)rowidx);
  else hipLaunchKernelGGL(window_attn_fwd_mfma_kernel<VDK_OPF_BF16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ld, (bf16_t*)o, (long)ldo, lse, bm, (int)nWm,
                          items, (int)H, scale, (const int*)rowidx);
  return vdk_check_launch("vdk_wa_fwd_bm");
}
int vdk_wa_bwd_bm(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, const float* bm, int32_t nWm, int64_t windows, int32_t H, float scale,
                  const int32_t* rowidx, void* dqkv, int64_t ldd, void* scratch, size_t scratch_bytes, const int32_t* uses, int32_t R, int32_t U, float* dtable, int opf, void* stream) {
  if (U > 64) return vdk_fail(VDK_EINVAL, "vdk_wa_bwd_bm: at most 64 uses per table entry");
  if (!scratch || scratch_bytes < vdk_wa_bwd_scratch_bytes(windows, H)) return vdk_fail(VDK_EWORKSPACE, "vdk_wa_bwd_bm: scratch too small");
  const long waves = wa_bwd_waves(windows, H);
  float* part = (float*)scratch;
  float* red = part + waves * WA_FRAG;
  hipStream_t st = (hipStream_t)stream;
  WA_LAUNCH_BWD(opf, dim3((unsigned)(waves / WA_BW)), st, (const bf16_t*)qkv, (long)ld, (const bf16_t*)o, (const bf16_t*)dout,
                (long)ldo, lse, bm, (int)nWm, (long)windows, (int)H, scale, (bf16_t*)dqkv, (long)ldd, part, (const int*)rowidx);
  const int rc = vdk_reduce_rows_f32(part, (int64_t)H * WA_FRAG, (int32_t)(waves / H), (int64_t)H * WA_FRAG, red, 1.0f, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(wa_table_grad_frag_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, (const float*)red, (const int*)uses, (int)R, (int)U, (int)H, dtable);
  return vdk_check_launch("vdk_wa_bwd_bm");
}

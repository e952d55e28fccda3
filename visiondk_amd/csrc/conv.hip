// conv.hip — the convolution side of hot path A for CNN backbones (timm ConvNeXt behind TimmWrapper,
// /root/reference models/faceX/backbone/timm_wrapper.py:16-37; configs/faceX/cbir.yaml:4-8).
//
// Everything is NHWC ("rows x channels", the same layout as the transformer's token rows), so ConvNeXt's pointwise convs, its
// 4x4/4 stem and its 2x2/2 downsamples are the existing MFMA GEMM, LayerNorm2d is the existing row LayerNorm, and what is left
// is HBM/VALU-bound:
//   * K6  depthwise 7x7 (ConvNeXtBlock.conv_dw, groups=C, padding 3): forward, input gradient (same kernel, flipped taps, with
//         the shortcut gradient added and a bf16 copy written for the next GEMM) and weight/bias gradient;
//   * space-to-depth / depth-to-space for the stride-2 2x2 convolutions (im2col-free GEMM operands);
//   * weight preparation: tap-major depthwise weights, (ky,kx,cin)-ordered 2x2 weights, layer-scale folded into fc2
//     (W2' = gamma (.) W2) and the chain rule back (dW2 = gamma (.) dW2', dgamma = <dW2', W2> + db2' b2).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "vdk_device.h"
#include "vdk_host.h"

#define DW_T 8          // output tile edge
#define DW_H (DW_T + 6)  // input tile edge (halo 3)
#define DW_CC 64        // channels per workgroup (weight gradient: one lane per channel)
#define DWF_CC 64       // channels per workgroup of the forward / input-gradient kernel (32 measured no better: the kernel is bound by its global -> LDS staging phase, not by occupancy)

// ------------------------------------------------------------------------------------ K6 forward / input gradient
// out[b,y,x,c] = bias[c] + res[b,y,x,c] + sum_{i,j} wt[(flip ? 48 - (7i+j) : 7i+j)][c] * in[b, y+i-3, x+j-3, c]   (zero padding)
// Output tile TH x TW (7 x 14 for ConvNeXt's 56/28/14 maps, 7 x 7 for the 7 x 7 map, 8 x 8 otherwise), 64 channels per workgroup.
// grid: (tiles_x * tiles_y * B, ceil(C / 64)); TW * 16 threads = TW columns x 16 channel quads; a thread owns a TH-row output strip of
// one column and one channel quad: for every horizontal tap it reads the TH + 6 inputs of its column once and feeds 7 x TH FMAs.
template <int TH, int TW>
__global__ __launch_bounds__(TW * (DWF_CC / 4)) void dwconv7_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                                                          const float* res, float* out /* may alias res */, bf16_t* __restrict__ outb, int B, int H,
                                                          int W, int C, int flip, int opf /* format of outb: VDK_OPF_BF16 | VDK_OPF_F16 */) {
  constexpr int IH = TH + 6, IW = TW + 6, NT = TW * (DWF_CC / 4);
  __shared__ __attribute__((aligned(16))) float xs[IH * IW * DWF_CC];
  __shared__ __attribute__((aligned(16))) float ws[49 * DWF_CC];            // 12 544 B
  const int tid = threadIdx.x;
  const int tx_n = (W + TW - 1) / TW, ty_n = (H + TH - 1) / TH;
  const int tile = blockIdx.x % (tx_n * ty_n), b = blockIdx.x / (tx_n * ty_n);
  const int y0 = (tile / tx_n) * TH, x0 = (tile % tx_n) * TW;
  const int c0 = blockIdx.y * DWF_CC;
  for (int i = tid; i < IH * IW * (DWF_CC / 4); i += NT) {
    const int cq = i % (DWF_CC / 4), p = i / (DWF_CC / 4), py = p / IW, px = p % IW;
    const int y = y0 + py - 3, x = x0 + px - 3, c = c0 + cq * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H && x >= 0 && x < W && c < C) v = *(const f32x4*)(in + (((long)b * H + y) * W + x) * C + c);
    *(f32x4*)(xs + p * DWF_CC + cq * 4) = v;
  }
  for (int i = tid; i < 49 * (DWF_CC / 4); i += NT) {
    const int cq = i % (DWF_CC / 4), t = i / (DWF_CC / 4), c = c0 + cq * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < C) v = *(const f32x4*)(wt + (long)(flip ? 48 - t : t) * C + c);
    *(f32x4*)(ws + t * DWF_CC + cq * 4) = v;
  }
  __syncthreads();
  const int cq = tid % (DWF_CC / 4), col = tid / (DWF_CC / 4), c = c0 + cq * 4;
  f32x4 acc[TH];
  {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && c < C) b4 = *(const f32x4*)(bias + c);
#pragma unroll
    for (int o = 0; o < TH; ++o) acc[o] = b4;
  }
#pragma unroll 1   // one horizontal tap at a time: unrolling all 7 makes hipcc hoist ~100 LDS reads and spill
  for (int j = 0; j < 7; ++j) {
    f32x4 wj[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) wj[i] = *(const f32x4*)(ws + (i * 7 + j) * DWF_CC + cq * 4);
#pragma unroll
    for (int r = 0; r < IH; ++r) {
      const f32x4 v = *(const f32x4*)(xs + (r * IW + col + j) * DWF_CC + cq * 4);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int o = r - i;
        if (o >= 0 && o < TH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(v[e], wj[i][e], acc[o][e]);
        }
      }
    }
  }
  const int x = x0 + col;
  if (x < W && c < C) {
#pragma unroll
    for (int o = 0; o < TH; ++o) {
      const int y = y0 + o;
      if (y < H) {
        const long off = (((long)b * H + y) * W + x) * C + c;
        f32x4 v = acc[o];
        if (res) { const f32x4 r4 = *(const f32x4*)(res + off); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
        if (out) *(f32x4*)(out + off) = v;
        if (outb) *(u32x2*)(outb + off) = opf ? (u32x2){pack_h2(v[0], v[1]), pack_h2(v[2], v[3])} : (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      }
    }
  }
}

// Row-streaming variant for ConvNeXt's square maps (W = 56 / 28 / 14 / 7): a PERSISTENT workgroup owns a channel chunk and walks over images; within an
// image it walks down the map in strips of 7 output rows, keeping the 13 input rows a strip needs in an LDS ring (6 of them are reused by the next strip).
// The rows the NEXT step needs -- the next strip's 7 new rows, or the first 13 rows of the next image at an image's last strip -- are loaded into registers
// before the strip's FMAs and written to the ring after them, so no HBM round trip is exposed between images (with 2 workgroups per CU at 79 KB of LDS a
// load -> barrier -> compute sequence per image left the CU waiting: 210 us against an 85 us byte floor at 14 x 14 x 512).
// Every input element is read from HBM once (the tiled kernel above re-reads its halo: 2.65x), x-halo columns are zeros.
// threads = W columns x CC/4 channel quads; a thread owns one column x one quad and produces the strip's 7 outputs of that column.
template <int W, int CC>
__global__ __launch_bounds__(W * (CC / 4), 2) void dwconv7_rows_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                                                                      const float* res, float* out /* may alias res */, bf16_t* __restrict__ outb, int B, int C,
                                                                      int flip, int opf /* format of outb: VDK_OPF_BF16 | VDK_OPF_F16 */) {
  constexpr int H = W, IW = W + 6, CQ = CC / 4, NT = W * CQ, RD = 13, NEW = 7;
  __shared__ __attribute__((aligned(16))) float ring[RD * IW * CC];
  __shared__ __attribute__((aligned(16))) float ws[49 * CC];
  const int tid = threadIdx.x;
  const int nchunk = (C + CC - 1) / CC;
  const int c0 = (blockIdx.x % nchunk) * CC;                             // gridDim.x is a multiple of nchunk: the chunk (and its taps) stay with the workgroup
  const int bstep = gridDim.x / nchunk;
  const int cq = tid % CQ, col = tid / CQ, c = c0 + cq * 4;
  for (int i = tid; i < 49 * CQ; i += NT) {
    const int q = i % CQ, t = i / CQ, cc = c0 + q * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cc < C) v = *(const f32x4*)(wt + (long)(flip ? 48 - t : t) * C + cc);
    *(f32x4*)(ws + t * CC + q * 4) = v;
  }
  for (int i = tid; i < RD * 6 * CQ; i += NT) {                          // x-halo columns: zero once, never written again
    const int q = i % CQ, h = (i / CQ) % 6, slot = i / (6 * CQ);
    *(f32x4*)(ring + (slot * IW + (h < 3 ? h : W + h)) * CC + q * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // a block of rows [y0, y0 + n): the thread moves its own (column, quad) element of every row -- one register per row, the address pattern of its outputs
  f32x4 pf[RD];
  auto issue = [&](int bb, int y0, int n) {
#pragma unroll
    for (int k = 0; k < RD; ++k) {
      const int y = y0 + k;
      if (k < n) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && c < C) v = *(const f32x4*)(in + (((long)bb * H + y) * W + col) * C + c);
        pf[k] = v;
      }
    }
  };
  auto commit = [&](int y0, int n) {                                     // row y lives in ring slot (y + 3) % 13
#pragma unroll
    for (int k = 0; k < RD; ++k)
      if (k < n) *(f32x4*)(ring + (((y0 + 3 + k) % RD) * IW + col + 3) * CC + cq * 4) = pf[k];
  };
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (bias && c < C) b4 = *(const f32x4*)(bias + c);
  int b = blockIdx.x / nchunk;
  if (b < B) { issue(b, -3, RD); commit(-3, RD); }
  __syncthreads();
  for (; b < B; b += bstep) {
    for (int s0 = 0; s0 < H; s0 += NEW) {
      // what the next step needs: rows s0 + 10 .. s0 + 16 of this image, or rows -3 .. 9 of the workgroup's next image
      const bool more = s0 + NEW < H, next_img = !more && b + bstep < B;
      if (more) issue(b, s0 + 10, NEW);
      else if (next_img) issue(b + bstep, -3, RD);
      f32x4 acc[NEW];
#pragma unroll
      for (int o = 0; o < NEW; ++o) acc[o] = b4;
#ifndef DW_RES_LATE
      // the shortcut rows of this strip are requested before the FMAs, not after them (cfg3 same-box A/B: 114.9 -> 113.6 ms per step; -DDW_RES_LATE is the old order)
      f32x4 rf[NEW];
#pragma unroll
      for (int o = 0; o < NEW; ++o) rf[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (res && c < C) {
#pragma unroll
        for (int o = 0; o < NEW; ++o) rf[o] = *(const f32x4*)(res + (((long)b * H + s0 + o) * W + col) * C + c);
      }
#endif
#pragma unroll 1
      for (int j = 0; j < 7; ++j) {
        f32x4 wj[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) wj[i] = *(const f32x4*)(ws + (i * 7 + j) * CC + cq * 4);
#pragma unroll
        for (int r = 0; r < RD; ++r) {                                   // input row s0 - 3 + r, ring slot (s0 + r) % 13
          const f32x4 v = *(const f32x4*)(ring + (((s0 + r) % RD) * IW + col + j) * CC + cq * 4);
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            const int o = r - i;
            if (o >= 0 && o < NEW) {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(v[e], wj[i][e], acc[o][e]);
            }
          }
        }
      }
      if (c < C) {
#pragma unroll
        for (int o = 0; o < NEW; ++o) {
          const long off = (((long)b * H + s0 + o) * W + col) * C + c;
          f32x4 v = acc[o];
#ifdef DW_RES_LATE
          if (res) { const f32x4 r4 = *(const f32x4*)(res + off); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
#else
          v[0] += rf[o][0]; v[1] += rf[o][1]; v[2] += rf[o][2]; v[3] += rf[o][3];
#endif
          if (out) *(f32x4*)(out + off) = v;
          if (outb) *(u32x2*)(outb + off) = opf ? (u32x2){pack_h2(v[0], v[1]), pack_h2(v[2], v[3])} : (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        }
      }
      if (more || next_img) {
        __syncthreads();                                                 // everybody is done reading the ring rows about to be replaced
        if (more) commit(s0 + 10, NEW); else commit(-3, RD);
        __syncthreads();
      }
    }
  }
}


// ------------------------------------------------------------------------------------ K6 weight / bias gradient
// dw[c][7i+j] = sum_{b,y,x} dy[b,y,x,c] * in[b, y+i-3, x+j-3, c];  db[c] = sum dy[b,y,x,c].
// grid: (S slices, ceil(C / 64)); 448 threads = 2 row halves x 7 vertical taps x 32 channel PAIRS; a workgroup walks its share of the (image, tile) list,
// stages the input tile (with halo) and the dy tile in LDS, and every thread keeps the 7 horizontal taps of its (channel pair, i) in registers: per output row
// TW dy pairs and TW + 6 input pairs (8-byte LDS reads) feed 7 TW packed FMAs (v_pk_fma_f32: two channels per instruction).  The first form -- one channel per
// lane, 4-byte LDS reads, scalar FMAs -- was bound by its instruction count (1850 per item and thread: 227 us at 14 x 14 x 512, batch 512, against an 85 us byte
// floor); this one issues half of them.  Each row half writes its own partial row: 2 S rows are combined by vdk_reduce_rows_f32.
template <int TH, int TW>
__global__ __launch_bounds__(448) void dwconv7_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dy, float* __restrict__ part, int B,
                                                            int H, int W, int C, int S) {
  constexpr int IH = TH + 6, IW = TW + 6;
  __shared__ __attribute__((aligned(16))) float xs[IH * IW * DW_CC];
  __shared__ __attribute__((aligned(16))) float ds[TH * TW * DW_CC];
  const int tid = threadIdx.x, cp = tid & 31, ti = (tid >> 5) % 7, rg = (tid >> 5) / 7;      // channel pair, vertical tap, row half (rows rg, rg + 2, ...)
  const int tx_n = (W + TW - 1) / TW, ty_n = (H + TH - 1) / TH;
  const long items = (long)B * tx_n * ty_n;
  const int c0 = blockIdx.y * DW_CC;
  vdk_f32x2 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = (vdk_f32x2){0.f, 0.f};
  vdk_f32x2 accb = {0.f, 0.f};
  // software pipeline: the next item's tile loads are issued into registers before this item's FMAs and written to LDS after them, so the HBM latency
  // hides under the compute (with one workgroup per CU at 92 KB of LDS, load -> barrier -> compute per item left the CU idle for a round trip per item)
  constexpr int XS_N = IH * IW * (DW_CC / 4), DS_N = TH * TW * (DW_CC / 4), PFX = (XS_N + 447) / 448, PFD = (DS_N + 447) / 448;
  f32x4 px[PFX], pd[PFD];
  auto fetch = [&](long it) {
    const int tile = (int)(it % (tx_n * ty_n)), b = (int)(it / (tx_n * ty_n));
    const int y0 = (tile / tx_n) * TH, x0 = (tile % tx_n) * TW;
#pragma unroll
    for (int k = 0; k < PFX; ++k) {
      const int i = tid + k * 448;
      const int cq = i & 15, p = i >> 4, py = p / IW, pxx = p % IW;
      const int y = y0 + py - 3, x = x0 + pxx - 3, c = c0 + cq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < XS_N && y >= 0 && y < H && x >= 0 && x < W && c < C) v = *(const f32x4*)(in + (((long)b * H + y) * W + x) * C + c);
      px[k] = v;
    }
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int i = tid + k * 448;
      const int cq = i & 15, p = i >> 4, py = p / TW, pxx = p % TW;
      const int y = y0 + py, x = x0 + pxx, c = c0 + cq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < DS_N && y < H && x < W && c < C) v = *(const f32x4*)(dy + (((long)b * H + y) * W + x) * C + c);
      pd[k] = v;
    }
  };
  if ((long)blockIdx.x < items) fetch(blockIdx.x);
  for (long it = blockIdx.x; it < items; it += S) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PFX; ++k) {
      const int i = tid + k * 448;
      if (i < XS_N) *(f32x4*)(xs + (i >> 4) * DW_CC + (i & 15) * 4) = px[k];
    }
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int i = tid + k * 448;
      if (i < DS_N) *(f32x4*)(ds + (i >> 4) * DW_CC + (i & 15) * 4) = pd[k];
    }
    __syncthreads();
    if (it + S < items) fetch(it + S);
#pragma unroll 1
    for (int h = rg; h < TH; h += 2) {
      vdk_f32x2 d[TW], xv[IW];
#pragma unroll
      for (int w = 0; w < TW; ++w) d[w] = *(const vdk_f32x2*)(ds + (h * TW + w) * DW_CC + 2 * cp);
#pragma unroll
      for (int w = 0; w < IW; ++w) xv[w] = *(const vdk_f32x2*)(xs + ((h + ti) * IW + w) * DW_CC + 2 * cp);
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int w = 0; w < TW; ++w) acc[j] = vdk_fma2(d[w], xv[w + j], acc[j]);
      if (ti == 0) {
#pragma unroll
        for (int w = 0; w < TW; ++w) accb += d[w];
      }
    }
  }
  const int c = c0 + 2 * cp;
  if (c < C) {     // (C % 4 == 0: a pair never straddles the end)
    float* p = part + ((long)blockIdx.x * 2 + rg) * ((long)C * 50);
#pragma unroll
    for (int j = 0; j < 7; ++j) { p[(long)c * 49 + ti * 7 + j] = acc[j][0]; p[(long)(c + 1) * 49 + ti * 7 + j] = acc[j][1]; }
    if (ti == 0) { p[(long)C * 49 + c] = accb[0]; p[(long)C * 49 + c + 1] = accb[1]; }
  }
}

// ------------------------------------------------------------------------------------ stride-2 2x2 conv operands
// space-to-depth: out[(b, y/2, x/2)][(2*(y&1) + (x&1)) * C + c] = in[(b, y, x)][c]   (bf16, 16-byte chunks); inverse = depth-to-space
__global__ __launch_bounds__(256) void s2d2_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int H, int W, int C, int inverse) {
  const long n = (long)B * H * W * (C / 8);
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= n) return;
  const int c8 = (int)(id % (C / 8));
  const long p = id / (C / 8);
  const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
  const long a = p * C + c8 * 8;                                                                             // NHWC element
  const long d = ((((long)b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1))) * C + c8 * 8;   // depth-major element
  if (inverse) *(u32x4*)(out + a) = *(const u32x4*)(in + d);
  else *(u32x4*)(out + d) = *(const u32x4*)(in + a);
}

// ------------------------------------------------------------------------------------ global average pool on f32 NHWC rows (timm SelectAdaptivePool2d('avg'))
// out[b][c] = mean over the HW rows of image b; backward spreads dpool / HW over the rows (f32 and / or bf16 copy)
__global__ __launch_bounds__(256) void avgpool_rows_f32_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int HW, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * C) return;
  const int c = (int)(i % C), b = (int)(i / C);
  const float* p = in + (long)b * HW * C + c;
  float s = 0.f;
  for (int r = 0; r < HW; ++r) s += p[(long)r * C];
  out[i] = s / (float)HW;
}
__global__ __launch_bounds__(256) void avgpool_rows_f32_bwd_kernel(const float* __restrict__ dpool, float* __restrict__ dmap, bf16_t* __restrict__ dmapb, int B, int HW,
                                                                   int C, int opf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * HW * C) return;
  const int c = (int)(i % C), b = (int)(i / ((long)HW * C));
  const float g = dpool[(long)b * C + c] / (float)HW;
  if (dmap) dmap[i] = g;
  if (dmapb) dmapb[i] = opf ? f2op<VDK_OPF_F16>(g) : f2bf(g);
}

// ------------------------------------------------------------------------------------ weight preparation
// depthwise weight [C][49] -> tap-major [49][C]
__global__ __launch_bounds__(256) void dw_weight_prep_kernel(const float* __restrict__ w, float* __restrict__ wt, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * 49) return;
  const int t = i / C, c = i % C;
  wt[i] = w[c * 49 + t];
}
// 2x2 conv weight [Co][Ci][2][2] -> wb bf16 [Co][4*Ci] with k = (2*ky + kx) * Ci + ci, and its transpose wtb bf16 [4*Ci][Co]
__global__ __launch_bounds__(256) void conv2x2_weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ wb, bf16_t* __restrict__ wtb, int Co,
                                                                  int Ci, int opf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)Co * Ci * 4) return;
  const int q = (int)(i & 3);
  const long rest = i >> 2;
  const int ci = (int)(rest % Ci), co = (int)(rest / Ci);
  const bf16_t v = opf ? f2op<VDK_OPF_F16>(w[i]) : f2bf(w[i]);
  const long k = (long)q * Ci + ci;
  wb[(long)co * 4 * Ci + k] = v;
  wtb[k * Co + co] = v;
}
// gradient back: dWp f32 [Co][4*Ci] (k order above) -> dW f32 [Co][Ci][2][2]
__global__ __launch_bounds__(256) void conv2x2_wgrad_unpermute_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)Co * Ci * 4) return;
  const int q = (int)(i & 3);
  const long rest = i >> 2;
  const int ci = (int)(rest % Ci), co = (int)(rest / Ci);
  dw[i] = dwp[(long)co * 4 * Ci + (long)q * Ci + ci];
}
// layer scale folded into fc2:  W2p = gamma (.) W2 (bf16 [C][M]),  W2pt = W2p^T (bf16 [M][C]),  b2p = gamma (.) b2 (f32)
__global__ __launch_bounds__(256) void lscale_weight_prep_kernel(const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ gamma,
                                                                 bf16_t* __restrict__ w2p, bf16_t* __restrict__ w2pt, float* __restrict__ b2p, int C, int M) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < C) b2p[i] = gamma[i] * b2[i];
  if (i >= (long)C * M) return;
  const int k = (int)(i % M), c = (int)(i / M);
  const bf16_t v = f2bf(gamma[c] * w2[i]);
  w2p[i] = v;
  w2pt[(long)k * C + c] = v;
}
// Every block's weight preparation in ONE launch (the engine's refresh after an optimizer step was 2 launches per block: ~75 launches of 5-25 us for ConvNeXt-B): job j
// covers workgroups [first[j], first[j + 1]); element i of a job does the layer-scale fold of W2[i] and, while i < 49 C, the tap-major copy of the depthwise weight.
struct CnPrepBatch { CnPrepJob job[40]; int first[41]; int n; };
// fp16 operands (jb.rs != NULL): gamma is NOT folded into the forward operand -- gamma (1e-6 at timm's init) times a weight of ~0.02 is below fp16's smallest subnormal -- so
//   w2p  = fp16(W2)                      forward B operand; the fc2 GEMM applies gamma as VdkGemmDesc.col_scale, b2p = gamma (.) b2 as its bias
//   w2pt = fp16((gamma r) (.) W2)^T      backward B operand, r = 2^-floor(log2 max|gamma|) (max |gamma r| in [1, 2)): the branch's gradients du, dh live at r times their
//                                        true size in fp16 and return through rs[1] = 1 / r (LayerNorm backward's dy_scale, the fc1 gradients' final scaling)
//   rs   = {r, 1 / r}
__global__ __launch_bounds__(256) void cn_prep_batch_kernel(CnPrepBatch b) {
  __shared__ float redm[4];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first[j + 1]) ++j;
  const CnPrepJob jb = b.job[j];
  const long i = (long)((int)blockIdx.x - b.first[j]) * 256 + threadIdx.x;
  const int C = jb.C, M = jb.M;
  float r = 1.0f;
  if (jb.rs) {       // (uniform per workgroup: every workgroup of a job finds the same maximum, C reads from the L2)
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(jb.gamma[c]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;      // floor(log2 m) for normal m; zero / subnormal -> -127
    if (!(m > 0.f) || !(m < 3.0e38f)) e = 0;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    r = __uint_as_float((unsigned)(127 - e) << 23);
    if (i == 0) { jb.rs[0] = r; jb.rs[1] = __uint_as_float((unsigned)(127 + e) << 23); }
  }
  if (i < (long)C * 49) { const int t = (int)(i / C), c = (int)(i % C); jb.dwt[i] = jb.dw_w[c * 49 + t]; }
  if (i < C) jb.b2p[i] = jb.gamma[i] * jb.b2[i];
  if (i >= (long)C * M) return;
  const int k = (int)(i % M), c = (int)(i / M);
  if (jb.rs) {
    jb.w2p[i] = f2op<VDK_OPF_F16>(jb.w2[i]);
    jb.w2pt[(long)k * C + c] = f2op<VDK_OPF_F16>((jb.gamma[c] * r) * jb.w2[i]);
    return;
  }
  const bf16_t v = f2bf(jb.gamma[c] * jb.w2[i]);
  jb.w2p[i] = v;
  jb.w2pt[(long)k * C + c] = v;
}
// x[i] *= s[0] (or / s[0]): a gradient region that was kept at a power-of-two scale returns to its true size (ConvNeXt's fc1 gradients under fp16 operands), and the
// loss scale enters a gradient tensor at the boundary of the fp16 graph (FaceTrainStep: d(loss)/d(embedding) x GradScaler's scale, engine/procedure/train.py:205)
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, long n, const float* __restrict__ s, int reciprocal) {
  const float f = reciprocal ? 1.0f / s[0] : s[0];
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) { f32x4 v = *(f32x4*)(x + i); v[0] *= f; v[1] *= f; v[2] *= f; v[3] *= f; *(f32x4*)(x + i) = v; }
  else for (long j = i; j < n; ++j) x[j] *= f;
}
// in-library: n <= 40 blocks per call
int vdk_convnext_prep_blocks(const CnPrepJob* jobs, int n, void* stream) {
  for (int i0 = 0; i0 < n; i0 += 40) {
    CnPrepBatch b; b.n = 0;
    int blocks = 0;
    for (int i = i0; i < n && i < i0 + 40; ++i) {
      b.first[b.n] = blocks; b.job[b.n++] = jobs[i];
      const long span = (long)jobs[i].C * (jobs[i].M > 49 ? jobs[i].M : 49);      // the longer of the two index ranges
      blocks += (int)((span + 255) / 256);
    }
    b.first[b.n] = blocks;
    if (blocks > 0) hipLaunchKernelGGL(cn_prep_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
  }
  return vdk_check_launch("vdk_convnext_prep_blocks");
}

// chain rule back through the fold (one wave per output channel c):
//   dW2[c][k] = gamma[c] dW2p[c][k];  db2[c] = gamma[c] db2p[c];  dgamma[c] = sum_k dW2p[c][k] W2[c][k] + db2p[c] b2[c]
__global__ __launch_bounds__(256) void lscale_grad_kernel(const float* __restrict__ dw2p, const float* __restrict__ db2p, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, const float* __restrict__ gamma, float* __restrict__ dw2,
                                                          float* __restrict__ db2, float* __restrict__ dgamma, int C, int M) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const float g = gamma[c];
  float s = 0.f;
  for (int k = lane; k < M; k += 64) {
    const float d = dw2p[(long)c * M + k];
    s = fmaf(d, w2[(long)c * M + k], s);
    dw2[(long)c * M + k] = g * d;
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float d = db2p[c];
    db2[c] = g * d;
    dgamma[c] = fmaf(d, b2[c], s);
  }
}

extern "C" {

// persistent grid of the row-streaming kernel: about two workgroups per CU (their LDS footprint), a whole number of images per channel chunk
static unsigned dw_rows_grid(int B, int nchunk) {
  int per_chunk = (2 * 256 + nchunk - 1) / nchunk;
  if (const char* e = getenv("VDK_DW_ROWS_PER_CHUNK")) { const int v = atoi(e); if (v > 0) per_chunk = v; }   // tests: force several images per workgroup
  if (per_chunk > B) per_chunk = B;
  if (per_chunk < 1) per_chunk = 1;
  return (unsigned)per_chunk * (unsigned)nchunk;
}
int vdk_dwconv7_fwd_16(const float* in, const float* wt, const float* bias, const float* res, float* out, void* out_bf16, int32_t B, int32_t H, int32_t W,
                       int32_t C, int32_t flip, int32_t opf, void* stream) {
  if (!in || !wt || (!out && !out_bf16) || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return vdk_fail(VDK_EINVAL, "vdk_dwconv7_fwd: bad argument (C % 4 == 0)");
  const unsigned cy = (unsigned)((C + DWF_CC - 1) / DWF_CC);
#define DW_LAUNCH(TH, TW)                                                                                                                          \
  hipLaunchKernelGGL((dwconv7_kernel<TH, TW>), dim3((unsigned)(((W + TW - 1) / TW) * ((H + TH - 1) / TH)) * (unsigned)B, cy), dim3(TW * (DWF_CC / 4)), 0, \
                     (hipStream_t)stream, in, wt, bias, res, out, (bf16_t*)out_bf16, (int)B, (int)H, (int)W, (int)C, (int)flip, (int)opf)
#define DW_ROWS(WW, CCC)                                                                                                                             \
  hipLaunchKernelGGL((dwconv7_rows_kernel<WW, CCC>), dim3(dw_rows_grid(B, (C + CCC - 1) / CCC)), dim3(WW * (CCC / 4)), 0, (hipStream_t)stream, in, wt, bias, \
                     res, out, (bf16_t*)out_bf16, (int)B, (int)C, (int)flip, (int)opf)
  if (H == W && W == 56) DW_ROWS(56, 32);      // 32 channels = one 128-byte line per pixel (with 16 the two halves of a line went to different XCDs: 669 -> 640 us)
  else if (H == W && W == 28) DW_ROWS(28, 32);
  else if (H == W && W == 14) DW_ROWS(14, 64);
  else if (H == W && W == 7) DW_ROWS(7, 128);
  else if (H % 7 == 0 && W % 14 == 0) DW_LAUNCH(7, 14);
  else if (H % 7 == 0 && W % 7 == 0) DW_LAUNCH(7, 7);
  else DW_LAUNCH(8, 8);
#undef DW_LAUNCH
#undef DW_ROWS
  return vdk_check_launch("vdk_dwconv7_fwd");
}
int vdk_dwconv7_fwd(const float* in, const float* wt, const float* bias, const float* res, float* out, void* out_bf16, int32_t B, int32_t H, int32_t W,
                    int32_t C, int32_t flip, void* stream) {
  return vdk_dwconv7_fwd_16(in, wt, bias, res, out, out_bf16, B, H, W, C, flip, VDK_OPF_BF16, stream);
}

static void dw_wgrad_tile(int H, int W, int* th, int* tw) {
  if (H % 14 == 0 && W % 14 == 0) { *th = 14; *tw = 14; }      // 152 KB of LDS, one workgroup per CU: a 14 x 14 map is one tile, no halo re-reads
  else if (H % 7 == 0 && W % 14 == 0) { *th = 7; *tw = 14; }
  else if (H % 7 == 0 && W % 7 == 0) { *th = 7; *tw = 7; }
  else { *th = 8; *tw = 8; }
}
static int dw_slices(int B, int H, int W, int C) {
  int th, tw; dw_wgrad_tile(H, W, &th, &tw);
  const long items = (long)B * ((W + tw - 1) / tw) * ((H + th - 1) / th);
  long s = (th == 14 ? 256 : 1024) / ((C + DW_CC - 1) / DW_CC);   // 14 x 14 tiles: one resident workgroup per CU, several items each (pipelined)
  if (s > items) s = items;
  if (s < 1) s = 1;
  return (int)s;
}
int vdk_dwconv7_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, size_t* bytes) {
  if (!bytes || B <= 0 || H <= 0 || W <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_dwconv7_wgrad_workspace_bytes: bad argument");
  *bytes = (size_t)2 * dw_slices(B, H, W, C) * C * 50 * 4;      // two partial rows (row halves) per slice
  return VDK_OK;
}
int vdk_reduce_rows_f32(const float*, int64_t, int32_t, int64_t, float*, float, void*);
/* dw f32 [C][49] (timm conv_dw.weight [C,1,7,7]) and db f32 [C]; ws: vdk_dwconv7_wgrad_workspace_bytes */
static int dw_wgrad_impl(const float* in, const float* dy, float* dw, float* db, int32_t B, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, void* stream,
                         VdkReduceJob* job);
int vdk_dwconv7_wgrad(const float* in, const float* dy, float* dw, float* db, int32_t B, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes,
                      void* stream) {
  return dw_wgrad_impl(in, dy, dw, db, B, H, W, C, ws, ws_bytes, stream, nullptr);
}
}  // extern "C"
int vdk_dwconv7_wgrad_deferred(const float* in, const float* dy, float* dw, float* db, int32_t B, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, void* stream,
                               VdkReduceJob* job) {
  if (!job || db != dw + (size_t)C * 49) return vdk_fail(VDK_EINVAL, "vdk_dwconv7_wgrad_deferred: needs a job slot and db == dw + 49 C");
  return dw_wgrad_impl(in, dy, dw, db, B, H, W, C, ws, ws_bytes, stream, job);
}
extern "C" {
static int dw_wgrad_impl(const float* in, const float* dy, float* dw, float* db, int32_t B, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, void* stream,
                         VdkReduceJob* job) {
  if (!in || !dy || !dw || !db || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return vdk_fail(VDK_EINVAL, "vdk_dwconv7_wgrad: bad argument (C % 4 == 0)");
  const int S = dw_slices(B, H, W, C);
  if (!ws || ws_bytes < (size_t)2 * S * C * 50 * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_dwconv7_wgrad: workspace too small");
  int th, tw; dw_wgrad_tile(H, W, &th, &tw);
  const dim3 grid((unsigned)S, (unsigned)((C + DW_CC - 1) / DW_CC));
  if (th == 14) hipLaunchKernelGGL((dwconv7_wgrad_kernel<14, 14>), grid, dim3(448), 0, (hipStream_t)stream, in, dy, (float*)ws, (int)B, (int)H, (int)W, (int)C, S);
  else if (tw == 14) hipLaunchKernelGGL((dwconv7_wgrad_kernel<7, 14>), grid, dim3(448), 0, (hipStream_t)stream, in, dy, (float*)ws, (int)B, (int)H, (int)W, (int)C, S);
  else if (tw == 7) hipLaunchKernelGGL((dwconv7_wgrad_kernel<7, 7>), grid, dim3(448), 0, (hipStream_t)stream, in, dy, (float*)ws, (int)B, (int)H, (int)W, (int)C, S);
  else hipLaunchKernelGGL((dwconv7_wgrad_kernel<8, 8>), grid, dim3(448), 0, (hipStream_t)stream, in, dy, (float*)ws, (int)B, (int)H, (int)W, (int)C, S);
  if (job) { *job = VdkReduceJob{(float*)ws, (long)C * 50, 2 * S, (long)C * 50, dw, 1.0f}; return vdk_check_launch("vdk_dwconv7_wgrad"); }
  if (db == dw + (size_t)C * 49)     // conv_dw.weight / conv_dw.bias of the flat gradient buffer: the partial rows [49 C | C] reduce in one launch
    return vdk_reduce_rows_f32((const float*)ws, (int64_t)C * 50, 2 * S, (int64_t)C * 50, dw, 1.0f, stream);
  int rc = vdk_reduce_rows_f32((const float*)ws, (int64_t)C * 50, 2 * S, (int64_t)C * 49, dw, 1.0f, stream);
  if (rc) return rc;
  return vdk_reduce_rows_f32((const float*)ws + (size_t)C * 49, (int64_t)C * 50, 2 * S, C, db, 1.0f, stream);
}

int vdk_space_to_depth2_bf16(const void* in, void* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t inverse, void* stream) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) return vdk_fail(VDK_EINVAL, "vdk_space_to_depth2_bf16: bad argument (H, W even; C % 8 == 0)");
  const long n = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(s2d2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, (int)B, (int)H, (int)W,
                     (int)C, (int)inverse);
  return vdk_check_launch("vdk_space_to_depth2_bf16");
}

int vdk_avgpool_rows_f32_fwd(const float* in, float* out, int32_t B, int32_t HW, int32_t C, void* stream) {
  if (!in || !out || B <= 0 || HW <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_avgpool_rows_f32_fwd: bad argument");
  hipLaunchKernelGGL(avgpool_rows_f32_fwd_kernel, dim3((unsigned)(((long)B * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, (int)B, (int)HW, (int)C);
  return vdk_check_launch("vdk_avgpool_rows_f32_fwd");
}
int vdk_avgpool_rows_f32_bwd_16(const float* dpool, float* dmap, void* dmap16, int32_t B, int32_t HW, int32_t C, int32_t opf, void* stream) {
  if (!dpool || (!dmap && !dmap16) || B <= 0 || HW <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_avgpool_rows_f32_bwd: bad argument");
  hipLaunchKernelGGL(avgpool_rows_f32_bwd_kernel, dim3((unsigned)(((long)B * HW * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dpool, dmap, (bf16_t*)dmap16,
                     (int)B, (int)HW, (int)C, (int)opf);
  return vdk_check_launch("vdk_avgpool_rows_f32_bwd");
}
int vdk_avgpool_rows_f32_bwd(const float* dpool, float* dmap, void* dmap_bf16, int32_t B, int32_t HW, int32_t C, void* stream) {
  return vdk_avgpool_rows_f32_bwd_16(dpool, dmap, dmap_bf16, B, HW, C, VDK_OPF_BF16, stream);
}

int vdk_dwconv7_weight_prep(const float* w, float* wt, int32_t C, void* stream) {
  if (!w || !wt || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_dwconv7_weight_prep: bad argument");
  hipLaunchKernelGGL(dw_weight_prep_kernel, dim3((unsigned)((C * 49 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wt, (int)C);
  return vdk_check_launch("vdk_dwconv7_weight_prep");
}
int vdk_conv2x2_weight_prep_16(const float* w, void* wb, void* wtb, int32_t Co, int32_t Ci, int32_t opf, void* stream) {
  if (!w || !wb || !wtb || Co <= 0 || Ci <= 0) return vdk_fail(VDK_EINVAL, "vdk_conv2x2_weight_prep: bad argument");
  hipLaunchKernelGGL(conv2x2_weight_prep_kernel, dim3((unsigned)(((long)Co * Ci * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wb, (bf16_t*)wtb,
                     (int)Co, (int)Ci, (int)opf);
  return vdk_check_launch("vdk_conv2x2_weight_prep");
}
int vdk_conv2x2_weight_prep(const float* w, void* wb, void* wtb, int32_t Co, int32_t Ci, void* stream) {
  return vdk_conv2x2_weight_prep_16(w, wb, wtb, Co, Ci, VDK_OPF_BF16, stream);
}
int vdk_conv2x2_wgrad_unpermute(const float* dwp, float* dw, int32_t Co, int32_t Ci, void* stream) {
  if (!dwp || !dw || Co <= 0 || Ci <= 0) return vdk_fail(VDK_EINVAL, "vdk_conv2x2_wgrad_unpermute: bad argument");
  hipLaunchKernelGGL(conv2x2_wgrad_unpermute_kernel, dim3((unsigned)(((long)Co * Ci * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dwp, dw, (int)Co, (int)Ci);
  return vdk_check_launch("vdk_conv2x2_wgrad_unpermute");
}
int vdk_layerscale_weight_prep(const float* w2, const float* b2, const float* gamma, void* w2p, void* w2pt, float* b2p, int32_t C, int32_t M, void* stream) {
  if (!w2 || !b2 || !gamma || !w2p || !w2pt || !b2p || C <= 0 || M < C) return vdk_fail(VDK_EINVAL, "vdk_layerscale_weight_prep: bad argument");
  hipLaunchKernelGGL(lscale_weight_prep_kernel, dim3((unsigned)(((long)C * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w2, b2, gamma, (bf16_t*)w2p,
                     (bf16_t*)w2pt, b2p, (int)C, (int)M);
  return vdk_check_launch("vdk_layerscale_weight_prep");
}
int vdk_scale_dev_f32(float* x, int64_t n, const float* scale, int32_t reciprocal, void* stream) {
  if (!x || !scale || n <= 0 || ((size_t)x & 15)) return vdk_fail(VDK_EINVAL, "vdk_scale_dev_f32: bad argument (x 16-byte aligned)");
  hipLaunchKernelGGL(scale_dev_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n, scale, (int)reciprocal);
  return vdk_check_launch("vdk_scale_dev_f32");
}
int vdk_layerscale_grad(const float* dw2p, const float* db2p, const float* w2, const float* b2, const float* gamma, float* dw2, float* db2, float* dgamma,
                        int32_t C, int32_t M, void* stream) {
  if (!dw2p || !db2p || !w2 || !b2 || !gamma || !dw2 || !db2 || !dgamma || C <= 0 || M <= 0) return vdk_fail(VDK_EINVAL, "vdk_layerscale_grad: bad argument");
  hipLaunchKernelGGL(lscale_grad_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dw2p, db2p, w2, b2, gamma, dw2, db2, dgamma, (int)C, (int)M);
  return vdk_check_launch("vdk_layerscale_grad");
}

}  // extern "C"

// gemm.hip — K2: C[M,N] = A[M,K] * B[N,K]^T on the bf16 MFMA pipe (fp32 accumulate), the one GEMM
// behind every Linear of the hot path (timm Attention.qkv/proj, Mlp.fc1/fc2, PatchEmbed.proj as a
// k=stride conv, head; reference call sites models/classifier/classify_model.py:49-54 ->
// timm vision_transformer; F.linear fwd = NT, dgrad = NT against a transposed weight copy, wgrad = NT
// on transposed activations).
//
// gfx950 structure: 128x128x64 workgroup tile, 4 waves (2x2) x (2x2) v_mfma_f32_32x32x16_bf16 per
// k-step, global->VGPR->LDS register staging (next tile's loads in flight under the MFMAs), LDS
// double-buffered with ONE barrier per k-tile, rows padded to 144 B so every ds_read_b128 lane
// group hits 16 distinct 16-B slots, XCD-aware bijective tile swizzle, epilogue staged through LDS
// for 16-B coalesced stores with bias / exact-erf GELU / dGELU / fp32 residual fused in.
#include <hip/hip_runtime.h>
#ifndef VDK_EMU_NO_HIP_EXT
#include <hip/hip_ext.h>
#endif
#include <cstdlib>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_gemm.h"
#include "vdk_gemm_epilogue.h"

#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_PITCH 72   // bf16 elements per LDS row (64 + 8 pad = 144 B)
#define G_CP 132     // fp32 pitch of the epilogue staging tile

__device__ __forceinline__ void g_load_tile(u32x4 (&r)[4], const bf16_t* __restrict__ base, long ld, int row0,
                                            int nrows, int k0, int kend) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int id = threadIdx.x + 256 * j;
    int row = id >> 3, ch = id & 7;
    int grow = row0 + row, k = k0 + ch * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (grow < nrows && k < kend) v = *(const u32x4*)(base + (long)grow * ld + k);
    r[j] = v;
  }
}
__device__ __forceinline__ void g_store_tile(const u32x4 (&r)[4], bf16_t* lds) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int id = threadIdx.x + 256 * j;
    int row = id >> 3, ch = id & 7;
    *(u32x4*)(lds + row * G_PITCH + ch * 8) = r[j];
  }
}

// ---- implicit-GEMM convolution: the A tile is gathered from an NHWC tensor, no im2col buffer --------------------------------------
// GEMM row m = (b, oy, ox) of the cOH x cOW row grid, GEMM column k = (ky, kx, c) with c fastest (Cin % 8 == 0, so a 16-byte chunk stays inside one
// tap).  Forward gather: source pixel (oy*stride + ky - pad, ox*stride + kx - pad) of the cH x cW tensor.  Transposed gather (input gradient of the
// same convolution, rows = input pixels, A = dY): source pixel ((oy + pad - ky) / stride, (ox + pad - kx) / stride) when both divide.  Out of range -> 0.
struct ConvRows { int b[4], oy[4], ox[4]; };
__device__ __forceinline__ void g_conv_rows(ConvRows& cr, const GemmParams& p, int row0) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int id = threadIdx.x + 256 * j, m = row0 + (id >> 3);
    const int hw = p.cOH * p.cOW, b = m / hw, rem = m - b * hw, oy = rem / p.cOW;
    cr.b[j] = m < p.M ? b : -1; cr.oy[j] = oy; cr.ox[j] = rem - oy * p.cOW;
  }
}
__device__ __forceinline__ void g_load_tile_conv(u32x4 (&r)[4], const GemmParams& p, const ConvRows& cr, int k0, int kend) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int id = threadIdx.x + 256 * j, ch = id & 7, k = k0 + ch * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (cr.b[j] >= 0 && k < kend) {
      const int tap = k / p.cCin, c = k - tap * p.cCin, ky = tap / p.cKW, kx = tap - ky * p.cKW;
      int y, x; bool ok;
      if (!p.ctrans) {
        y = cr.oy[j] * p.cstride + ky - p.cpad; x = cr.ox[j] * p.cstride + kx - p.cpad;
        ok = y >= 0 && y < p.cH && x >= 0 && x < p.cW;
      } else {
        const int ty = cr.oy[j] + p.cpad - ky, tx = cr.ox[j] + p.cpad - kx;
        y = ty / p.cstride; x = tx / p.cstride;
        ok = ty >= 0 && tx >= 0 && y * p.cstride == ty && x * p.cstride == tx && y < p.cH && x < p.cW;
      }
      if (ok) v = *(const u32x4*)(p.A + (((long)cr.b[j] * p.cH + y) * p.cW + x) * p.cCin + c);
    }
    r[j] = v;
  }
}

template <bool CONV, int OF = 0>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(GemmParams p) {
  // 2 buffers x (A 128 rows + B 128 rows) x 144 B = 73,728 B; reused as the fp32 epilogue tile (67,584 B)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * 128 * G_PITCH * 2];
  bf16_t* const As = (bf16_t*)smem;                       // [2][128][PITCH]
  bf16_t* const Bs = As + 2 * 128 * G_PITCH;              // [2][128][PITCH]
  float* const Cs = (float*)smem;                         // [128][G_CP]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1, hi = lane >> 5, l31 = lane & 31;

  // XCD-aware bijective swizzle (cdna_hip_programming.md T1): blocks that share an XCD's L2 get a
  // contiguous run of tiles, N-tiles fastest, so they reuse the same A row panel / weight matrix.
  const int ntn = (p.N + G_BN - 1) / G_BN, ntm = (p.M + G_BM - 1) / G_BM;
  const int nwg = ntn * ntm;
  int bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tn = bid % ntn, tm = bid / ntn;
  const int m0 = tm * G_BM, n0 = tn * G_BN;
  const int z = blockIdx.y;
  const int kbeg = z * p.k_per_split;
  int kend = kbeg + p.k_per_split; if (kend > p.K) kend = p.K;
  const int nk = (kend - kbeg + G_BK - 1) / G_BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[4], rb[4];
  ConvRows cr;
  if (CONV) g_conv_rows(cr, p, m0);
  if (nk > 0) {
    if (CONV) g_load_tile_conv(ra, p, cr, kbeg, kend); else g_load_tile(ra, p.A, p.lda, m0, p.M, kbeg, kend);
    g_load_tile(rb, p.B, p.ldb, n0, p.N, kbeg, kend);
    g_store_tile(ra, As);
    g_store_tile(rb, Bs);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      if (CONV) g_load_tile_conv(ra, p, cr, kbeg + (kt + 1) * G_BK, kend); else g_load_tile(ra, p.A, p.lda, m0, p.M, kbeg + (kt + 1) * G_BK, kend);
      g_load_tile(rb, p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * G_BK, kend);
    }
    const bf16_t* a_base = As + cur * 128 * G_PITCH + (wm * 64 + l31) * G_PITCH + hi * 8;
    const bf16_t* b_base = Bs + cur * 128 * G_PITCH + (wn * 64 + l31) * G_PITCH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < G_BK / 16; ++ks) {
      s16x8 a0 = *(const s16x8*)(a_base + ks * 16);
      s16x8 a1 = *(const s16x8*)(a_base + 32 * G_PITCH + ks * 16);
      s16x8 b0 = *(const s16x8*)(b_base + ks * 16);
      s16x8 b1 = *(const s16x8*)(b_base + 32 * G_PITCH + ks * 16);
      acc[0][0] = vdk_mfma32<OF>(a0, b0, acc[0][0]);
      acc[0][1] = vdk_mfma32<OF>(a0, b1, acc[0][1]);
      acc[1][0] = vdk_mfma32<OF>(a1, b0, acc[1][0]);
      acc[1][1] = vdk_mfma32<OF>(a1, b1, acc[1][1]);
    }
    if (kt + 1 < nk) {
      g_store_tile(ra, As + (cur ^ 1) * 128 * G_PITCH);
      g_store_tile(rb, Bs + (cur ^ 1) * 128 * G_PITCH);
    }
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> coalesced 8-wide row chunks ----------------------
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        int col = wn * 64 + j * 32 + l31;
        Cs[row * G_CP + col] = acc[i][j][r];
      }
  __syncthreads();
  const int cc = (tid & 15) * 8;
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 16 + (tid >> 4);
    const long mi = m0 + row;
    const int n = n0 + cc;
    if (mi >= p.M || n >= p.N) continue;
    float v[8];
    {
      f32x4 x0 = *(const f32x4*)(Cs + row * G_CP + cc);
      f32x4 x1 = *(const f32x4*)(Cs + row * G_CP + cc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
    }
    g_epilogue_store8<OF>(p, mi, n, v, z);
  }
}


// =====================================================================================================
// 256x256x64 "8-segment" kernel (cdna_hip_programming.md §5 template, re-derived for 32x32x16 MFMA):
//   * 8 waves = 2 (M) x 4 (N), wave tile 128x64 = 4x2 v_mfma_f32_32x32x16_bf16 accumulators (128 regs)
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  The DMA image is
//     lane-linear, so the bank swizzle lives on the SOURCE address: LDS chunk c' of row r holds global chunk
//     c' ^ ((r >> 1) & 7); reading logical chunk c uses the same involution -> conflict-free ds_read_b128.
//   * a K-tile is 4 LDS regions of 16 KB: RA0/RA1 = the first/second 64 rows of BOTH waves-rows' 128-row halves,
//     RB0/RB1 likewise for the 4 wave-columns.  A K-tile is consumed in 4 phases (A0B0, A0B1, A1B1, A1B0), each a
//     read segment R and an MFMA segment M separated by s_barrier; a region is refilled for tile t+2 (RA1: t+1) a
//     few segments after its last reader, so HBM latency spans >= 5 segments with only 2 tile buffers (128 KB).
//   * counted vmcnt: the single wait per K-tile (segment R4) is vmcnt(4) = "everything but the 4 newest DMAs",
//     never 0 in steady state.
//   * the two wave-rows run staggered by one barrier interval, so on every SIMD one wave's MFMA segment overlaps
//     the other wave's LDS reads / DMA issue.
#define H_REGION 16384
#define H_TILEBUF (4 * H_REGION)
#define H_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))   /* vmcnt(n), n < 16; expcnt/lgkmcnt untouched */
#define H_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// issue the 2 DMA instructions that fill one region.  is_b: B operand (rows = N index).  sub: 0/1.
__device__ __forceinline__ void h_issue_region(unsigned char* region, const bf16_t* __restrict__ base, long ld, int row0, int nrows,
                                               int k0, bool is_b, int sub, int w, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rr = j * 64 + w * 8 + (lane >> 3);          // region row 0..127
    const int cp = lane & 7;                              // LDS chunk position
    const int c = cp ^ ((rr >> 1) & 7);                   // global chunk that lives there
    int trow;
    if (is_b) trow = (rr >> 5) * 64 + sub * 32 + (rr & 31);
    else trow = (rr >> 6) * 128 + sub * 64 + (rr & 63);
    int grow = row0 + trow;
    if (grow > nrows - 1) grow = nrows - 1;               // clamp: rows beyond the matrix are masked at the store
    const bf16_t* src = base + (long)grow * ld + k0 + c * 8;
    unsigned char* dst = region + (j * 512 + w * 64) * 16;  // wave-uniform base; the DMA adds lane * 16
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(src), VDK_LDS_PTR(dst), 16, 0, 0);
  }
}
// fragment (32 rows x one 16-wide k-step) of region row block `rb` (multiple of 32): lane (l31, hi), k-step ks
__device__ __forceinline__ s16x8 h_read_frag(const unsigned char* region, int rb, int ks, int l31, int hi) {
  const int rr = rb + l31;
  const int c = (ks * 2 + hi) ^ ((rr >> 1) & 7);
  return *(const s16x8*)(region + rr * 128 + c * 16);
}

// ---- TN operands: A is [K, M] row-major (element (k, m) at A[k*lda + m]), B is [K, N]: C = A^T . B (wgrad straight from
// dY[t][out] and X[t][in], no transposed copies).  A region holds the same logical 128 operand-rows x 64 k, stored k-major:
// [64 k][128 cols] = 64 x 256 B; 16-B chunk c' of k-row t holds global chunk c' ^ (4 * (t & 3)), which spreads the four
// k-rows of a ds_read_b64_tr_b16 over the four 64-B bank quarters.
__device__ __forceinline__ void h_issue_region_tn(unsigned char* region, const bf16_t* __restrict__ base, long ld, int col0, int ncols,
                                                  int k0, bool is_b, int sub, int w, int lane, int row_group) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int L = j * 512 + w * 64 + lane;                 // 16-B slot index inside the region
    const int trow = L >> 4, cp = L & 15;
    const int c = cp ^ (4 * (trow & 3));                   // logical chunk (8 columns) that lives at position cp
    int gcol;
    if (is_b) gcol = (c >> 2) * 64 + sub * 32 + (c & 3) * 8;
    else gcol = (c >> 3) * 128 + sub * 64 + (c & 7) * 8;
    gcol += col0;
    if (gcol > ncols - 8) gcol = ncols - 8;                // clamp: columns beyond the matrix are masked at the store
    long krow = k0 + trow;
    if (row_group > 0) krow = krow + krow / row_group + 1; // token buffer without its cls rows (patch-embed wgrad)
    const bf16_t* src = base + krow * ld + gcol;
    unsigned char* dst = region + (j * 512 + w * 64) * 16;
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(src), VDK_LDS_PTR(dst), 16, 0, 0);
  }
}
// fragment of 32 operand-rows (region columns cb .. cb+31), k-step ks: two transpose reads of 4 k-rows each
__device__ __forceinline__ s16x8 h_read_frag_tn(const unsigned char* region, int cb, int ks, int lane) {
  const int s = lane & 15, chalf = (lane >> 4) & 1, hi = lane >> 5;
  const int col = cb + 16 * chalf + 4 * (s & 3);
  const int t1 = ks * 16 + hi * 8 + (s >> 2), t2 = t1 + 4;
  const unsigned char* p1 = region + t1 * 256 + (((col >> 3) ^ (4 * (t1 & 3))) * 16) + ((col >> 2) & 1) * 8;
  const unsigned char* p2 = region + t2 * 256 + (((col >> 3) ^ (4 * (t2 & 3))) * 16) + ((col >> 2) & 1) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1));
  s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p2));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return r;
}


// SK (stream-K): a persistent launch of G = gridDim.x workgroups (one per CU).  The work is the list of (tile, k-tile) units in tile-raster order; workgroup at
// position pos = xcd * (G / 8) + slot-in-xcd (so neighbouring ranges share an XCD's L2) owns the contiguous range [pos * U / G, (pos + 1) * U / G): a partial tile
// at its head, whole tiles, a partial tile at its tail.  Whole tiles take the normal epilogue.  A partial segment stores its raw accumulators to its own fp32 slab
// (2 per workgroup: head / tail), publishes them (release fence + device-scope atomic add of its k-tile count on the tile's counter), and whoever completes the
// count re-reads every contributor's slab IN K ORDER (its own included: the sum order is fixed by the partition, not by arrival -> bit-reproducible), resets
// the counter for the next launch and runs the epilogue.  Nobody waits for anybody: no co-residency requirement, no deadlock with RCCL kernels on the same CUs.
template <bool TN, int E, bool SK, bool CS>
__global__ __launch_bounds__(512) void gemm256_bf16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * H_TILEBUF];   // 128 KB: operand buffers, reused by the epilogue
  __shared__ float cs_lds[8][64];                                              // + 2 KB: per-wave column sums of the a_colsum by-product
  __shared__ int sk_last, sk_st[4];
  const int tid0 = threadIdx.x;
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  const int nwg = ntn * ntm;
  const int nkt = p.K / 64;                               // SK: k-tiles per output tile
  const int z = SK ? 0 : blockIdx.y;
  // SK state lives in LDS, not in registers: the main loop already uses every VGPR and nearly every SGPR (loop-carried scalars spilled 34 VGPRs into the K loop)
  //   sk_st[0] = cursor in the tail's unit space, [1] = end of this workgroup's tail range, [2] = whole-tile rounds done, [3] = start of the tail range
  if (SK) {
    if (tid0 == 0) {
      const int G = gridDim.x, pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
      const long U = (long)(nwg - nwg / G * G) * nkt;
      sk_st[0] = sk_st[3] = (int)(pos * U / G); sk_st[1] = (int)((pos + 1) * U / G); sk_st[2] = 0;
    }
  }
  do {
  // SK: the lane / wave ids go through an opaque asm every segment, so nothing derived from them (DMA source offsets, fragment addresses, epilogue addresses) is
  // hoisted out of the segment loop and kept alive -- i.e. spilled -- across the K loop of every segment
  int tid = tid0;
#ifndef VDK_EMU
  if (SK) asm volatile("" : "+v"(tid));
#endif
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3, hi = lane >> 5, l31 = lane & 31;
  int tile, kbeg, nk;
  if (SK) {
    // hybrid: floor(tiles / G) rounds of whole tiles (an XCD's G/8 workgroups on G/8 consecutive tiles, like the one-tile-per-workgroup launch: they walk the
    // same A / B panels in step, which is what keeps those panels in that XCD's L2), then the remaining tiles' k-tiles dealt evenly
    __syncthreads();                                      // state visible; the previous segment's epilogue / slab traffic is done with the LDS
    const int G = gridDim.x, R = nwg / G;
    const int dp = __builtin_amdgcn_readfirstlane(sk_st[2]), u = __builtin_amdgcn_readfirstlane(sk_st[0]), ue = __builtin_amdgcn_readfirstlane(sk_st[1]);
    if (dp < R) {
      tile = (blockIdx.x & 7) * (R * (G >> 3)) + dp * (G >> 3) + (blockIdx.x >> 3);
      kbeg = 0; nk = nkt;
    } else {
      if (u >= ue) break;
      tile = u / nkt;
      const int kt0 = u - tile * nkt;
      nk = nkt - kt0; if (nk > ue - u) nk = ue - u;
      kbeg = kt0 * 64;
      tile += R * G;
    }
  } else {
    int bid = blockIdx.x;
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    kbeg = z * p.k_per_split;
    int kend = kbeg + p.k_per_split; if (kend > p.K) kend = p.K;
    nk = (kend - kbeg) / 64;                              // launcher guarantees (kend - kbeg) % 64 == 0
  }
  const int tn = tile % ntn, tm = tile / ntn;
  const int m0 = tm * 256, n0 = tn * 256;
  const int kt_abs0 = kbeg / 64;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define H_REG(buf, id) (smem + (buf) * H_TILEBUF + (id) * H_REGION)   /* id: 0 RA0, 1 RA1, 2 RB0, 3 RB1 */
#define H_ISSUE_A(buf, sub, t)                                                                                              \
  do {                                                                                                                     \
    if (TN) h_issue_region_tn(H_REG(buf, sub), p.A, p.lda, m0, p.M, kbeg + (t) * 64, false, sub, w, lane, p.a_row_group);   \
    else h_issue_region(H_REG(buf, sub), p.A, p.lda, m0, p.M, kbeg + (t) * 64, false, sub, w, lane);                        \
  } while (0)
#define H_ISSUE_B(buf, sub, t)                                                                                              \
  do {                                                                                                                     \
    if (TN) h_issue_region_tn(H_REG(buf, 2 + (sub)), p.B, p.ldb, n0, p.N, kbeg + (t) * 64, true, sub, w, lane, 0);          \
    else h_issue_region(H_REG(buf, 2 + (sub)), p.B, p.ldb, n0, p.N, kbeg + (t) * 64, true, sub, w, lane);                   \
  } while (0)
#define H_FRAG_A(reg, rt, ks) (TN ? h_read_frag_tn(reg, wr * 64 + (rt) * 32, ks, lane) : h_read_frag(reg, wr * 64 + (rt) * 32, ks, l31, hi))
#define H_FRAG_B(reg, ks) (TN ? h_read_frag_tn(reg, wc * 32, ks, lane) : h_read_frag(reg, wc * 32, ks, l31, hi))

  unsigned long long t_start = 0, t_landed = 0, t_main = 0;
  if (!SK && p.dbg) t_start = __builtin_readcyclecounter();
  // ---- prologue: tiles 0 and 1 completely ---------------------------------------------------------------------------
  if (nk > 0) { H_ISSUE_A(0, 0, 0); H_ISSUE_B(0, 0, 0); H_ISSUE_B(0, 1, 0); H_ISSUE_A(0, 1, 0); }
  if (nk > 1) { H_ISSUE_A(1, 0, 1); H_ISSUE_B(1, 0, 1); H_ISSUE_B(1, 1, 1); H_ISSUE_A(1, 1, 1); H_WAIT_VM(8); } else { H_WAIT_VM(0); }
  H_BAR();
  if (!SK && p.dbg) t_landed = __builtin_readcyclecounter();
  if (wr == 1) H_BAR();                                   // stagger: the second wave-row runs one interval behind

  // K-tile schedule: 4 barrier intervals per tile; wave-row 1 runs one interval behind wave-row 0, so on every SIMD one wave's
  // 16-MFMA segment overlaps the other wave's LDS reads.  With group 0 at intervals 4t .. 4t+3:
  //   RA(t): read A0 (8), B0 (4), B1 (4) of tile t;  wait vmcnt(6): RA1 of tile t has landed (read two barriers later, in RB)
  //   MA(t): 16 MFMA A0 x (B0, B1) with the DMA of RA1 of tile t+1 (2) issued after the 6th
  //   RB(t): read A1 (8);                             wait vmcnt(2): RA0, RB0, RB1 of tile t+1 have landed (read in RA(t+1))
  //   MB(t): 16 MFMA A1 x (B1, B0) with the DMAs of RA0, RB0, RB1 of tile t+2 (6) issued in pairs after the 2nd, 6th and 10th
  // WAR: every refill is ISSUED >= 2 intervals after the lagging row issued its last read of that region (RA1: read in RB(t-1),
  // i.e. interval 4t-1 for row 1, refilled at 4t+1; RA0/RB0/RB1: read at 4t+1, refilled at 4t+3), so those reads have retired
  // behind an lgkmcnt(0) + barrier.  RAW: each wait is followed by two barriers before the first read of the data it covers.
  // vmcnt is never 0 in steady state: 2-8 DMAs stay in flight across every barrier.
  s16x8 a0[2][4], a1[2][4], b0[4], b1[4];
  // optional by-product (NT only): column sums of the A operand = bias gradient of the Linear whose dY this dgrad GEMM reads.  Only the tn == 0
  // workgroup of a row tile does it; wave w adds region rows 16w .. 16w+15 of RA0 (in RA) and RA1 (in RB), lane = k column, straight from the staged tile.
  // Lane = (k pair kp, row parity rh): 8 ds_read_b32 per region; the 8 waves' sums meet in cs_lds one K-tile later (wave-row 1 lags one interval).
  // The K-tiles of a row tile are dealt round-robin to its ntn workgroups (tile t belongs to tn == t % ntn): every workgroup pays 1/ntn of the extra
  // LDS reads instead of one workgroup per row tile paying all of them and holding up its whole round.
  const bool cs_en = CS;                                 // compile-time: the by-product costs ~8 VGPRs the other variants need (256 are in use)
  float cs0 = 0.f, cs1 = 0.f;
  const int cs_kp = lane & 31, cs_rh = lane >> 5;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const unsigned char* RA0 = H_REG(cur, 0); const unsigned char* RA1 = H_REG(cur, 1);
    const unsigned char* RB0 = H_REG(cur, 2); const unsigned char* RB1 = H_REG(cur, 3);
    // ---- RA --------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b0[ks] = H_FRAG_B(RB0, ks);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a0[rt][ks] = H_FRAG_A(RA0, rt, ks);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b1[ks] = H_FRAG_B(RB1, ks);
    const bool cs_on = cs_en && ((kt_abs0 + t) % ntn) == tn;
    if (cs_en) {
      if (t > 0 && w == 0 && ((kt_abs0 + t - 1) % ntn) == tn) {   // K-tile t-1: all eight waves' sums are in cs_lds (the lagging wave-row wrote them one barrier ago)
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v += cs_lds[i][lane];
        p.colsum_part[(long)tm * p.K + kbeg + (t - 1) * 64 + lane] = v;
      }
    }
    if (cs_on) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = w * 16 + 2 * i + cs_rh;
        const unsigned u = *(const unsigned*)(RA0 + rr * 128 + (((cs_kp >> 2) ^ ((rr >> 1) & 7)) * 16) + (cs_kp & 3) * 4);
        cs0 += bf_lo(u); cs1 += bf_hi(u);
      }
    }
    if (t + 1 < nk) { H_WAIT_VM(6); } else { H_WAIT_VM(0); }
    H_BAR();
    // ---- MA --------------------------------------------------------------------------------------------------------
    // The DMA instructions are issued BETWEEN the MFMAs of the segment, not in front of them: an LDS-DMA costs 60-180 issue cycles, and a batch of two
    // (MA) or six (MB) ahead of the first MFMA left the matrix pipe idle that long in every interval (+10 % at 4096^3 when interleaved).
    VDK_PIN2(acc[0][0], acc[1][0]); VDK_PIN2(acc[0][1], acc[1][1]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0][ks], b0[ks], acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1][ks], b0[ks], acc[1][0], 0, 0, 0);
      if (ks == 1 && t >= 1 && t + 1 < nk) {                  // (tile 1's RA1 was issued by the prologue)
        __builtin_amdgcn_sched_barrier(0); H_ISSUE_A(cur ^ 1, 1, t + 1); __builtin_amdgcn_sched_barrier(0);
      }
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0][ks], b1[ks], acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1][ks], b1[ks], acc[1][1], 0, 0, 0);
    }
    VDK_PIN2(acc[0][0], acc[1][0]); VDK_PIN2(acc[0][1], acc[1][1]);
    __builtin_amdgcn_s_setprio(0);
    H_BAR();
    // ---- RB --------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a1[rt][ks] = H_FRAG_A(RA1, rt, ks);
    if (cs_on) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = w * 16 + 2 * i + cs_rh;
        const unsigned u = *(const unsigned*)(RA1 + rr * 128 + (((cs_kp >> 2) ^ ((rr >> 1) & 7)) * 16) + (cs_kp & 3) * 4);
        cs0 += bf_lo(u); cs1 += bf_hi(u);
      }
      cs0 += __shfl_xor(cs0, 32); cs1 += __shfl_xor(cs1, 32);
      if (cs_rh == 0) { cs_lds[w][2 * cs_kp] = cs0; cs_lds[w][2 * cs_kp + 1] = cs1; }
      cs0 = 0.f; cs1 = 0.f;
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the sums are in LDS before this wave passes the barrier (wave 0 reads them one interval later)
    }
    if (t + 1 < nk) { H_WAIT_VM(2); } else { H_WAIT_VM(0); }
    H_BAR();
    // ---- MB --------------------------------------------------------------------------------------------------------
    VDK_PIN2(acc[2][1], acc[3][1]); VDK_PIN2(acc[2][0], acc[3][0]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0][ks], b1[ks], acc[2][1], 0, 0, 0);
      acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1][ks], b1[ks], acc[3][1], 0, 0, 0);
      if (t + 2 < nk) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) H_ISSUE_A(cur, 0, t + 2);
        if (ks == 1) H_ISSUE_B(cur, 0, t + 2);
        if (ks == 2) H_ISSUE_B(cur, 1, t + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0][ks], b0[ks], acc[2][0], 0, 0, 0);
      acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1][ks], b0[ks], acc[3][0], 0, 0, 0);
    }
    VDK_PIN2(acc[2][1], acc[3][1]); VDK_PIN2(acc[2][0], acc[3][0]);
    __builtin_amdgcn_s_setprio(0);
    H_BAR();
  }
  if (wr == 0) H_BAR();                                   // match the barrier count of the lagging wave-row
  if (cs_en && nk > 0 && w == 0 && ((kt_abs0 + nk - 1) % ntn) == tn) {   // last K-tile's column sums (every wave is past its final RB)
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v += cs_lds[i][lane];
    p.colsum_part[(long)tm * p.K + kbeg + (nk - 1) * 64 + lane] = v;
  }
  if (!SK && p.dbg) t_main = __builtin_readcyclecounter();

  if (SK && nk < nkt) {
    // ---- partial tile: publish the raw accumulators; the workgroup that completes the tile's k-tile count finishes it ----------------------------------
    // Slab traffic uses device-scope relaxed accesses (sc1 write-through stores, L2-bypassing loads) ordered by s_waitcnt + barrier around one device-scope
    // atomic per segment: no L2 write-back / invalidate.  With two contributors (every tile when there are at least as many tiles as workgroups) the sum of
    // the two parts is commutative, so the later one adds the earlier one's slab to its registers and, when it can see beforehand that it is the later one,
    // does not write its own.  Three or more: everybody writes, the last arriver re-reads all of them in K order (own included) -> same bits on every launch.
    const long WGF = 65536;                                // floats per slab (256 x 256)
    const int sk_G = gridDim.x, sk_pos = (blockIdx.x & 7) * (sk_G >> 3) + (blockIdx.x >> 3);
    const long sk_U = (long)(nwg - nwg / sk_G * sk_G) * nkt;
    const bool sk_first = __builtin_amdgcn_readfirstlane(sk_st[0]) == __builtin_amdgcn_readfirstlane(sk_st[3]);
    const long tb = (long)(tile - nwg / sk_G * sk_G) * nkt, te = tb + nkt;
    int p_lo = (int)(tb * sk_G / sk_U);
    while ((long)(p_lo + 1) * sk_U / sk_G <= tb) ++p_lo;
    while ((long)p_lo * sk_U / sk_G > tb) --p_lo;
    int ncontrib = 0;
    for (int pp = p_lo; (long)pp * sk_U / sk_G < te; ++pp)
      if ((long)(pp + 1) * sk_U / sk_G > (long)pp * sk_U / sk_G) ++ncontrib;
    if (tid == 0) sk_last = (ncontrib == 2 && VDK_AGENT_LD_I32(p.sk_cnt + tile) + nk == nkt) ? 2 : 0;
    __syncthreads();
    int last = sk_last;
    if (last == 0) {
      float* my = p.slabs + ((long)sk_pos * 2 + (sk_first ? 0 : 1)) * WGF + w * 8192 + lane * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const unsigned long long v = ((unsigned long long)__float_as_uint(acc[i][j][2 * g + 1]) << 32) | __float_as_uint(acc[i][j][2 * g]);
            VDK_AGENT_ST_U64(my + ((i * 2 + j) * 8 + g) * 128, v);
          }
      __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's write-through stores have completed at device scope
      __syncthreads();
      if (tid == 0) sk_last = (VDK_AGENT_ADD_I32(p.sk_cnt + tile, nk) + nk == nkt) ? 1 : 0;
      __syncthreads();
      last = sk_last;
    }
    if (tid == 0) sk_st[0] += nk;                          // (read again behind the barrier at the top of the next segment)
    if (last == 0) continue;
    if (tid == 0) VDK_AGENT_ST_I32(p.sk_cnt + tile, 0);     // self-cleaning: the next launch finds zeros
    if (ncontrib != 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    for (int pp = p_lo;; ++pp) {
      const long cs_ = (long)pp * sk_U / sk_G, ce_ = (long)(pp + 1) * sk_U / sk_G;
      if (cs_ >= te) break;
      if (ce_ == cs_) continue;
      if (ncontrib == 2 && pp == sk_pos) continue;          // own part is in the registers
      const float* src = p.slabs + ((long)pp * 2 + (cs_ >= tb ? 0 : 1)) * WGF + w * 8192 + lane * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const unsigned long long v = VDK_AGENT_LD_U64(src + ((i * 2 + j) * 8 + g) * 128);
            acc[i][j][2 * g] += __uint_as_float((unsigned)v); acc[i][j][2 * g + 1] += __uint_as_float((unsigned)(v >> 32));
          }
    }
  } else if (SK) {
    if (tid == 0) {
      if (sk_st[2] < nwg / (int)gridDim.x) sk_st[2] += 1;
      else sk_st[0] += nk;
    }
  }

  // ---- epilogue: wave-private 16 KB LDS slab, 64 rows x 64 fp32 at a time -> 8-wide coalesced row chunks ---------
  float* slab = (float*)(smem + w * 16384);
  const int ncol = n0 + wc * 64 + (lane & 7) * 8;
  float bias8[8], ocs8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; ocs8[e] = 0.f; }
  if ((E != E_GENERIC) && (E & E_BIAS) && ncol < p.N) {
    f32x4 b0v = *(const f32x4*)(p.bias + ncol), b1v = *(const f32x4*)(p.bias + ncol + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias8[e] = b0v[e]; bias8[4 + e] = b1v[e]; }
  }
  float q8_unused = 0.f;                                  // (the fp8 by-product E_Q8 is instantiated by the fp8 kernel only)
#pragma unroll
  for (int half = 0; half < 2; ++half) {   // fully unrolled: acc must be indexed statically (else it lives in scratch)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + ct * 32 + l31] = acc[half * 2 + rt][ct][r];
    __builtin_amdgcn_wave_barrier();
    if (E != E_GENERIC) {
      h_epilogue_half<E>(p, slab, lane, (long)m0 + wr * 128 + half * 64, ncol, z, bias8, ocs8, q8_unused);
    } else {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 8 + (lane >> 3), cc = (lane & 7) * 8;
        const long mi = (long)m0 + wr * 128 + half * 64 + row;
        const int n = n0 + wc * 64 + cc;
        if (mi < p.M && n < p.N) {
          float v[8];
          f32x4 x0 = *(const f32x4*)(slab + row * 64 + cc), x1 = *(const f32x4*)(slab + row * 64 + cc + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
          g_epilogue_store8(p, mi, n, v, z);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if ((E != E_GENERIC) && (E & E_OCS)) {   // lanes with the same (lane & 7) hold the same 8 columns: sum over the 8 row slots, one partial row per (row tile, wave row)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = ocs8[e];
      v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      ocs8[e] = v;
    }
    if (lane < 8 && ncol < p.N) {
      float* dst = p.ocs_part + ((long)tm * 2 + wr) * p.N + ncol;
      *(f32x4*)dst = (f32x4){ocs8[0], ocs8[1], ocs8[2], ocs8[3]};
      *(f32x4*)(dst + 4) = (f32x4){ocs8[4], ocs8[5], ocs8[6], ocs8[7]};
    }
  }
  if (!SK && p.dbg && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the stamp is taken when this wave's stores have left
    unsigned long long* o = p.dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    o[0] = t_start; o[1] = t_landed; o[2] = t_main; o[3] = __builtin_readcyclecounter();
  }
  } while (SK);
#undef H_REG
#undef H_ISSUE_A
#undef H_ISSUE_B
#undef H_FRAG_A
#undef H_FRAG_B
}

// out[i] = alpha * sum_s slabs[s][i]  (deterministic split-K combine), optional bf16 output
template <int OF = 0>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int S, long n4 /*=MN/4*/,
                                                            long mn, float alpha, float* __restrict__ outf,
                                                            bf16_t* __restrict__ outb) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = *(const f32x4*)(slabs + i * 4);
  for (int k = 1; k < S; ++k) {
    f32x4 t = *(const f32x4*)(slabs + (long)k * mn + i * 4);
    s += t;
  }
  s *= alpha;
  if (outf) *(f32x4*)(outf + i * 4) = s;
  if (outb) *(u32x2*)(outb + i * 4) = (u32x2){pack_op2<OF>(s[0], s[1]), pack_op2<OF>(s[2], s[3])};
}

// out[c][r] = in[r][c] for r < R, 0 for R <= r < Rpad (bf16).  64x64 tiles through LDS.
template <int OF = 0>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, long ldi, int R, int Cc,
                                                             bf16_t* __restrict__ out, long ldo, int Rpad, int row_group,
                                                             float* __restrict__ colsum_partial) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  // load: 64 rows x 64 cols, each thread 2 elements (one 32-bit word) x 8 passes
  for (int i = 0; i < 8; ++i) {
    int row = i * 8 + (tid >> 5), cw = (tid & 31) * 2;
    int gr = r0 + row, gc = c0 + cw;
    bf16_t v0 = 0, v1 = 0;
    if (gr < R) {
      const long sr = row_group > 0 ? (long)gr + gr / row_group + 1 : (long)gr;  // skip one cls row per group
      if (gc + 1 < Cc) { unsigned wv = *(const unsigned*)(in + sr * ldi + gc); v0 = (bf16_t)(wv & 0xffff); v1 = (bf16_t)(wv >> 16); }
      else if (gc < Cc) v0 = in[sr * ldi + gc];
    }
    tile[row][cw] = v0; tile[row][cw + 1] = v1;
  }
  __syncthreads();
  if (colsum_partial && tid < 64 && c0 + tid < Cc) {  // bias gradient rides along: per-row-tile column sums
    float sacc = 0.f;
    for (int r = 0; r < 64; ++r) sacc += op2f<OF>(tile[r][tid]);
    colsum_partial[(long)blockIdx.x * Cc + c0 + tid] = sacc;
  }
  for (int i = 0; i < 8; ++i) {
    int col = i * 8 + (tid >> 5), rw = (tid & 31) * 2;   // output row = input col
    int gc = c0 + col, gr = r0 + rw;
    if (gc < Cc && gr < Rpad) {  // Rpad is even
      unsigned wv = (unsigned)tile[rw][col] | ((unsigned)tile[rw + 1][col] << 16);
      *(unsigned*)(out + (long)gc * ldo + gr) = wv;
    }
  }
}

// ---- optional live profiling of the GEMM launches (bench.py's `roofline` object) --------------------------------
// vdk_prof_begin(n) pre-creates n event pairs; while enabled every vdk_gemm_bf16_nt call attaches a (start, stop) pair to its GEMM dispatch itself
// (hipExtLaunchKernelGGL: the timestamps of the dispatch packet's own completion signal -- no extra packets in the queue; bracketing with hipEventRecord cost
// 1.0 ms of the 43 ms step, two marker packets around each of its 149 GEMMs); vdk_prof_end() synchronises and returns total GEMM time, launches and flops.
#include <vector>
// Tuning / test knobs and the profiling hooks are PER CALLING THREAD (no process-global mutable state on the compute path): a host thread that forces a kernel
// structure or times its launches does not change what another thread's calls do.  Consequence for PyTorch hosts: autograd runs Function.backward on its own worker
// threads, so a knob armed on the Python main thread (vdk_gemm_force_kernel, vdk_prof_begin) reaches the launches of the native engines' C calls made from that thread --
// the fused steps and bench.py -- but NOT GEMMs launched from the backward of the autograd-node forms (_NeckFn, _SwinFunction, SwinTransformerAutograd).
static thread_local std::vector<hipEvent_t> g_prof_ev;
static thread_local std::vector<double> g_prof_flops;
static thread_local double g_prof_bytes = 0.0;   // algorithmic bytes of the profiled launches: every operand / output / epilogue tensor counted once
static thread_local size_t g_prof_used = 0;
static thread_local bool g_prof_on = false;
static thread_local void* g_dbg_ptr = nullptr;
static thread_local int g_force_kernel = 0;   // 0 auto, 1 = 128x128 register-staged, 2 = 256x256 LDS-DMA, 3 = 256x256 stream-K whenever a workspace is passed, 4 = never stream-K (tests / A-B benchmarking)
static thread_local int g_sk_grid = 0;        // stream-K workgroups; 0 = one per CU of the current device
static thread_local int g_last_kernel = 0;
// which 256x256 structure serves the big problems: the 4-wave kernel of gemm_w4.hip (default) or the 8-wave kernel above (VDK_GEMM_W4=0, vdk_gemm_force_kernel(2 / 3);
// vdk_gemm_force_kernel(5) = the 4-wave kernel whenever it can serve)
static bool w4_enabled_env() {
  static const bool on = !(getenv("VDK_GEMM_W4") && atoi(getenv("VDK_GEMM_W4")) == 0);
  return on;
}
// ... and which problems go to the 256x128 / two-workgroups-per-CU form instead (gemm_w4h_kernel): the ones whose epilogue is long -- a second resident workgroup
// multiplies meanwhile -- and short problems with few tiles per CU.  VDK_GEMM_W4H (optional): explicit bit mask over {1: GELU, 2: dGELU, 4: residual, 8: other NT, 16: TN}.
static thread_local int t_tn_half = 0;
void vdk_gemm_tn_prefer_half(int on) { t_tn_half = on; }
static bool w4h_wanted(int E, bool trans, long M, long N, long K) {
  static const int mask = getenv("VDK_GEMM_W4H") ? atoi(getenv("VDK_GEMM_W4H")) : -1;
  if (g_force_kernel == 6) return true;
  if (g_force_kernel == 2 || g_force_kernel == 3 || g_force_kernel == 5 || !w4_enabled_env()) return false;
  if (mask >= 0) {   // explicit choice per form
    if (trans) return (mask & 16) != 0;
    if (E == E_GENERIC) return false;
    if (E & E_GELU) return (mask & 1) != 0;
    if (E & E_DGELU) return (mask & 2) != 0;
    if (E & E_RES) return (mask & 4) != 0;
    return (mask & 8) != 0;
  }
  // default, from tools/bench_gemm_w4.py at the ViT-B/16 shapes (T = 50 432; us, persistent four-wave / 256x128 two-workgroup form):
  //   dGELU + column sums 393 / 335 -> 256x128;  GELU + saved pre-activation 327 / 336 -> four-wave;  fp32 residual: K 768 111 / 105, K 3072 239 / 255 -> by K;
  //   light epilogues: N 768 K 768 63 / 56 -> 256x128 when the k-range is short and every CU gets at most ~3 tiles, else four-wave (158-205 / 162-221);  TN -> four-wave
  if (trans) return t_tn_half != 0;
  if (E == E_GENERIC) return false;
  const long tiles = ((M + 255) / 256) * ((N + 255) / 256);
  // round 4, tools/bench_gemm_swin.py (swin_base at batch 128; us, four-wave / 256x128): fewer 256x256 tiles than HALF the CUs -> the half tiles are twice the parallelism
  // (6 272 rows, stage 4: N 1024 K 4096 69 / 51, N 1024 K 1024 29 / 22, N 3072 -> 1024 52 / 39), whatever the epilogue.  (196 tiles -- 25 088 x 512 -- still prefer the
  // four-wave kernel when K is long: K 2048 49 / 53, in the step 50 / 66.)
  if (tiles < 128) return true;
  // round 6 (tools/r6_stagger_ab.py, persistent walk with its start phase): the one-multiplication dGELU form (saved derivative) 275-281 us four-wave against 285.5; the
  // form that evaluates GELU' stays with the two-workgroup kernel (323 against 340-355)
  if ((E & E_DGELU) && (E & E_AUXD) && tiles > 256) return false;
  if (E & E_DGELU) return true;
  // GELU: the four-wave kernel's whole-tile rounds against hardware-dispatched half tiles -- 25 088 x 2048 x 512 (stage 3) is 784 tiles = 3.06 rounds, 99 / 90 us
  // (round 6, after the epilogue stores became non-temporal: fc1 + GELU of ViT-B/16 on the two-workgroup form, whose second workgroup multiplies under the activation
  //  arithmetic, is 0.45 ms per step ahead in same-box A/B -- 34.78 / 34.83 against 35.19 / 35.40 ms; cfg3 -0.3 ms, swin_base -0.05: K <= 768 is the default now)
  static const long gelu_k = getenv("VDK_GEMM_W4H_GELU_K") ? atol(getenv("VDK_GEMM_W4H_GELU_K")) : 768;      // GELU problems with K <= this go to the two-workgroup form
  if ((E & E_GELU) && gelu_k > 0 && K <= gelu_k) return true;
  if (E & E_GELU) return K <= 768 && (double)tiles / (double)((tiles + 255) / 256 * 256) < 0.8;
  if (E & E_RES) return K < 1536 && tiles <= 256;      // (round 6: more tiles than CUs -> the persistent walk with its start phase: proj + fp32 residual 102-105 us against 107 us)
  return K <= 1024 && tiles <= 3 * 256;
}
// rows from which a 128 <= N < 256 problem goes to the 256x128 kernel (one workgroup per 256 rows: below ~one workgroup per CU the 128x128 kernel has more parallelism);
// VDK_GEMM_NARROW_MIN_M overrides (tests)
static long narrow_min_m() {
  static const long v = getenv("VDK_GEMM_NARROW_MIN_M") ? atol(getenv("VDK_GEMM_NARROW_MIN_M")) : 65536L;
  return v;
}
static bool w4_enabled() {
  return (w4_enabled_env() || g_force_kernel == 5) && g_force_kernel != 2 && g_force_kernel != 3 && g_force_kernel != 6;
}
#define SK_CNT_BYTES 65536       // 16384 tile counters in front of the slabs
static int sk_grid() {
  if (g_sk_grid > 0) return g_sk_grid;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    cus = n & ~7;
  }
  return cus;
}

#define RC_MARGIN(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
extern "C" {

/* rows of the a_colsum by-product ([rows][K] f32) if the NT problem (M, N, K) is served by the 256x256 kernel and M % 256 == 0, else 0 */
int vdk_gemm_a_colsum_rows(int32_t M, int32_t N, int32_t K) {
  const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  const bool big = g_force_kernel == 2 || g_force_kernel == 3 || g_force_kernel == 5 || g_force_kernel == 6 || ((g_force_kernel == 0 || g_force_kernel == 4) && (K % 64 == 0) && tiles256 >= 128 && M >= 256 && N >= 256);
  return (big && (K % 64 == 0) && (M % 256 == 0)) ? (M / 256) : 0;
}

/* rows of the c_colsum by-product ([rows][N] f32: column sums of the stored bf16 output per (row tile, wave row)) if the 256x256 NT kernel serves (M, N, K), else 0 */
int vdk_gemm_c_colsum_rows(int32_t M, int32_t N, int32_t K) {
  const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  const long tiles_h = w4_enabled_env() ? (long)((M + 255) / 256) * ((N + 127) / 128) : 0;      // (the dispatcher's second admission rule: half tiles of the 256x128 kernel)
  const bool big = g_force_kernel == 2 || g_force_kernel == 3 || g_force_kernel == 5 || g_force_kernel == 6 || ((g_force_kernel == 0 || g_force_kernel == 4) && (K % 64 == 0) && (tiles256 >= 128 || tiles_h >= 128) && M >= 256 && N >= 256);
  return (big && (K % 64 == 0)) ? 2 * ((M + 255) / 256) : 0;
}

int vdk_gemm_force_kernel(int32_t which) { g_force_kernel = which; return VDK_OK; }
int vdk_gemm_last_kernel(void) { return g_last_kernel; }

/* stream-K (VdkGemmDesc.splitk == -1): `ws` of vdk_gemm_bf16_nt is then a PERSISTENT workspace of this many bytes whose first 64 KB (the tile counters) the caller
   zeroed once; every launch leaves them zero again.  One workspace serves one stream at a time. */
int vdk_gemm_streamk_workspace_bytes(size_t* bytes) {
  if (!bytes) return vdk_fail(VDK_EINVAL, "vdk_gemm_streamk_workspace_bytes: null");
  *bytes = (size_t)SK_CNT_BYTES + (size_t)sk_grid() * 2 * 65536 * 4;
  return VDK_OK;
}
/* tests / tuning: number of persistent workgroups (multiple of 8; 0 = one per CU) */
int vdk_gemm_streamk_grid(int32_t g) {
  if (g < 0 || (g & 7)) return vdk_fail(VDK_EINVAL, "vdk_gemm_streamk_grid: must be a multiple of 8");
  g_sk_grid = g;
  return VDK_OK;
}
int vdk_gemm_debug_stamps(void* device_u64_buffer) { g_dbg_ptr = device_u64_buffer; return VDK_OK; }

int vdk_prof_begin(int32_t max_launches) {
  if (max_launches < 0) return vdk_fail(VDK_EINVAL, "vdk_prof_begin: bad argument");
  while (g_prof_ev.size() < (size_t)max_launches * 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_prof_begin: hipEventCreate failed");
    g_prof_ev.push_back(e);
  }
  g_prof_flops.clear();
  g_prof_bytes = 0.0;
  g_prof_used = 0;
  g_prof_on = true;
  return VDK_OK;
}
/* suspend / resume the event recording between vdk_prof_begin and vdk_prof_end (bench.py samples every 4th step: a timed dispatch costs ~5 us of queue time) */
int vdk_prof_pause(int32_t paused) { g_prof_on = !paused && !g_prof_ev.empty(); return VDK_OK; }
int vdk_prof_end(double* total_ms, int64_t* launches, double* total_flops) {
  g_prof_on = false;
  double ms = 0.0, fl = 0.0;
  for (size_t i = 0; i + 1 < g_prof_used; i += 2) {
    if (hipEventSynchronize(g_prof_ev[i + 1]) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_prof_end: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof_ev[i], g_prof_ev[i + 1]) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_prof_end: elapsed failed");
    ms += t; fl += g_prof_flops[i / 2];
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int64_t)(g_prof_used / 2);
  if (total_flops) *total_flops = fl;
  return VDK_OK;
}

/* algorithmic bytes (each operand / output / epilogue tensor once) of the launches covered by the last vdk_prof_begin .. vdk_prof_end */
int vdk_prof_bytes(double* total_bytes) {
  if (!total_bytes) return vdk_fail(VDK_EINVAL, "vdk_prof_bytes: null");
  *total_bytes = g_prof_bytes;
  return VDK_OK;
}

int vdk_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t splitk, size_t* bytes) {
  if (!bytes || M <= 0 || N <= 0 || splitk < 1) return vdk_fail(VDK_EINVAL, "vdk_gemm_splitk_workspace_bytes: bad argument");
  *bytes = splitk > 1 ? (size_t)splitk * M * N * 4 : 0;
  return VDK_OK;
}

// C = epilogue(alpha * A[M,K] * B[N,K]^T).  A, B bf16 row-major with leading dims lda/ldb (elements).
// Requirements: K % 8 == 0, N % 8 == 0, lda/ldb/ldc/ldaux % 8 == 0 (16-B rows), pointers 16-B aligned.
// splitk > 1: fp32 partial slabs in `ws` then a deterministic combine; only bias-free, act-free,
// residual-free outputs are allowed in that mode (it is the wgrad path).
int vdk_gemm_bf16_nt(const GemmDesc* d, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !d->A || !d->B || !d->C) return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: null pointer");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: empty problem");
  if ((d->K & 7) || (d->N & 7) || (!d->conv && (d->lda & 7)) || (d->ldb & 7) || (d->ldc & 7))
    return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: K, N, lda, ldb, ldc must be multiples of 8");
  // fp16 operands run on the four-wave kernels only, whose rows leave (and arrive) through 32-bit buffer offsets: an output, residual, aux or NT-A tensor of 2 GB or more
  // (the margin head at C = 10^6: cos and dW^ are [512, 10^6] fp32) is computed in ROW BLOCKS that fit -- the same kernels on sub-problems, every output element from the same
  // k-order, so the result does not depend on the split.
  if (d->ab_dtype == VDK_F16 && !d->conv && d->splitk <= 1 && d->splitk != -1 && d->row_group == 0 && !d->a_colsum && !d->c_colsum && d->a_row_group == 0) {
    double lim = 2147483648.0 - 65536.0;
    if (const char* e = getenv("VDK_GEMM_ROWBLOCK_LIMIT")) { const double v = atof(e); if (v > 0) lim = v; }      // tests: exercise the row-block path on small problems
    double rowb = (double)d->ldc * (d->c_dtype == VDK_F32 ? 4.0 : 2.0);
    if (d->aux && (double)d->ldaux * 2.0 > rowb) rowb = (double)d->ldaux * 2.0;
    if (d->residual && (double)d->ldr * 4.0 > rowb) rowb = (double)d->ldr * 4.0;
    if (!d->trans && (double)d->lda * 2.0 > rowb) rowb = (double)d->lda * 2.0;
    if (((double)d->M + 256.0) * rowb >= lim) {
      long mb = (long)(lim / rowb) - 256;
      mb = mb / 256 * 256;
      if (mb >= 256) {
        for (long m0 = 0; m0 < d->M; m0 += mb) {
          GemmDesc sub = *d;
          sub.M = (int32_t)((d->M - m0) < mb ? (d->M - m0) : mb);
          sub.A = d->trans ? (const void*)((const bf16_t*)d->A + m0) : (const void*)((const bf16_t*)d->A + m0 * d->lda);
          sub.C = d->c_dtype == VDK_F32 ? (void*)((float*)d->C + m0 * d->ldc) : (void*)((bf16_t*)d->C + m0 * d->ldc);
          if (d->residual) sub.residual = d->residual + m0 * d->ldr;
          if (d->aux) sub.aux = (void*)((bf16_t*)d->aux + m0 * d->ldaux);
          if (d->row_scale) { if (m0 % d->rows_per_scale) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: row blocks do not align with rows_per_scale"); sub.row_scale = d->row_scale + m0 / d->rows_per_scale; }
          const int rc = vdk_gemm_bf16_nt(&sub, ws, ws_bytes, stream_);
          if (rc) return rc;
        }
        return VDK_OK;
      }
    }
  }
  if (d->conv) {
    const VdkConvGeom* c = d->conv;
    if (c->Cin <= 0 || (c->Cin & 7) || c->KH <= 0 || c->KW <= 0 || c->stride <= 0 || c->pad < 0 || c->OH <= 0 || c->OW <= 0 || c->H <= 0 || c->W <= 0)
      return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: bad convolution geometry (Cin % 8 == 0)");
    if (!d->trans && (d->K != c->KH * c->KW * c->Cin || (d->M % (c->OH * c->OW))))
      return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: bad convolution geometry (K == KH*KW*Cin, M == B*OH*OW)");
    // trans = 1: the weight gradient with the im2col operand gathered on the fly (VdkConvGeom.rows); the four-wave TN kernel serves it
    if (d->trans && (c->transposed || d->N != c->KH * c->KW * c->Cin || c->rows <= 0 || c->rows >= (1 << 24) || (c->rows % (c->OH * c->OW)) || d->K < c->rows || (d->K % 128) ||
                     d->K - c->rows >= 128 || d->a_row_group != 0))
      return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: bad weight-gradient geometry (N == KH*KW*Cin, rows == B*OH*OW < 2^24, K == rows rounded up to 128)");
  }
  if (d->ab_dtype != VDK_BF16 && d->ab_dtype != VDK_F16) return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: ab_dtype must be VDK_BF16 or VDK_F16");
  const int opf = d->ab_dtype == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16;
  // a 16-bit output is written in the operand format (c_dtype names it: VDK_BF16 with bf16 operands, VDK_F16 with fp16 operands)
  if (d->c_dtype != VDK_F32 && d->c_dtype != (opf ? VDK_F16 : VDK_BF16)) return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: bad c_dtype (VDK_F32, or the operand format)");
  if (opf && (d->a_colsum || d->splitk == -1))
    return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: fp16 operands are served by the four-wave and the 128x128 kernels (no a_colsum, no stream-K)");
  if (d->act < VDK_ACT_NONE || d->act > VDK_ACT_MUL_AUX) return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: bad act");
  if (((d->act == VDK_ACT_DGELU || d->act == VDK_ACT_GELU_SAVE_GRAD || d->act == VDK_ACT_MUL_AUX) && !d->aux) || (d->aux && (d->ldaux & 7)))
    return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: bad aux");
  int splitk = d->splitk < 1 ? 1 : d->splitk;
  GemmParams p = {};
  p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.opf = opf;
  p.c_dtype = d->c_dtype == VDK_F32 ? VDK_F32 : VDK_BF16; p.bias = (const float*)d->bias; p.residual = (const float*)d->residual; p.ldr = d->ldr;
  p.act = d->act; p.aux = (bf16_t*)d->aux; p.ldaux = d->ldaux; p.alpha = d->alpha; p.row_group = d->row_group < 0 ? -d->row_group : d->row_group; p.row_shift = d->row_group > 0 ? 1 : 0; p.a_row_group = d->a_row_group;
  p.splitk = splitk; p.slabs = nullptr; p.sk_cnt = nullptr; p.dbg = (unsigned long long*)g_dbg_ptr;
  p.colsum_part = nullptr; p.ocs_part = nullptr;
  p.cscale = d->col_scale;
  p.rscale = d->row_scale; p.rps = d->rows_per_scale;
  if (d->row_scale && (d->rows_per_scale <= 0 || d->c_dtype != VDK_F32 || !d->bias || !d->residual || d->act != VDK_ACT_NONE || d->splitk > 1 || d->trans || d->row_group != 0))
    return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: row_scale goes with rows_per_scale > 0, an fp32 output, a bias, a residual, act NONE and neither split-K, TN nor a row remap");
  if (d->col_scale && (d->c_dtype != VDK_F32 || !d->bias || d->act != VDK_ACT_NONE || d->splitk > 1 || d->trans || d->row_group != 0 || (d->N & 7)))
    return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: col_scale goes with an fp32 output, a bias, act NONE, N % 8 == 0 and neither split-K, TN nor a row remap");
  p.conv_on = d->conv != nullptr;
  if (d->conv) {
    const VdkConvGeom* c = d->conv;
    p.cCin = c->Cin; p.cH = c->H; p.cW = c->W; p.cOH = c->OH; p.cOW = c->OW; p.cKH = c->KH; p.cKW = c->KW; p.cstride = c->stride; p.cpad = c->pad; p.ctrans = c->transposed;
    p.crows = c->rows;
  }
  int kps = d->K;
  if (splitk > 1) {
    if (d->bias || d->residual || d->act != VDK_ACT_NONE || d->row_group != 0) return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: split-K excludes fused epilogues");
    if (d->ldc != d->N) return vdk_fail(VDK_EINVAL, "vdk_gemm_bf16_nt: split-K needs ldc == N");
    const int kq = (d->K % 128 == 0) ? 128 : G_BK;     // whole PAIRS of k-tiles per split where K allows it (the four-wave kernel multiplies k-tiles in pairs)
    kps = ((d->K + splitk - 1) / splitk + kq - 1) / kq * kq;
    splitk = (d->K + kps - 1) / kps;
    p.splitk = splitk;
    if (splitk > 1) {
      size_t need = (size_t)splitk * d->M * d->N * 4;
      if (!ws || ws_bytes < need) return vdk_fail(VDK_EWORKSPACE, "vdk_gemm_bf16_nt: split-K workspace too small");
      p.slabs = (float*)ws;
    }
  }
  p.k_per_split = kps;
  const int ntn = (d->N + G_BN - 1) / G_BN, ntm = (d->M + G_BM - 1) / G_BM;
  const bool prof = g_prof_on && g_prof_used + 2 <= g_prof_ev.size();
  // big problems go to the 256x256 LDS-DMA kernel (needs whole 64-wide k-tiles per split and >= 1 full wave of tiles)
  const long tiles256 = (long)((d->M + 255) / 256) * ((d->N + 255) / 256) * splitk;
  // ... or >= 128 of the 256x128 kernel's tiles (6 272 x 1024 outputs: 100 whole tiles, 200 half tiles; the 128x128 kernel ran those at 0.6-0.7 of the half-tile form)
  const long tiles_h = (!d->trans && splitk == 1 && !d->a_colsum && w4_enabled_env()) ? (long)((d->M + 255) / 256) * ((d->N + 127) / 128) : 0;
  // narrow outputs (128 <= N < 256, many rows: ConvNeXt's first stage, C = 128) are one column of 256x128 tiles for the two-workgroups-per-CU kernel
  const bool narrow = !d->conv && !d->trans && g_force_kernel == 0 && w4_enabled_env() && (d->K % 64 == 0) && (kps % 64 == 0) && d->N >= 128 && d->N < 256 && d->M >= narrow_min_m() &&
                      !d->a_colsum && !d->c_colsum && splitk == 1 && vdk_gemm_w4h_serves(p, false);
  const bool big = !d->conv && (narrow || g_force_kernel == 2 || g_force_kernel == 3 || g_force_kernel == 5 || g_force_kernel == 6 || ((g_force_kernel == 0 || g_force_kernel == 4) && (d->K % 64 == 0) && (kps % 64 == 0) && (tiles256 >= 128 || tiles_h >= 128) && d->M >= 256 && d->N >= 256));
  // compile-time epilogue variant (the common ViT forms); anything else takes the run-time-flag path
  int E = E_GENERIC;
  {
    const bool bias = d->bias != nullptr, res = d->residual != nullptr, f32 = p.c_dtype == VDK_F32, rg = d->row_group != 0;
    const bool plain_alpha = d->alpha == 1.0f;
    if (splitk > 1) E = E_SPLITK;
    else if (plain_alpha && !rg && !res && d->act == VDK_ACT_NONE && !f32) E = bias ? E_BIAS : 0;
    else if (plain_alpha && !rg && !res && d->act == VDK_ACT_GELU && !f32 && bias && d->aux) E = E_BIAS | E_GELU;
    else if (plain_alpha && !rg && !res && d->act == VDK_ACT_DGELU && !f32 && !bias) E = E_DGELU;
    else if (plain_alpha && !rg && !res && d->act == VDK_ACT_GELU_SAVE_GRAD && !f32 && bias) E = E_BIAS | E_GELU | E_AUXD;
    else if (plain_alpha && !rg && !res && d->act == VDK_ACT_MUL_AUX && !f32 && !bias) E = E_DGELU | E_AUXD;
    else if (plain_alpha && !rg && res && d->act == VDK_ACT_NONE && f32 && bias) E = E_BIAS | E_RES | E_F32;
    else if (plain_alpha && rg && res && d->act == VDK_ACT_NONE && f32 && bias) E = E_BIAS | E_RES | E_F32 | E_ROWGRP;
    else if (plain_alpha && !rg && !res && d->act == VDK_ACT_NONE && f32 && !bias) E = E_F32;
    else if (plain_alpha && !rg && !res && d->act == VDK_ACT_NONE && f32 && bias) E = E_BIAS | E_F32;      // ConvNeXt's stem / downsample convolutions
  }
  dim3 grid256((unsigned)(((d->M + 255) / 256) * ((d->N + 255) / 256)), (unsigned)splitk);
  // stream-K when the caller lent a workspace (splitk == -1) and whole-tile rounds would leave the last one mostly empty (N = 768 at T = 50 432: 591 tiles = 2.31 rounds)
  bool sk = false;
  if (d->splitk == -1 && !d->trans && !d->a_colsum && big && ws && g_force_kernel != 4) {   // (with the a_colsum by-product the persistent form spills)
    const long T = (long)grid256.x, G = sk_grid();
    const long rounds = (T + G - 1) / G;
    // Measured on MI355X (tools/bench_gemm_sk.py, round 2): the persistent form is NOT faster than hardware-dispatched rounds -- N = 768 / K = 3072 at T = 50 432:
    // 261 us against 225 us -- so it is opt-in (VDK_GEMM_STREAMK=1 or vdk_gemm_force_kernel(3)); see DESIGN.md section 6.
    static const bool optin = getenv("VDK_GEMM_STREAMK") && atoi(getenv("VDK_GEMM_STREAMK")) > 0;
    const bool worth = optin && T >= G && (double)T / (double)(rounds * G) < 0.94;
    if ((worth || g_force_kernel == 3) && T <= SK_CNT_BYTES / 4 && T * (d->K / 64) >= G && ws_bytes >= (size_t)SK_CNT_BYTES + (size_t)G * 2 * 65536 * 4) {
      sk = true;
      p.sk_cnt = (int*)ws; p.slabs = (float*)((char*)ws + SK_CNT_BYTES);
      grid256 = dim3((unsigned)G, 1u);
    }
  }
#define VDK_GEMM_LAUNCH(KERN, GRID, BLOCK)                                                                                                   \
  do {                                                                                                                                       \
    if (prof) hipExtLaunchKernelGGL(KERN, GRID, BLOCK, 0, stream, g_prof_ev[g_prof_used], g_prof_ev[g_prof_used + 1], 0, p);                 \
    else hipLaunchKernelGGL(KERN, GRID, BLOCK, 0, stream, p);                                                                                \
  } while (0)
#define LAUNCH256X(TNF, EE, CSF)                                                                                         \
  do {                                                                                                                   \
    if (opf) {   /* fp16 operands: the four-wave kernels' fp16 instantiation (gemm_w4_f16.hip), whichever form serves the problem */                 \
      void* e0_ = prof ? (void*)g_prof_ev[g_prof_used] : nullptr; void* e1_ = prof ? (void*)g_prof_ev[g_prof_used + 1] : nullptr;                \
      const bool hfirst_ = narrow || w4h_wanted(EE, TNF, p.M, p.N, p.K) || !(w4_enabled() && vdk_gemm_w4_serves(p, TNF));                         \
      if (hfirst_ && vdk_gemm_w4h_serves(p, TNF) && vdk_gemm_w4h_launch_f16(p, TNF, EE, grid256.y, stream, e0_, e1_)) { g_last_kernel = 6; break; }   \
      if (vdk_gemm_w4_serves(p, TNF) && vdk_gemm_w4_launch_f16(p, TNF, EE, grid256.x, grid256.y, stream, e0_, e1_)) { g_last_kernel = 5; break; }     \
      if (vdk_gemm_w4h_serves(p, TNF) && vdk_gemm_w4h_launch_f16(p, TNF, EE, grid256.y, stream, e0_, e1_)) { g_last_kernel = 6; break; }          \
      fp16_unserved = true; break;                                                                                        \
    }                                                                                                                    \
    if (!sk && !(CSF) && (narrow || w4h_wanted(EE, TNF, p.M, p.N, p.K)) && vdk_gemm_w4h_serves(p, TNF) &&                                           \
        vdk_gemm_w4h_launch(p, TNF, EE, grid256.y, stream, prof ? (void*)g_prof_ev[g_prof_used] : nullptr,               \
                            prof ? (void*)g_prof_ev[g_prof_used + 1] : nullptr)) { g_last_kernel = 6; break; }           \
    if (!sk && !(CSF) && w4_enabled() && vdk_gemm_w4_serves(p, TNF) &&                                                   \
        vdk_gemm_w4_launch(p, TNF, EE, grid256.x, grid256.y, stream, prof ? (void*)g_prof_ev[g_prof_used] : nullptr,     \
                           prof ? (void*)g_prof_ev[g_prof_used + 1] : nullptr)) { g_last_kernel = 5; break; }            \
    g_last_kernel = sk ? 3 : 2;                                                                                          \
    if (sk) VDK_GEMM_LAUNCH((gemm256_bf16_kernel<TNF, EE, !TNF, CSF>), grid256, dim3(512));                               \
    else VDK_GEMM_LAUNCH((gemm256_bf16_kernel<TNF, EE, false, CSF>), grid256, dim3(512));                                 \
  } while (0)
#define LAUNCH256(TNF, EE) LAUNCH256X(TNF, EE, false)
#define LAUNCH256CS(EE)                                                                                                  \
  do {                                                                                                                   \
    if (p.colsum_part) VDK_GEMM_LAUNCH((gemm256_bf16_kernel<false, EE, false, true>), grid256, dim3(512));                \
    else LAUNCH256X(false, EE, false);                                                                                   \
  } while (0)
  bool fp16_unserved = false;
  if (d->trans && d->conv) {      // the implicit weight gradient lives in the four-wave kernel only
    if ((kps % 128) || (d->M & 7) || d->M < 8 || !vdk_gemm_w4_serves(p, true)) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: this implicit weight gradient is outside the four-wave TN kernel's range");
    void* e0_ = prof ? (void*)g_prof_ev[g_prof_used] : nullptr; void* e1_ = prof ? (void*)g_prof_ev[g_prof_used + 1] : nullptr;
    const int ew_ = E == E_SPLITK ? E_SPLITK : (E == E_F32 ? E_F32 : E_GENERIC);
    if (!(opf ? vdk_gemm_w4_launch_f16(p, true, ew_, grid256.x, grid256.y, stream, e0_, e1_) : vdk_gemm_w4_launch(p, true, ew_, grid256.x, grid256.y, stream, e0_, e1_)))
      return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: implicit weight gradient: no kernel");
    g_last_kernel = 5;
  } else if (d->trans) {
    if ((d->K % 64) || (kps % 64) || (d->M & 7) || d->M < 8 || d->N < 8)
      return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: trans=1 needs K and the split size to be multiples of 64 and M % 8 == 0");
    switch (E) {
      case E_SPLITK: LAUNCH256(true, E_SPLITK); break;
      case E_F32: LAUNCH256(true, E_F32); break;
      case 0: LAUNCH256(true, 0); break;
      default: LAUNCH256(true, E_GENERIC); break;
    }
  } else if (big && (d->K % 64 == 0) && (kps % 64 == 0)) {
    if (d->a_colsum) {
      if (d->M % 256) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: a_colsum needs M % 256 == 0");
      p.colsum_part = d->a_colsum;
      if (E != 0 && E != E_DGELU && E != E_F32) E = E_GENERIC;   // the by-product is compiled into the dgrad forms only (E_DGELU | E_AUXD: run-time-flag path)
    }
    if (d->c_colsum) {   // compiled into the dGELU form (dL/du = bias gradient of fc1) and the plain bf16 form
      if ((E != E_DGELU && E != (E_DGELU | E_AUXD) && E != 0) || d->a_colsum) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: c_colsum goes with a plain or dGELU bf16 epilogue and without a_colsum");
      p.ocs_part = d->c_colsum;
      E |= E_OCS;
    }
    switch (E) {
      case E_OCS: LAUNCH256(false, E_OCS); break;
      case E_DGELU | E_OCS: LAUNCH256(false, E_DGELU | E_OCS); break;
      case 0: LAUNCH256CS(0); break;
      case E_BIAS: LAUNCH256(false, E_BIAS); break;
      case E_BIAS | E_GELU: LAUNCH256(false, E_BIAS | E_GELU); break;
      case E_DGELU: LAUNCH256CS(E_DGELU); break;
      case E_BIAS | E_GELU | E_AUXD: LAUNCH256(false, E_BIAS | E_GELU | E_AUXD); break;
      case E_DGELU | E_AUXD: LAUNCH256(false, E_DGELU | E_AUXD); break;
      case E_DGELU | E_OCS | E_AUXD: LAUNCH256(false, E_DGELU | E_OCS | E_AUXD); break;
      case E_BIAS | E_RES | E_F32: LAUNCH256(false, E_BIAS | E_RES | E_F32); break;
      case E_BIAS | E_F32: LAUNCH256(false, E_BIAS | E_F32); break;
      case E_BIAS | E_RES | E_F32 | E_ROWGRP: LAUNCH256(false, E_BIAS | E_RES | E_F32 | E_ROWGRP); break;
      case E_SPLITK: LAUNCH256(false, E_SPLITK); break;
      case E_F32: LAUNCH256CS(E_F32); break;
      default: LAUNCH256CS(E_GENERIC); break;
    }
  }
#undef LAUNCH256CS
#undef LAUNCH256X
#undef LAUNCH256
  else if (d->a_colsum || d->c_colsum)
    return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: a_colsum / c_colsum are by-products of the 256x256 NT kernel only (see vdk_gemm_a_colsum_rows)");
  else if (d->conv) {
    g_last_kernel = 1;
    if (opf) VDK_GEMM_LAUNCH((gemm_bf16_nt_kernel<true, VDK_OPF_F16>), dim3((unsigned)(ntn * ntm), (unsigned)splitk), dim3(256));
    else VDK_GEMM_LAUNCH(gemm_bf16_nt_kernel<true>, dim3((unsigned)(ntn * ntm), (unsigned)splitk), dim3(256));
  } else {
    g_last_kernel = 1;
    if (opf) VDK_GEMM_LAUNCH((gemm_bf16_nt_kernel<false, VDK_OPF_F16>), dim3((unsigned)(ntn * ntm), (unsigned)splitk), dim3(256));
    else VDK_GEMM_LAUNCH(gemm_bf16_nt_kernel<false>, dim3((unsigned)(ntn * ntm), (unsigned)splitk), dim3(256));
  }
  if (fp16_unserved) {
    // (a big NT problem the four-wave kernels cannot address -- operands beyond 2 GB -- still runs, on the 128x128 kernel; TN has no other home)
    if (d->trans) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_bf16_nt: this fp16 TN problem is outside the four-wave kernels' range");
    g_last_kernel = 1;
    VDK_GEMM_LAUNCH((gemm_bf16_nt_kernel<false, VDK_OPF_F16>), dim3((unsigned)(ntn * ntm), (unsigned)splitk), dim3(256));
  }
  if (prof) {   // the GEMM kernel only (the split-K combine is a separate, HBM-bound kernel)
    g_prof_flops.push_back(2.0 * d->M * d->N * d->K);
    {
      const double mn = (double)d->M * d->N;
      double b = ((double)d->M * d->K + (double)d->N * d->K) * 2.0 + mn * (p.c_dtype == VDK_F32 ? 4.0 : 2.0);
      if (d->residual) b += mn * 4.0;
      if (d->aux) b += mn * 2.0;
      if (d->bias) b += d->N * 4.0;
      if (d->splitk > 1) b += 2.0 * d->splitk * mn * 4.0;   // slabs written, then read by the reduce
      g_prof_bytes += b;
    }
    g_prof_used += 2;
  }
  if (splitk > 1) {
    long mn = (long)d->M * d->N, n4 = mn / 4;
    if (opf) hipLaunchKernelGGL(splitk_reduce_kernel<VDK_OPF_F16>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float*)p.slabs, splitk, n4, mn, d->alpha,
                                p.c_dtype == VDK_F32 ? (float*)d->C : (float*)nullptr, p.c_dtype == VDK_BF16 ? (bf16_t*)d->C : (bf16_t*)nullptr);
    else
    hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream,
                       (const float*)p.slabs, splitk, n4, mn, d->alpha,
                       p.c_dtype == VDK_F32 ? (float*)d->C : (float*)nullptr,
                       p.c_dtype == VDK_BF16 ? (bf16_t*)d->C : (bf16_t*)nullptr);
  }
  return vdk_check_launch("vdk_gemm_bf16_nt");
}

// The cos GEMM of the margin heads with the head applied to the tile in registers ("B x C logits are never written").  cos[Bp, Cp] = f^ W^ is the TN product of
// fbt [K, Bp] (the normalised features, transposed; K = D or 3 D split planes) and wb [K, Cp] (the column-normalised weight); nothing of it reaches memory as fp32:
//   pass 1 (E_MSTAT): per (row, 64-column slice) partials (max logit, sum exp(logit - max), sum logit) -> stats f32 [B][ceil(Cp / 64)][4], the target's logit -> tlogit [B];
//                     vdk_margin_rowstat (margin_head.hip) reduces them to the row statistics and the loss;
//   pass 2 (E_MGRAD): the same product again, its epilogue writes d(loss)/d(cos) as bf16 [Bp, lddc] from the row statistics (rows >= B and columns >= C zero).
// Needs the 256x256 TN kernel: K % 64 == 0, Bp % 8 == 0, Cp % 8 == 0; returns VDK_EUNSUPPORTED otherwise (callers keep the materialised form).
int vdk_margin_cos_pass(const VdkMarginHead* h, int32_t pass, const void* fbt, int64_t ld_f, const void* wb, int64_t ld_w, int32_t B, int32_t Bp, int32_t C, int32_t Cp, int32_t K,
                        const int64_t* labels, const float* gt, float* stats, float* tlogit, const float* rowstat, float label_smoothing, float grad_scale, void* dcos,
                        int64_t lddc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!h || !fbt || !wb || !labels || B <= 0 || Bp < B || C <= 0 || Cp < C || (pass != 1 && pass != 2)) return vdk_fail(VDK_EINVAL, "vdk_margin_cos_pass: bad argument");
  if ((pass == 1 && (!stats || !tlogit)) || (pass == 2 && (!rowstat || !dcos))) return vdk_fail(VDK_EINVAL, "vdk_margin_cos_pass: missing buffer for this pass");
  if ((K % 64) || (Bp & 7) || (Cp & 7) || (ld_f & 7) || (ld_w & 7) || (pass == 2 && (lddc & 7)) || Bp < 8 || Cp < 256)
    return VDK_EUNSUPPORTED;
  GemmParams p = {};
  p.A = (const bf16_t*)fbt; p.B = (const bf16_t*)wb; p.C = dcos; p.lda = ld_f; p.ldb = ld_w; p.ldc = lddc; p.M = Bp; p.N = Cp; p.K = K; p.c_dtype = VDK_BF16;
  p.alpha = 1.0f; p.splitk = 1; p.k_per_split = K;
  RC_MARGIN(fill_params(h, &p.me.P));
  p.me.labels = (const long long*)labels; p.me.gt = gt; p.me.stats = stats; p.me.nslice = (Cp + 63) / 64; p.me.tlogit = tlogit; p.me.rowstat = rowstat;
  p.me.smoothing = label_smoothing; p.me.gscale = grad_scale; p.me.epsc = label_smoothing / (float)C; p.me.B = B; p.me.C = C;
  const dim3 grid((unsigned)(((Bp + 255) / 256) * ((Cp + 255) / 256)), 1u);
  g_last_kernel = 2;
  if (w4_enabled() && vdk_gemm_w4_serves(p, true) && vdk_gemm_w4_launch(p, true, pass == 1 ? E_MSTAT : E_MGRAD, grid.x, 1u, stream, nullptr, nullptr)) g_last_kernel = 5;
  else if (pass == 1) hipLaunchKernelGGL((gemm256_bf16_kernel<true, E_MSTAT, false, false>), grid, dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((gemm256_bf16_kernel<true, E_MGRAD, false, false>), grid, dim3(512), 0, stream, p);
  return vdk_check_launch("vdk_margin_cos_pass");
}

// out[C][ldo] (bf16) = in[R][ldi]^T, rows R..Rpad-1 of the contraction dim zero-filled.  Rpad even.
// in_row_group > 0: logical row r is read from physical row r + r / in_row_group + 1 (token buffer minus cls rows).
// colsum_partial (optional): f32 [ceil(Rpad/64)][C] per-row-tile column sums of `in` (Linear bias gradient); reduce with
// vdk_reduce_rows_f32(colsum_partial, C, ceil(Rpad/64), C, db, 1).
int vdk_transpose_bf16(const void* in, int64_t ldi, int32_t R, int32_t Cc, void* out, int64_t ldo, int32_t Rpad,
                       int32_t in_row_group, float* colsum_partial, void* stream) {
  return vdk_transpose_16(in, ldi, R, Cc, out, ldo, Rpad, in_row_group, colsum_partial, VDK_OPF_BF16, stream);
}

}  // extern "C"

// (the transpose itself moves 16-bit words whatever they encode; only the column-sum by-product reads values: opf = their format)
int vdk_transpose_16(const void* in, int64_t ldi, int32_t R, int32_t Cc, void* out, int64_t ldo, int32_t Rpad, int32_t in_row_group, float* colsum_partial, int opf,
                     void* stream) {
  if (!in || !out || R < 0 || Cc <= 0 || Rpad < R || (Rpad & 1) || ldo < Rpad || (ldi & 1) || (ldo & 1))
    return vdk_fail(VDK_EINVAL, "vdk_transpose_bf16: bad argument");
  if (Rpad == 0) return VDK_OK;
  const dim3 grid((unsigned)((Rpad + 63) / 64), (unsigned)((Cc + 63) / 64));
  if (opf) hipLaunchKernelGGL(transpose_bf16_kernel<VDK_OPF_F16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (long)ldi, (int)R, (int)Cc, (bf16_t*)out,
                              (long)ldo, (int)Rpad, (int)in_row_group, colsum_partial);
  else hipLaunchKernelGGL(transpose_bf16_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (long)ldi, (int)R, (int)Cc, (bf16_t*)out, (long)ldo,
                          (int)Rpad, (int)in_row_group, colsum_partial);
  return vdk_check_launch("vdk_transpose_bf16");
}

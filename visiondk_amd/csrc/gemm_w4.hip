// gemm_w4.hip — K2, second structure: the 256x256x64 bf16 GEMM as FOUR waves, one per SIMD.
//
// Same product as gemm.hip (C[M,N] = A[M,K] * B[N,K]^T, or C = A^T B for the weight gradients; every Linear of the hot path,
// reference call sites models/classifier/classify_model.py:49-54 -> timm vision_transformer), different machine mapping:
//   * one workgroup = 4 waves = the CU's 4 SIMDs, one wave each with the whole 512-register file: wave tile 128x128 = 4x4
//     v_mfma_f32_32x32x16_bf16 accumulators = 256 AGPRs.  Per 64-deep k-tile a wave reads 32 fragments for 64 MFMAs (the 8-wave
//     kernel: 48 per 64) and no second wave competes for its SIMD's matrix pipe.
//   * operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds): per-lane offsets are loop-invariant, the k / tile position lives in
//     the scalar offset, rows beyond the matrix read as zero through the descriptor's bounds check.  LDS image and bank swizzle
//     as in gemm.hip (source-side involution, cdna_hip_programming.md rule 21).
//   * a k-tile is 4 regions of 16 KB (RA0 / RA1 = first / second 64 rows of both wave-rows, RB0 / RB1 likewise for the wave
//     columns) consumed in 4 phases of 16 MFMAs: A0B0, A0B1, A1B1, A1B0.  The fragments of a phase are read from LDS one phase
//     EARLIER, behind the previous phase's MFMAs, so a region is free again a phase before its tile is multiplied and its
//     refill (k-tile t+2) has 6 phases (~3 k cycles) to land with only two tile buffers (128 KB):
//         phase g reads the region whose DMA group was issued in phase g-6; every phase issues one group of 4 pieces per wave
//         => the single wait per phase is vmcnt(20) ("all but the 5 newest groups"), never 0 in steady state.
//   * 32 KB of LDS beside the operand ring stage the epilogue (4 waves x 32 rows x 64 fp32), so the ring is never torn down.
#include <hip/hip_runtime.h>
#ifndef VDK_EMU_NO_HIP_EXT
#include <hip/hip_ext.h>
#endif
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_gemm.h"
#include "vdk_gemm_epilogue.h"

#define W4_REGION 16384
#define W4_TILEBUF 65536
#define W4_STAGE 131072
#define W4_SMEM 163840
#define W4_RA0 0
#define W4_RA1 16384
#define W4_RB0 32768
#define W4_RB1 49152

#define W4_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))   /* vmcnt(n), n < 64; expcnt / lgkmcnt untouched */
#define W4_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define W4_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// ---- LDS-DMA of one operand: scalar state + loop-invariant lane offsets -------------------------------------------------------------
// piece j (0..3) of wave w fills LDS bytes [(4w + j) * 1024, +1024) of a region.
//   NT (operand [rows, K] row-major): the piece is region rows 32w + 8j .. +7 (lane: row i = lane >> 3, 16-B position cp = lane & 7, which holds global
//      chunk cp ^ (4 (j & 1) + (i >> 1)) = cp ^ ((row >> 1) & 7)); region row r is operand row (r >> 6) * 128 + sub * 64 + (r & 63).
//   TN (operand [K, cols] row-major): the piece is k-rows 4 (4w + j) .. +3 of the region's [64 k][128 cols] image (lane: k-row lane >> 4, position
//      cp = lane & 15 holding chunk c = cp ^ (4 (k-row & 3)), i.e. operand columns (c >> 3) * 128 + sub * 64 + (c & 7) * 8 .. +7).
struct W4Dma {
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff0, voff1;     // lane byte offsets of even / odd pieces (TN: equal)
  unsigned sbase;            // scalar: tile origin + this wave's share, bytes
  unsigned sub_stride;       // scalar: bytes from the sub 0 region's source to the sub 1 region's
  unsigned piece_stride;     // scalar: bytes between consecutive pieces
  unsigned kpos;             // scalar: byte offset of the NEXT k-tile to be issued
  unsigned kstep;            // scalar: bytes per k-tile
};

template <bool TN>
__device__ __forceinline__ void w4_dma_init(W4Dma& d, const bf16_t* base, long ld, int origin /* first operand row (NT) / column (TN) of the tile */, int extent /* operand rows (NT) */,
                                            int K, int kbeg, int w, int lane) {
  const unsigned ldb = (unsigned)ld * 2u;
  if (!TN) {
    d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((unsigned)extent * ldb), 0x00020000);
    const unsigned i = (unsigned)lane >> 3, cp = (unsigned)lane & 7u;
    d.voff0 = i * ldb + ((cp ^ (i >> 1)) << 4);
    d.voff1 = i * ldb + ((cp ^ (4u + (i >> 1))) << 4);
    d.sbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)origin + (unsigned)(w >> 1) * 128u + (unsigned)(w & 1) * 32u) * ldb));
    d.sub_stride = 64u * ldb;
    d.piece_stride = 8u * ldb;
    d.kpos = (unsigned)kbeg * 2u;
    d.kstep = 128u;
  } else {
    d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((unsigned)K * ldb), 0x00020000);
    const unsigned kr = (unsigned)lane >> 4, cp = (unsigned)lane & 15u, c = cp ^ (4u * (kr & 3u));
    d.voff0 = d.voff1 = kr * ldb + ((c >> 3) * 128u + (c & 7u) * 8u) * 2u;
    d.sbase = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)origin * 2u + (unsigned)w * 16u * ldb));
    d.sub_stride = 128u;
    d.piece_stride = 4u * ldb;
    d.kpos = (unsigned)kbeg * ldb;
    d.kstep = 64u * ldb;
  }
}
// one piece: j = 0..3 of the region `sub` (0 / 1) of the k-tile at d.kpos + kahead * d.kstep, into LDS at dst (wave-uniform)
__device__ __forceinline__ void w4_piece(const W4Dma& d, unsigned char* dst, int sub, int j, unsigned kofs) {
  const unsigned so = d.sbase + (unsigned)sub * d.sub_stride + (unsigned)j * d.piece_stride + kofs;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rs, VDK_LDS_PTR(dst), 16, (j & 1) ? d.voff1 : d.voff0, so, 0, 0);
}
__device__ __forceinline__ void w4_region(const W4Dma& d, unsigned char* region, int sub, int w, unsigned kofs) {
#pragma unroll
  for (int j = 0; j < 4; ++j) w4_piece(d, region + (4 * w + j) * 1024, sub, j, kofs);
}

// ---- fragments -----------------------------------------------------------------------------------------------------------------------
// NT: off[ks] = byte offset of the lane's 16 B of k-step ks in region row (wrow * 64 + l31); the second 32-row block is +4096.
// TN: off[rt] = byte offset of the lane's first transpose read of k-step 0 for the 32-column block rt; k-step ks is + ks * 4096, the second read +1024.
template <bool TN>
struct W4Frag { unsigned a[TN ? 2 : 4], b[TN ? 2 : 4]; };

template <bool TN>
__device__ __forceinline__ void w4_frag_init(W4Frag<TN>& f, int wr, int wc, int lane) {
  const unsigned l31 = lane & 31, hi = lane >> 5;
  if (!TN) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned ch = ((unsigned)(ks * 2) + hi) ^ ((l31 >> 1) & 7u);
      f.a[ks] = ((unsigned)wr * 64u + l31) * 128u + ch * 16u;
      f.b[ks] = ((unsigned)wc * 64u + l31) * 128u + ch * 16u;
    }
  } else {
    const unsigned s = lane & 15, chalf = (lane >> 4) & 1, t1 = hi * 8u + (s >> 2);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const unsigned ca = (unsigned)wr * 64u + (unsigned)rt * 32u + 16u * chalf + 4u * (s & 3u);
      const unsigned cb = (unsigned)wc * 64u + (unsigned)rt * 32u + 16u * chalf + 4u * (s & 3u);
      f.a[rt] = t1 * 256u + (((ca >> 3) ^ (4u * (t1 & 3u))) * 16u) + ((ca >> 2) & 1u) * 8u;
      f.b[rt] = t1 * 256u + (((cb >> 3) ^ (4u * (t1 & 3u))) * 16u) + ((cb >> 2) & 1u) * 8u;
    }
  }
}
template <bool TN>
__device__ __forceinline__ s16x8 w4_frag(const unsigned char* region, const unsigned* off, int rt, int ks) {
  if (!TN) {
    return *(const s16x8*)(region + off[ks] + rt * 4096);
  } else {
    const unsigned char* p1 = region + off[rt] + ks * 4096;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1));
    s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1 + 1024));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
    return r;
  }
}

// ---- one phase: 16 MFMAs (2 x 2 accumulators x 4 k-steps) with 8 fragment reads and 4 DMA pieces spread behind them -------------------
// RD: 0 none, 1 read 8 fragments from `rd_region` through offsets rd_off into RDST[2][4].  IS: issue the 4 pieces of (dma, is_region, is_sub).
template <bool TN, bool RD, bool IS>
__device__ __forceinline__ void w4_phase(f32x16 (&c00), f32x16 (&c01), f32x16 (&c10), f32x16 (&c11), const s16x8 (&X)[2][4], const s16x8 (&Y)[2][4],
                                         const unsigned char* rd_region, const unsigned* rd_off, s16x8 (&RDST)[2][4],
                                         const W4Dma& dma, unsigned char* is_region, int is_sub, int w) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[0][ks], Y[0][ks], c00, 0, 0, 0);
    if (RD) RDST[0][ks] = w4_frag<TN>(rd_region, rd_off, 0, ks);
    c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[0][ks], Y[1][ks], c01, 0, 0, 0);
    if (RD) RDST[1][ks] = w4_frag<TN>(rd_region, rd_off, 1, ks);
    c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[1][ks], Y[0][ks], c10, 0, 0, 0);
    if (IS) w4_piece(dma, is_region + (4 * w + ks) * 1024, is_sub, ks, dma.kpos + 2u * dma.kstep);   // k-tile t+2 (kpos is the tile being multiplied)
    c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[1][ks], Y[1][ks], c11, 0, 0, 0);
  }
  // the order above is the order wanted in the instruction stream: one LDS read (TN: one pair) or one DMA piece in the shadow of each MFMA, never a batch in
  // front of the phase (left alone, the scheduler hoists all 8 reads and 4 pieces above the first MFMA: ~100 idle matrix-pipe cycles per phase)
#ifndef VDK_EMU
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (RD) __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (RD) __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (IS) { __builtin_amdgcn_sched_group_barrier(0x004, 3, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  }
#endif
}

// ---- one k-tile -----------------------------------------------------------------------------------------------------------------------
// CUR: buffer of this k-tile.  V2..V4 / VN: vmcnt that must hold before the barrier of phases 2..4 / of the NEXT k-tile's phase 1 (-1: no wait); each is waited
// for at the END of the phase before, so it never sits between a phase's reads and its MFMAs.  ISSUE: refill this buffer with k-tile t+2.  NEXT: prefetch the
// first fragments (A0, B0) of k-tile t+1 in phases 3 / 4.  B0 holds this tile's B0 fragments, B0N receives the next tile's.
template <int V>
__device__ __forceinline__ void w4_phase_end() {
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (V >= 0) W4_WAIT_VM(V);
  W4_WAIT_LGKM0();          // this phase's fragment reads have returned: the next phase may multiply them, and (after its barrier) anyone may overwrite their region
}
template <bool TN, int CUR, int V2, int V3, int V4, int VN, bool ISSUE, bool NEXT>
__device__ __forceinline__ void w4_ktile(unsigned char* smem, const W4Frag<TN>& F, W4Dma& da, W4Dma& db, int w, f32x16 (&acc)[4][4],
                                         s16x8 (&A0)[2][4], s16x8 (&A1)[2][4], s16x8 (&B0)[2][4], s16x8 (&B1)[2][4], s16x8 (&B0N)[2][4]) {
  unsigned char* const buf = smem + CUR * W4_TILEBUF;
  unsigned char* const nbuf = smem + (CUR ^ 1) * W4_TILEBUF;
  // P1: A0 x B0; read B1(t); refill RA0 (read in P3 of the previous tile)
  W4_BAR();
  w4_phase<TN, true, ISSUE>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], A0, B0, buf + W4_RB1, F.b, B1, da, buf + W4_RA0, 0, w);
  w4_phase_end<V2>();
  // P2: A0 x B1; read A1(t); refill RB0 (read in P4 of the previous tile)
  W4_BAR();
  w4_phase<TN, true, ISSUE>(acc[0][2], acc[0][3], acc[1][2], acc[1][3], A0, B1, buf + W4_RA1, F.a, A1, db, buf + W4_RB0, 0, w);
  w4_phase_end<V3>();
  // P3: A1 x B1; read A0(t+1); refill RB1 (read in P1)
  W4_BAR();
  w4_phase<TN, NEXT, ISSUE>(acc[2][2], acc[2][3], acc[3][2], acc[3][3], A1, B1, nbuf + W4_RA0, F.a, A0, db, buf + W4_RB1, 1, w);
  w4_phase_end<V4>();
  // P4: A1 x B0; read B0(t+1); refill RA1 (read in P2)
  W4_BAR();
  w4_phase<TN, NEXT, ISSUE>(acc[2][0], acc[2][1], acc[3][0], acc[3][1], A1, B0, nbuf + W4_RB0, F.b, B0N, da, buf + W4_RA1, 1, w);
  w4_phase_end<VN>();
  da.kpos += da.kstep; db.kpos += db.kstep;
}

template <bool TN, int E>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[W4_SMEM];   // 128 KB operand ring + 32 KB epilogue staging: the CU's whole LDS, one object
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1, hi = lane >> 5, l31 = lane & 31;
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  const int nwg = ntn * ntm;
  const int z = blockIdx.y;
  int tile;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tn = tile % ntn, tm = tile / ntn;
  const int m0 = tm * 256, n0 = tn * 256;
  const int kbeg = z * p.k_per_split;
  int kend = kbeg + p.k_per_split; if (kend > p.K) kend = p.K;
  const int nk = (kend - kbeg) / 64;                      // launcher guarantees whole k-tiles

  W4Dma da, db;
  w4_dma_init<TN>(da, p.A, p.lda, m0, p.M, p.K, kbeg, w, lane);
  w4_dma_init<TN>(db, p.B, p.ldb, n0, p.N, p.K, kbeg, w, lane);
  W4Frag<TN> F;
  w4_frag_init<TN>(F, wr, wc, lane);

  f32x16 acc[4][4];
  s16x8 A0[2][4], A1[2][4], B1[2][4], BX[2][4], BY[2][4];

  if (nk > 0) {
    // ---- prologue: k-tiles 0 and 1 completely, in the steady state's issue order (RA0, RB0, RB1, RA1); k-tile nk-2 lives in buffer 0, nk-1 in buffer 1 ----
    const int b0 = nk & 1;
    {
      unsigned char* const buf = smem + b0 * W4_TILEBUF;
      w4_region(da, buf + W4_RA0, 0, w, da.kpos); w4_region(db, buf + W4_RB0, 0, w, db.kpos);
      w4_region(db, buf + W4_RB1, 1, w, db.kpos); w4_region(da, buf + W4_RA1, 1, w, da.kpos);
    }
    if (nk > 1) {
      unsigned char* const buf = smem + (b0 ^ 1) * W4_TILEBUF;
      w4_region(da, buf + W4_RA0, 0, w, da.kpos + da.kstep); w4_region(db, buf + W4_RB0, 0, w, db.kpos + db.kstep);
      w4_region(db, buf + W4_RB1, 1, w, db.kpos + db.kstep); w4_region(da, buf + W4_RA1, 1, w, da.kpos + da.kstep);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nk > 1) { W4_WAIT_VM(24); } else { W4_WAIT_VM(8); }
    W4_BAR();
    const int S = nk >= 2 ? nk - 2 : 0;                   // steady k-tiles (refill + next-tile prefetch)
    const bool odd = S & 1;                               // the first k-tile then sits in buffer 1 and starts from the BY fragment set
    {
      const unsigned char* const buf = smem + b0 * W4_TILEBUF;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) A0[rt][ks] = w4_frag<TN>(buf + W4_RA0, F.a, rt, ks);
      if (b0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) BY[rt][ks] = w4_frag<TN>(buf + W4_RB0, F.b, rt, ks);
      } else {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) BX[rt][ks] = w4_frag<TN>(buf + W4_RB0, F.b, rt, ks);
      }
    }
    // (b0 == 1  <=>  nk odd  <=>  S odd for nk >= 2; for nk == 1 the only k-tile is the buffer-1 tail below, which reads BY)
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) { W4_WAIT_VM(20); } else { W4_WAIT_VM(4); }   // RB1 of k-tile 0, read in its phase 1
    W4_WAIT_LGKM0();
    int done = 0;
    if (odd) { w4_ktile<TN, 1, 20, 20, 20, 20, true, true>(smem, F, da, db, w, acc, A0, A1, BY, B1, BX); done = 1; }
    for (; done < S; done += 2) {
      w4_ktile<TN, 0, 20, 20, 20, 20, true, true>(smem, F, da, db, w, acc, A0, A1, BX, B1, BY);
      w4_ktile<TN, 1, 20, 20, 20, 20, true, true>(smem, F, da, db, w, acc, A0, A1, BY, B1, BX);
    }
    if (nk > 1) w4_ktile<TN, 0, 16, 12, 8, 4, false, true>(smem, F, da, db, w, acc, A0, A1, BX, B1, BY);
    w4_ktile<TN, 1, 0, -1, -1, -1, false, false>(smem, F, da, db, w, acc, A0, A1, BY, B1, BX);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  // ---- epilogue: wave-private 8 KB staging slab, 32 rows x 64 fp32 at a time -> 8-wide coalesced row chunks (vdk_gemm_epilogue.h) -----------------
  float* const slab = (float*)(smem + W4_STAGE + w * 8192);
  float q8_unused = 0.f;
#pragma unroll
  for (int cp = 0; cp < 2; ++cp) {
    const int ncol = n0 + wc * 128 + cp * 64 + (lane & 7) * 8;
    float bias8[8], ocs8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; ocs8[e] = 0.f; }
    if ((E != E_GENERIC) && (E & E_BIAS) && ncol < p.N) {
      f32x4 b0v = *(const f32x4*)(p.bias + ncol), b1v = *(const f32x4*)(p.bias + ncol + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[e] = b0v[e]; bias8[4 + e] = b1v[e]; }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + ct * 32 + l31] = acc[rt][cp * 2 + ct][r];
      __builtin_amdgcn_wave_barrier();
      const long mbase = (long)m0 + wr * 128 + rt * 32;
      if (E != E_GENERIC) {
        h_epilogue_half<E, 4>(p, slab, lane, mbase, ncol, z, bias8, ocs8, q8_unused);
      } else {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int row = pass * 8 + (lane >> 3), cc = (lane & 7) * 8;
          const long mi = mbase + row;
          if (mi < p.M && ncol < p.N) {
            float v[8];
            f32x4 x0 = *(const f32x4*)(slab + row * 64 + cc), x1 = *(const f32x4*)(slab + row * 64 + cc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
            g_epilogue_store8(p, mi, ncol, v, z);
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
  }
}

// ---- launcher (called by vdk_gemm_bf16_nt / vdk_margin_cos_pass in gemm.hip) ------------------------------------------------------------------------
// Serves what the 8-wave 256x256 kernel serves except its a_colsum by-product and stream-K.  Operand byte sizes stay below 2 GB (32-bit buffer offsets).
bool vdk_gemm_w4_serves(const GemmParams& p, bool trans) {
  const double lim = 2147483648.0 - 4096.0;
  if (p.colsum_part || p.sk_cnt || p.a_row_group > 0 || p.q8) return false;
  if (!trans) return ((double)p.M + 256.0) * (double)p.lda * 2.0 < lim && ((double)p.N + 256.0) * (double)p.ldb * 2.0 < lim;
  return ((double)p.K + 64.0) * (double)p.lda * 2.0 < lim && ((double)p.K + 64.0) * (double)p.ldb * 2.0 < lim;
}

#define W4_LAUNCH(TNF, EE)                                                                                                          \
  do {                                                                                                                              \
    if (ev0) hipExtLaunchKernelGGL((gemm_w4_kernel<TNF, EE>), grid, dim3(256), 0, stream, (hipEvent_t)ev0, (hipEvent_t)ev1, 0, p);  \
    else hipLaunchKernelGGL((gemm_w4_kernel<TNF, EE>), grid, dim3(256), 0, stream, p);                                              \
    return true;                                                                                                                    \
  } while (0)

bool vdk_gemm_w4_launch(const GemmParams& p, bool trans, int E, unsigned tiles, unsigned splitk, void* stream_, void* ev0, void* ev1) {
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid(tiles, splitk);
  if (trans) {
    switch (E) {
      case E_SPLITK: W4_LAUNCH(true, E_SPLITK);
      case E_F32: W4_LAUNCH(true, E_F32);
      case 0: W4_LAUNCH(true, 0);
      case E_MSTAT: W4_LAUNCH(true, E_MSTAT);
      case E_MGRAD: W4_LAUNCH(true, E_MGRAD);
      default: W4_LAUNCH(true, E_GENERIC);
    }
  }
  switch (E) {
    case 0: W4_LAUNCH(false, 0);
    case E_OCS: W4_LAUNCH(false, E_OCS);
    case E_DGELU | E_OCS: W4_LAUNCH(false, E_DGELU | E_OCS);
    case E_BIAS: W4_LAUNCH(false, E_BIAS);
    case E_BIAS | E_GELU: W4_LAUNCH(false, E_BIAS | E_GELU);
    case E_DGELU: W4_LAUNCH(false, E_DGELU);
    case E_BIAS | E_RES | E_F32: W4_LAUNCH(false, E_BIAS | E_RES | E_F32);
    case E_BIAS | E_RES | E_F32 | E_ROWGRP: W4_LAUNCH(false, E_BIAS | E_RES | E_F32 | E_ROWGRP);
    case E_SPLITK: W4_LAUNCH(false, E_SPLITK);
    case E_F32: W4_LAUNCH(false, E_F32);
    default: W4_LAUNCH(false, E_GENERIC);
  }
  return false;
}

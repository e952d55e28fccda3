// gemm_w4.hip — K2, second structure: the 256x256x64 bf16 GEMM as FOUR waves, one per SIMD, persistent over its output tiles.
//
// Same product as gemm.hip (C[M,N] = A[M,K] * B[N,K]^T, or C = A^T B for the weight gradients; every Linear of the hot path,
// reference call sites models/classifier/classify_model.py:49-54 -> timm vision_transformer), different machine mapping:
//   * one workgroup = 4 waves = the CU's 4 SIMDs, one wave each with the whole 512-register file: wave tile 128x128 = 4x4
//     v_mfma_f32_32x32x16_bf16 accumulators = 256 AGPRs.  Per 64-deep k-tile a wave reads 32 fragments for 64 MFMAs (the 8-wave
//     kernel: 48 per 64) and no second wave competes for its SIMD's matrix pipe (measured at 8192^3: 78 % of the elapsed cycles
//     are MFMA cycles against 70 %; the chip then clocks to its power budget).
//   * operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds, inline asm: the compiler neither waits for it nor serialises LDS
//     reads behind it): per-lane offsets are loop-invariant, the k / tile position lives in the scalar offset, rows beyond the
//     matrix read as zero through the descriptor's bounds check.  LDS image and bank swizzle as in gemm.hip (source-side
//     involution, cdna_hip_programming.md rule 21).
//   * a k-tile is 4 regions of 16 KB (RA0 / RA1 = first / second 64 rows of both wave-rows, RB0 / RB1 likewise for the wave
//     columns) consumed in 4 phases of 16 MFMAs: A0B0, A0B1, A1B1, A1B0.  The fragments of a phase are read from LDS one phase
//     EARLIER, behind the previous phase's MFMAs, so a region is free again a phase before its tile is multiplied and its
//     refill (k-tile t+2 of the stream) has 6 phases (~3 k cycles) to land with only two tile buffers (128 KB):
//         phase g reads the region whose DMA group was issued in phase g-6; every phase issues one group of 4 pieces per wave
//         => the single wait per phase is vmcnt(20) ("all but the 5 newest groups"), never 0 in steady state.
//   * the k-tile stream does not stop at an output tile's end: a workgroup walks its tiles (XCD-local raster) and the DMA cursor,
//     two k-tiles ahead of the MFMAs, moves on to the next tile's operands while the current tile finishes, so the next tile's
//     first operands land and its first fragments are read under the current tile's last MFMAs and its epilogue.
//   * the MFMAs take the B fragment as their first operand: a lane then owns ONE output row and groups of 4 consecutive columns,
//     so the epilogue packs bf16 rows with 8-byte LDS writes into 32 KB of staging beside the operand ring (never torn down) and
//     stores whole 256-byte row segments.
#include <hip/hip_runtime.h>
#ifndef VDK_EMU_NO_HIP_EXT
#include <hip/hip_ext.h>
#endif
#include "vdk_device.h"
#include "vdk_host.h"
#include <atomic>
#include <string.h>
#include <stdlib.h>
#include "vdk_gemm.h"
#include "vdk_gemm_epilogue.h"

// VDK_W4_OF: the operand format this translation unit instantiates (vdk_device.h: 0 = bf16, 1 = fp16).  gemm_w4_f16.hip defines it to 1 and includes this file:
// the kernels are templates over OF (distinct symbols), the two launchers get an _f16 suffix there, and the format-independent host helpers are compiled here only.
#ifndef VDK_W4_OF
#define VDK_W4_OF 0
#endif
#if VDK_W4_OF == 0
#define W4_SYM(name) name
#else
#define W4_SYM(name) name##_f16
#endif

#define W4_REGION 16384
#define W4_TILEBUF 65536
#define W4_STAGE 131072
#define W4_SMEM 163840
#define W4_RA0 0
#define W4_RA1 16384
#define W4_RB0 32768
#define W4_RB1 49152
#define W4_OOB 0x80000000u
#ifndef W4_AUX_PREFETCH
#define W4_AUX_PREFETCH 0   /* dGELU forms of the persistent kernel: the first block's saved-operand rows are requested in front of the tile's last k-tile (0: at the epilogue's start) */
#endif
#ifndef W4_STORE_POLICY
#define W4_STORE_POLICY 2   /* cache policy of the 16-bit epilogue stores: 2 = nt (see rows_out), 0 = default (A/B builds) */
#endif
#ifndef W4_EPI_AHEAD
#define W4_EPI_AHEAD 1      /* epilogue operand rows (saved GELU operand, fp32 residual) requested one 32-row block ahead; 0 = at the block's start (A/B builds) */
#endif   /* scalar offset of a DMA that must read nothing: beyond every descriptor (operands stay below 2 GB), so it lands zeros */

#define W4_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))   /* vmcnt(n), n < 64; expcnt / lgkmcnt untouched */
#define W4_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define W4_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// ---- LDS-DMA -----------------------------------------------------------------------------------------------------------------------
// One piece = one buffer_load_dwordx4 ... lds: 64 lanes x 16 B land at LDS [dst, dst + 1024) in lane order; the source of lane l is
// descriptor base + voff(l) + soff.  The compiler does not see the instruction (cdna_hip_programming.md §5.7): completion is counted by
// hand with W4_WAIT_VM, and no s_waitcnt vmcnt(0) appears in front of the LDS reads (with the builtin form hipcc put one in front of
// every ds_read_b64_tr_b16 of the TN kernel: 3x slower).
#ifdef VDK_EMU
typedef emu_buffer_rsrc w4_rsrc_t;
__device__ __forceinline__ w4_rsrc_t w4_make_rsrc(const void* base, unsigned bytes) { return emu_buffer_rsrc{(const unsigned char*)base, bytes}; }
#define W4_DMA16(rs, voff, dst, soff) emu_buffer_load_lds((rs), (void*)(dst), 16, (voff), (soff), 0, 0)
#else
typedef u32x4 w4_rsrc_t;
__device__ __forceinline__ w4_rsrc_t w4_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  w4_rsrc_t r = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
  return r;
}
#define W4_DMA16(rs, voff, dst, soff)                                                                                                 \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"                                              \
               ::"s"((unsigned)(unsigned long long)VDK_LDS_PTR(dst)), "v"(voff), "s"(rs), "s"(soff) : "memory")
#endif

// piece j (0..3) of wave w fills LDS bytes [(4w + j) * 1024, +1024) of a region.
//   NT (operand [rows, K] row-major): the piece is region rows 32w + 8j .. +7 (lane: row i = lane >> 3, 16-B position cp = lane & 7, which holds global
//      chunk cp ^ (4 (j & 1) + (i >> 1)) = cp ^ ((row >> 1) & 7)); region row r is operand row (r >> 6) * 128 + sub * 64 + (r & 63).
//   TN (operand [K, cols] row-major): the piece is k-rows 4 (4w + j) .. +3 of the region's [64 k][128 cols] image (lane: k-row lane >> 4, position
//      cp = lane & 15 holding chunk c = cp ^ (4 (k-row & 3)), i.e. operand columns (c >> 3) * 128 + sub * 64 + (c & 7) * 8 .. +7).
struct W4Dma {
  w4_rsrc_t rs;
  unsigned voff0, voff1;     // lane byte offsets of even / odd pieces (TN: equal)
  unsigned sub_stride;       // scalar: bytes from the sub 0 region's source to the sub 1 region's
  unsigned piece_stride;     // scalar: bytes between consecutive pieces
  unsigned kstep;            // scalar: bytes per k-tile
  unsigned kbeg;             // scalar: byte offset of a tile's first k-tile
  unsigned wave_off;         // scalar: this wave's share inside a tile, bytes
  unsigned ldb;              // scalar: operand row pitch, bytes
  // conv != 0 (TN B operand of the implicit weight gradient, VdkConvGeom.rows): the k-major im2col operand [rows, KH*KW*Cin] is GATHERED from the NHWC input -- the source of
  // a lane's 16 bytes is ((b H + oy stride + ky - pad) W + ox stride + kx - pad) Cin + ci for k-row (b, oy, ox) and column (ky, kx, ci), or nothing (zeros) in the padding.
  // The lane's column part (cy / cx / cib per region half) is fixed per output tile; its k-row changes with every piece: two divisions by constants per piece and lane, in
  // the shadow of the MFMAs.  `so0` of w4_piece is then the k-tile's first k-row (kstep = 64 rows, tile base 0), >= 2^31 once the cursor has run off the end.
  int conv;
  int cH, cW, cOW, cHW, cCin, cKW, cstride, cpad, crows, cN;
  float inv_hw, inv_w;
  unsigned w16, kr;          // 16 * wave, the lane's k-row inside a piece
  unsigned ncol0;            // the lane's first column inside a 256-column tile for region half 0 (half 1: + 64)
  int cy[2], cx[2]; unsigned cib[2]; int cok[2];
};
// n / d and n % d for 0 <= n < 2^24, d < 2^24: float quotient (exactly representable operands), corrected by at most one either way
__device__ __forceinline__ void w4_divmod(unsigned n, int d, float inv, int& q, int& r) {
  q = (int)((float)n * inv);
  r = (int)n - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
}
// the lane's column part for the output tile whose first column is n0
__device__ __forceinline__ void w4_conv_set_tile(W4Dma& d, int n0) {
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int n = n0 + (int)d.ncol0 + sub * 64;
    const int tap = n / d.cCin, ci = n - tap * d.cCin, ky = tap / d.cKW, kx = tap - ky * d.cKW;
    d.cok[sub] = n < d.cN;
    d.cy[sub] = ky - d.cpad; d.cx[sub] = kx - d.cpad; d.cib[sub] = (unsigned)ci * 2u;
  }
}

template <bool TN>
__device__ __forceinline__ void w4_dma_init(W4Dma& d, const bf16_t* base, long ld, int extent /* operand rows (NT) */, int K, int kbeg, int w, int lane,
                                            unsigned blk = 128u /* operand rows (NT) / columns (TN) between the two halves of a region: 128, or 64 for the 128-wide B of gemm_w4h_kernel */) {
  const unsigned ldb = (unsigned)ld * 2u;
  d.ldb = ldb;
  d.conv = 0;
  if (!TN) {
    d.rs = w4_make_rsrc(base, (unsigned)extent * ldb);
    const unsigned i = (unsigned)lane >> 3, cp = (unsigned)lane & 7u;
    d.voff0 = i * ldb + ((cp ^ (i >> 1)) << 4);
    d.voff1 = i * ldb + ((cp ^ (4u + (i >> 1))) << 4);
    d.wave_off = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(w >> 1) * blk + (unsigned)(w & 1) * 32u) * ldb));
    d.sub_stride = 64u * ldb;
    d.piece_stride = 8u * ldb;
    d.kbeg = (unsigned)kbeg * 2u;
    d.kstep = 128u;
  } else {
    d.rs = w4_make_rsrc(base, (unsigned)K * ldb);
    const unsigned kr = (unsigned)lane >> 4, cp = (unsigned)lane & 15u, c = cp ^ (4u * (kr & 3u));
    d.voff0 = d.voff1 = kr * ldb + ((c >> 3) * blk + (c & 7u) * 8u) * 2u;
    d.wave_off = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)w * 16u * ldb));
    d.sub_stride = 128u;
    d.piece_stride = 4u * ldb;
    d.kbeg = (unsigned)kbeg * ldb;
    d.kstep = 64u * ldb;
  }
}
// scalar offset of a tile's origin (first operand row (NT) / column (TN)) for this wave
template <bool TN>
__device__ __forceinline__ unsigned w4_tile_base(const W4Dma& d, int origin) {
  return (TN ? (unsigned)origin * 2u : (unsigned)origin * d.ldb) + d.wave_off;
}
// piece j of region `sub` of the k-tile at scalar offset so0 (tile base + k offset)
__device__ __forceinline__ void w4_piece(const W4Dma& d, unsigned char* dst, int sub, int j, unsigned so0) {
  if (d.conv) {      // (wave-uniform) gathered im2col operand: see W4Dma
    const unsigned rr = so0 + d.w16 + 4u * (unsigned)j + d.kr;
    unsigned voff = W4_OOB;
    if (rr < (unsigned)d.crows && d.cok[sub]) {
      int b, rem, oy, ox;
      w4_divmod(rr, d.cHW, d.inv_hw, b, rem);
      w4_divmod((unsigned)rem, d.cOW, d.inv_w, oy, ox);
      const int iy = oy * d.cstride + d.cy[sub], ix = ox * d.cstride + d.cx[sub];
      if ((unsigned)iy < (unsigned)d.cH && (unsigned)ix < (unsigned)d.cW) voff = (unsigned)(((b * d.cH + iy) * d.cW + ix) * d.cCin) * 2u + d.cib[sub];
    }
    W4_DMA16(d.rs, voff, dst, 0u);
    return;
  }
  const unsigned so = so0 + (unsigned)sub * d.sub_stride + (unsigned)j * d.piece_stride;
  W4_DMA16(d.rs, (j & 1) ? d.voff1 : d.voff0, dst, so);
}
__device__ __forceinline__ void w4_region(const W4Dma& d, unsigned char* region, int sub, int w, unsigned so0) {
#pragma unroll
  for (int j = 0; j < 4; ++j) w4_piece(d, region + (4 * w + j) * 1024, sub, j, so0);
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
// c[i][j]: i = A block (rows of C), j = B block (columns of C); the MFMA is (B fragment, A fragment): D rows = C columns, so lane l holds C row l & 31.
// RD: the phase reads 8 fragments from `rd_region` through offsets rd_off into RDST[2][4].  It issues the 4 pieces of (dma, is_region, is_sub) at scalar offset so0.
// FIRST: the accumulators start from zero (first k-tile of an output tile).
template <bool TN, bool RD, bool FIRST, int OF>
__device__ __forceinline__ void w4_phase(f32x16 (&c00), f32x16 (&c01), f32x16 (&c10), f32x16 (&c11), const s16x8 (&X)[2][4], const s16x8 (&Y)[2][4],
                                         const unsigned char* rd_region, const unsigned* rd_off, s16x8 (&RDST)[2][4],
                                         const W4Dma& dma, unsigned char* is_region, int is_sub, unsigned so0, int w) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    c00 = vdk_mfma32<OF>(Y[0][ks], X[0][ks], (FIRST && ks == 0) ? zero : c00);
    if (RD) RDST[0][ks] = w4_frag<TN>(rd_region, rd_off, 0, ks);
    c01 = vdk_mfma32<OF>(Y[1][ks], X[0][ks], (FIRST && ks == 0) ? zero : c01);
    if (RD) RDST[1][ks] = w4_frag<TN>(rd_region, rd_off, 1, ks);
    c10 = vdk_mfma32<OF>(Y[0][ks], X[1][ks], (FIRST && ks == 0) ? zero : c10);
    w4_piece(dma, is_region + (4 * w + ks) * 1024, is_sub, ks, so0);
    c11 = vdk_mfma32<OF>(Y[1][ks], X[1][ks], (FIRST && ks == 0) ? zero : c11);
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
    __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  }
#endif
}

// ---- one k-tile -----------------------------------------------------------------------------------------------------------------------
// CUR: buffer of this k-tile.  VM: vmcnt that must hold before the barrier of the next phase (-1: no wait needed, the first k-tile after a deep wait); it is
// waited for at the END of a phase, so it never sits between a phase's reads and its MFMAs.  Every k-tile refills its own buffer with the stream's k-tile
// t+2 (scalar offsets soa / sob: wherever the DMA cursor stands, possibly the next output tile or W4_OOB).  NEXT: phases 3 / 4 read the first fragments (A0, B0)
// of k-tile t+1 (not in an output tile's last k-tile: the epilogue wants the registers, the next tile reads them afterwards).  B0 holds this tile's B0
// fragments, B0N receives the next tile's.
template <int V>
__device__ __forceinline__ void w4_phase_end() {
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (V >= 0) W4_WAIT_VM(V);
  W4_WAIT_LGKM0();          // this phase's fragment reads have returned: the next phase may multiply them, and (after its barrier) anyone may overwrite their region
}
template <bool TN, int CUR, int VM, bool FIRST, bool NEXT, int OF>
__device__ __forceinline__ void w4_ktile(unsigned char* smem, const W4Frag<TN>& F, const W4Dma& da, const W4Dma& db, unsigned soa, unsigned sob, int w, f32x16 (&acc)[4][4],
                                         s16x8 (&A0)[2][4], s16x8 (&A1)[2][4], s16x8 (&B0)[2][4], s16x8 (&B1)[2][4], s16x8 (&B0N)[2][4]) {
  unsigned char* const buf = smem + CUR * W4_TILEBUF;
  unsigned char* const nbuf = smem + (CUR ^ 1) * W4_TILEBUF;
  // (round 6: TWO barriers per k-tile -- in front of P1 and P3, vmcnt(16) at the ends of P2 and P4, none at the ends of P1 and P3: valid, bit-equal, and 0.3 ms per
  //  ViT-B/16 step SLOWER, 34.67 -> 35.0: the four barriers keep the four SIMDs' read bursts and DMA issues in step.  Four stay.)
  // P1: A0 x B0; read B1(t); refill RA0 (read in P3 of the previous k-tile)
  W4_BAR();
  w4_phase<TN, true, FIRST, OF>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], A0, B0, buf + W4_RB1, F.b, B1, da, buf + W4_RA0, 0, soa, w);
  w4_phase_end<VM>();
  // P2: A0 x B1; read A1(t); refill RB0 (read in P4 of the previous k-tile)
  W4_BAR();
  w4_phase<TN, true, FIRST, OF>(acc[0][2], acc[0][3], acc[1][2], acc[1][3], A0, B1, buf + W4_RA1, F.a, A1, db, buf + W4_RB0, 0, sob, w);
  w4_phase_end<VM>();
  // P3: A1 x B1; read A0(t+1); refill RB1 (read in P1)
  W4_BAR();
  w4_phase<TN, NEXT, FIRST, OF>(acc[2][2], acc[2][3], acc[3][2], acc[3][3], A1, B1, nbuf + W4_RA0, F.a, A0, db, buf + W4_RB1, 1, sob, w);
  w4_phase_end<VM>();
  // P4: A1 x B0; read B0(t+1); refill RA1 (read in P2)
  W4_BAR();
  w4_phase<TN, NEXT, FIRST, OF>(acc[2][0], acc[2][1], acc[3][0], acc[3][1], A1, B0, nbuf + W4_RB0, F.b, B0N, da, buf + W4_RA1, 1, soa, w);
  w4_phase_end<VM>();
}

// ---- epilogue -------------------------------------------------------------------------------------------------------------------------
// Accumulator layout: acc[rt][ct][4 g + e] = C[m0 + wr*128 + rt*32 + l31][n0 + wc*128 + ct*32 + 8 g + 4 hi + e].
// Fast forms (bf16 output: plain, bias, bias + GELU with the saved pre-activation, dGELU, either with the column sums of the stored output): everything is computed
// in that layout and only bf16 rows go through the wave's 8 KB staging ([32 rows][128 bf16], 16-byte chunk c of row r at position c ^ (r & 15): the 8-byte
// writes of 16 rows and the 16-byte reads of a row pair both hit every bank once); a row leaves as one 256-byte segment.
// Everything else (fp32 outputs, residuals, split-K slabs, margin heads, run-time flags): 32 x 64 fp32 slabs through the same staging and vdk_gemm_epilogue.h.
#ifdef VDK_EMU
#define W4_EPI_SYNC() VDK_WAVE_LDS_SYNC()
#else
#define W4_EPI_SYNC() do { __builtin_amdgcn_s_waitcnt(0xC07F); VDK_WAVE_LDS_SYNC(); } while (0)   /* EXPERIMENT: drain the LDS queue at every staging hand-off */
#endif
// NCT: 32-column blocks per wave (4: the 256x256 tile of gemm_w4_kernel, staging rows of 256 B, 4 rows per 64-lane pass; 2: the 256x128 tile of gemm_w4h_kernel,
// staging rows of 128 B with chunk c of row r at position c ^ ((r >> 1) & 7), 8 rows per pass).  DEEP: vmcnt waited for before the first global access (-1: none).
template <int E, int NCT, int DEEP, int OF>
__device__ __forceinline__ void w4_epilogue(const GemmParams& p, unsigned char* stage, f32x16 (&acc)[4][NCT], int lane, int wr, int wc, int m0, int n0, int z, int tm, unsigned long long (&ts)[5],
                                            const u32x4* auxpre = nullptr /* E_DGELU: the saved operand's rows of the first 32-row block, requested by the caller (w4_aux_prefetch) */) {
  constexpr int NCH = NCT * 4;            // 16-byte chunks per staged row
  constexpr int RPP = 64 / NCH;           // rows per 64-lane pass
  constexpr int NPS = 32 / RPP;           // passes per 32-row block
  constexpr int ROWB = NCT * 64;          // staged row, bytes
  constexpr int EB = E & ~E_AUXD;                             // (E_AUXD only changes what aux holds)
  constexpr bool FAST = (E == 0 || E == E_BIAS || EB == (E_BIAS | E_GELU) || EB == E_DGELU || EB == (E_DGELU | E_OCS) || E == E_OCS);
  const int l31 = lane & 31, hi = lane >> 5;
  const int mrow0 = m0 + wr * 128, ncol0 = n0 + wc * (NCT * 32);
  const int swz_w = NCT == 4 ? (l31 & 15) : ((l31 >> 1) & 7);   // staging swizzle of this lane's row
  if constexpr (FAST) {
    const int rrow = lane / NCH, rc = lane % NCH;               // row-pass coordinates: rows pass * RPP + rrow, 16-byte chunk rc (8 columns)
    const int ncol = ncol0 + rc * 8;
    const bool nok = ncol < p.N;
    float bias[NCT][4][4];
    if (E & E_BIAS) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol0 + ct * 32 + 8 * g + 4 * hi;
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (n < p.N) b = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) bias[ct][g][e] = b[e];
        }
    }
    float ocs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ocs[e] = 0.f;
    unsigned char* const wbase = stage + l31 * ROWB + hi * 8;   // + ((ct * 4 + g) ^ swz_w) * 16
    // Rows leave (and the dGELU operand arrives) through buffer descriptors: the lane part of the offset is loop-invariant, the row block is a scalar, rows
    // beyond M fall outside num_records and columns beyond N get an out-of-range lane offset, so there is neither 64-bit address arithmetic nor a branch.
    const unsigned lane_out = nok ? (unsigned)rrow * (unsigned)p.ldc * 2u + (unsigned)ncol * 2u : W4_OOB;
    const unsigned lane_aux = nok ? (unsigned)rrow * (unsigned)p.ldaux * 2u + (unsigned)ncol * 2u : W4_OOB;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((unsigned)p.M * (unsigned)p.ldc * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc((E & (E_GELU | E_DGELU)) ? (void*)p.aux : p.C, 0,
                                                                            (int)((unsigned)p.M * (unsigned)((E & (E_GELU | E_DGELU)) ? p.ldaux : p.ldc) * 2u), 0x00020000);
    // rows of the staged 32 x 128 block -> the tensor behind rs (row pitch ldbytes); OCS: what is stored also goes into the column sums
    auto rows_out = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned lane_off, unsigned ldbytes, int rt, bool with_ocs) {
      W4_EPI_SYNC();
      const unsigned srow = (unsigned)(mrow0 + rt * 32) * ldbytes;
      u32x4 d[NPS];
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int row = ps * RPP + rrow;
        d[ps] = *(const u32x4*)(stage + row * ROWB + ((rc ^ (NCT == 4 ? (row & 15) : ((row >> 1) & 7))) << 4));
      }
      if (with_ocs) {
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
          for (int e = 0; e < 4; ++e) { ocs[2 * e] += op_lo<OF>(d[ps][e]); ocs[2 * e + 1] += op_hi<OF>(d[ps][e]); }
      }
      // Everything that reads the row registers comes BEFORE the stores, and the row block goes into the VECTOR offset.  Measured on the MI355X with
      // buffer_store_dwordx4 v[a:a+3], v, s[..], s offen followed within four instructions by VALU writes of v[a:a+3] (the column-sum arithmetic): dword 1 of
      // lanes 12-15 / 28-31 / 44-47 / 60-63 reached memory already overwritten.  The compiler pads the wide-store data hazard only for the form without a
      // scalar-offset register.
      // Non-temporal stores (cache policy nt): what these epilogues write -- activations and gradients of [tokens, features] size, the saved GELU operand -- is either
      // re-read only after far more than the Infinity Cache's 256 MB has streamed by, or larger than it on its own; written with the default policy it evicts the operand
      // panels and weights the GEMMs DO re-read.  Round 6, same-box A/B of the ViT-B/16 step: 35.85 -> 35.12 ms (the saved operand alone: 35.4); swin_base 29.5 -> 28.8 ms, cfg3 108.0 -> 106.4 ms.
      // (A run-time choice per launch -- a wave-uniform branch around the stores -- cost 1.4 ms of the step in spills: the policy is a compile-time constant.
      //  The same policy measured null, +-0.1 ms, on every other stream of the step: the fp32 residual rows in and out, the saved-operand loads of the dGELU forms,
      //  LayerNorm backward's saved input, the attention backward's dq / dk / dv rows, the split-K slab reduce, the optimizer pass; non-temporal operand DMAs of the
      //  weight-gradient GEMMs -- every operand byte is read by one XCD once -- cost 0.25 ms: 34.8 -> 35.05.)
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
        __builtin_amdgcn_raw_buffer_store_b128(d[ps], rs, lane_off + (srow + (unsigned)(ps * RPP) * ldbytes), 0, W4_STORE_POLICY);
      W4_EPI_SYNC();
    };
    if constexpr (DEEP >= 0) W4_WAIT_VM(DEEP);                  // deep wait: every DMA group but the newest has landed (the next tile's first k-tile then runs without waits)
    if (p.dbg) ts[0] = __builtin_readcyclecounter();
    u32x4 auxrows[NPS];
    auto aux_request = [&](int rt_) {
      const unsigned srow = (unsigned)(mrow0 + rt_ * 32) * (unsigned)p.ldaux * 2u;
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
        auxrows[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_aux, lane_aux, srow + (unsigned)(ps * RPP) * (unsigned)p.ldaux * 2u, 0);   // (out of range: zeros)
    };
    if (E & E_DGELU) {
      if (auxpre) {
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) auxrows[ps] = auxpre[ps];
      } else aux_request(0);
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      if (p.dbg && rt > 0) ts[rt] = __builtin_readcyclecounter();
#ifndef VDK_EMU
      __builtin_amdgcn_sched_barrier(0);                        // one 32-row block at a time: blocks interleaved by the scheduler keep several of them in registers
#endif
      u32x2 held[NCT][4];                                         // GELU: the activated values wait here (packed) while the pre-activation rows leave
      if (E & E_DGELU) {
        if (!W4_EPI_AHEAD && rt > 0) aux_request(rt);
        // the saved operand of these 32 rows (requested one block ahead): whole row segments -> staging -> this lane's layout
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int row = ps * RPP + rrow;
          *(u32x4*)(stage + row * ROWB + ((rc ^ (NCT == 4 ? (row & 15) : ((row >> 1) & 7))) << 4)) = auxrows[ps];
        }
        if (W4_EPI_AHEAD && rt < 3) aux_request(rt + 1);          // travels while this block is multiplied, staged back and stored (a lone wave has nothing else to hide it behind)
        W4_EPI_SYNC();
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          unsigned char* const wp = wbase + (((ct * 4 + g) ^ swz_w) << 4);
          float v[4];
          if (E & E_DGELU) {
            const u32x2 u = *(const u32x2*)wp;                  // (the same 8 bytes this lane overwrites below: u goes out, dL/du comes in)
            if (E & E_AUXD) {                                   // aux holds GELU'(u) already
              v[0] = acc[rt][ct][4 * g + 0] * h_lo(u[0]); v[1] = acc[rt][ct][4 * g + 1] * h_hi(u[0]);
              v[2] = acc[rt][ct][4 * g + 2] * h_lo(u[1]); v[3] = acc[rt][ct][4 * g + 3] * h_hi(u[1]);
            } else {
              const vdk_f32x2 d0 = gelu_grad_f2((vdk_f32x2){op_lo<OF>(u[0]), op_hi<OF>(u[0])}), d1 = gelu_grad_f2((vdk_f32x2){op_lo<OF>(u[1]), op_hi<OF>(u[1])});
              v[0] = acc[rt][ct][4 * g + 0] * d0[0]; v[1] = acc[rt][ct][4 * g + 1] * d0[1];
              v[2] = acc[rt][ct][4 * g + 2] * d1[0]; v[3] = acc[rt][ct][4 * g + 3] * d1[1];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (E & E_BIAS) ? acc[rt][ct][4 * g + e] + bias[ct][g][e] : acc[rt][ct][4 * g + e];
          }
          if ((E & E_GELU) && (E & E_AUXD)) {                   // GELU and its derivative from one erf / exp: the derivative rows leave through aux
            vdk_f32x2 g0, g1, d0, d1;
            gelu_both_f2((vdk_f32x2){v[0], v[1]}, g0, d0); gelu_both_f2((vdk_f32x2){v[2], v[3]}, g1, d1);
            *(u32x2*)wp = (u32x2){pack_h2(d0[0], d0[1]), pack_h2(d1[0], d1[1])};
            held[ct][g] = (u32x2){pack_op2<OF>(g0[0], g0[1]), pack_op2<OF>(g1[0], g1[1])};
          } else
          *(u32x2*)wp = (u32x2){pack_op2<OF>(v[0], v[1]), pack_op2<OF>(v[2], v[3])};
          // (the activations stay in hipcc's order, value pair by value pair on packed fp32 ops: a lone wave issues a VALU instruction every ~8 cycles whatever
          //  the dependencies, so instruction COUNT is the cost -- a stage-by-stage 8-wide form without packed ops measured 10-25 % slower)
          if ((E & E_GELU) && !(E & E_AUXD)) {
            const vdk_f32x2 g0 = gelu_f2((vdk_f32x2){v[0], v[1]}), g1 = gelu_f2((vdk_f32x2){v[2], v[3]});
            held[ct][g] = (u32x2){pack_op2<OF>(g0[0], g0[1]), pack_op2<OF>(g1[0], g1[1])};
          }
#ifndef VDK_EMU
          if ((E & (E_GELU | E_DGELU)) && g == 3) __builtin_amdgcn_sched_barrier(0);   // 16 activations in flight are plenty; all 64 at once spill
#endif
        }
      if (E & E_GELU) {
        rows_out(rs_aux, lane_aux, (unsigned)p.ldaux * 2u, rt, false);   // the pre-activation, for the backward pass
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int g = 0; g < 4; ++g) *(u32x2*)(wbase + (((ct * 4 + g) ^ swz_w) << 4)) = held[ct][g];
      }
      rows_out(rs_out, lane_out, (unsigned)p.ldc * 2u, rt, (E & E_OCS) != 0);
    }
    if (E & E_OCS) {   // lanes with the same chunk rc hold the same 8 columns: sum over the 4 row slots, one partial row per (row tile, wave row)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = ocs[e];
        if (NCT == 2) s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        ocs[e] = s;
      }
      if (lane < NCH && nok) {
        float* dst = p.ocs_part + ((long)tm * 2 + wr) * p.N + ncol;
        *(f32x4*)dst = (f32x4){ocs[0], ocs[1], ocs[2], ocs[3]};
        *(f32x4*)(dst + 4) = (f32x4){ocs[4], ocs[5], ocs[6], ocs[7]};
      }
    }
  } else if constexpr (E == (E_BIAS | E_RES | E_F32) || E == (E_BIAS | E_F32)) {
    constexpr bool RES = (E & E_RES) != 0;                      // (E_BIAS | E_F32: the same row pass without the residual rows -- ConvNeXt's downsample / stem outputs)
    // fp32 output + fp32 residual (proj / fc2 into the residual stream): 32 x 64 fp32 blocks of the accumulators go through the staging
    // (16-byte chunk c of row r at position c ^ (r & 15)), and a row pass (4 rows x 256 B per instruction) adds the residual rows and stores -- buffer
    // descriptors again: no 64-bit address arithmetic, no branch, rows beyond M and columns beyond N fall out of range.
    float* const slab = (float*)stage;
    const int rrow = lane >> 4, rc = lane & 15;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((unsigned)p.M * (unsigned)p.ldc * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(RES ? (void*)p.residual : p.C, 0, (int)((unsigned)p.M * (unsigned)(RES ? p.ldr : p.ldc) * 4u), 0x00020000);
    if constexpr (DEEP >= 0) W4_WAIT_VM(DEEP);
    if (p.dbg) ts[0] = __builtin_readcyclecounter();
    // Bias is added in the row pass (this lane: 4 columns of every row, one f32x4 per column pair) in the order (acc + bias) + residual of the other kernels; the
    // residual rows of a 32 x 64 block are requested ONE BLOCK AHEAD: a lone wave has nothing to hide the ~2 us of an HBM read behind, and 8 blocks per tile waited for it.
    constexpr int NCP = NCT / 2, NB = NCP * 4;
    unsigned lane_out[NCP], lane_res[NCP];
    f32x4 bias4[NCP], cs4[NCP];
    const bool csc = p.cscale != nullptr;                       // VdkGemmDesc.col_scale: acc * col_scale before bias / residual (wave-uniform; absent: not a single extra operation)
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      const int ncol = ncol0 + cp * 64 + rc * 4;
      const bool nok = ncol < p.N;
      lane_out[cp] = nok ? (unsigned)rrow * (unsigned)p.ldc * 4u + (unsigned)ncol * 4u : W4_OOB;
      lane_res[cp] = (RES && nok) ? (unsigned)rrow * (unsigned)p.ldr * 4u + (unsigned)ncol * 4u : W4_OOB;
      bias4[cp] = (f32x4){0.f, 0.f, 0.f, 0.f};
      cs4[cp] = (f32x4){1.f, 1.f, 1.f, 1.f};
      if (nok) bias4[cp] = *(const f32x4*)(p.bias + ncol);
      if (nok && csc) cs4[cp] = *(const f32x4*)(p.cscale + ncol);
    }
    // (round 6, measured and dropped: two / three blocks of residual rows in flight instead of one -- proj + residual 97.0 -> 108 / 127 us, fc2 246 -> 251 / 277: the 32 / 64
    //  extra registers do not exist beside the 256 accumulators and what the persistent walk keeps across the epilogue: 19 / 28 spilled registers)
    u32x4 rn[8];
    auto res_request = [&](int b) {
      const int cp_ = b >> 2, rt_ = b & 3;
      const unsigned srow_r = (unsigned)(mrow0 + rt_ * 32) * (unsigned)p.ldr * 4u;
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) rn[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, lane_res[cp_], srow_r + (unsigned)(ps * 4) * (unsigned)p.ldr * 4u, 0);
    };
    if (RES) res_request(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int cp = b >> 2, rt = b & 3;
      const unsigned srow_o = (unsigned)(mrow0 + rt * 32) * (unsigned)p.ldc * 4u;
      if (RES && !W4_EPI_AHEAD && b > 0) res_request(b);
      f32x4 r[8];
#pragma unroll
      for (int ps = 0; ps < 8; ++ps)
        r[ps] = RES ? (f32x4){__uint_as_float(rn[ps][0]), __uint_as_float(rn[ps][1]), __uint_as_float(rn[ps][2]), __uint_as_float(rn[ps][3])} : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (RES && W4_EPI_AHEAD && b + 1 < NB) res_request(b + 1);
#pragma unroll
      for (int ctl = 0; ctl < 2; ++ctl)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x16& a = acc[rt][cp * 2 + ctl];
          *(f32x4*)(slab + l31 * 64 + (((ctl * 8 + 2 * g + hi) ^ (l31 & 15)) << 2)) = (f32x4){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        }
      W4_EPI_SYNC();
      f32x4 d[8];
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 4 + rrow;
        d[ps] = *(const f32x4*)(slab + row * 64 + ((rc ^ (row & 15)) << 2));
      }
      if (csc) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) d[ps] = d[ps] * cs4[cp];
      }
      if (RES && p.rscale) {      // VdkGemmDesc.row_scale (stochastic depth): residual + factor(sample of this row) * (acc + bias); rows beyond M read the last entry (never stored)
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          int row = mrow0 + rt * 32 + ps * 4 + rrow;
          row = row < p.M ? row : p.M - 1;
          const float f = p.rscale[row / p.rps];
          d[ps] = (d[ps] + bias4[cp]) * f + r[ps];
        }
      } else
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) d[ps] = RES ? (d[ps] + bias4[cp]) + r[ps] : d[ps] + bias4[cp];
      // (everything that touches the row registers precedes the stores; row block in the vector offset: see the bf16 form above)
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const u32x4 o = {__float_as_uint(d[ps][0]), __float_as_uint(d[ps][1]), __float_as_uint(d[ps][2]), __float_as_uint(d[ps][3])};
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, lane_out[cp] + (srow_o + (unsigned)(ps * 4) * (unsigned)p.ldc * 4u), 0, 0);
      }
      W4_EPI_SYNC();
    }
  } else {
    float* const slab = (float*)stage;
    float q8_unused = 0.f;
    if constexpr (DEEP >= 0) W4_WAIT_VM(DEEP);
    if (p.dbg) ts[0] = __builtin_readcyclecounter();
#pragma unroll
    for (int cp = 0; cp < NCT / 2; ++cp) {
      const int ncol = ncol0 + cp * 64 + (lane & 7) * 8;
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
        for (int ctl = 0; ctl < 2; ++ctl)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16& a = acc[rt][cp * 2 + ctl];
            const f32x4 x = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
            *(f32x4*)(slab + l31 * 64 + (((ctl * 8 + 2 * g + hi) ^ (l31 & 15)) << 2)) = x;
          }
        W4_EPI_SYNC();
        const long mbase = (long)mrow0 + rt * 32;
        if (E != E_GENERIC) {
          h_epilogue_half<E, 4, true, OF>(p, slab, lane, mbase, ncol, z, bias8, ocs8, q8_unused);
        } else {
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), ch = (lane & 7) * 2, sw = row & 15;
            const long mi = mbase + row;
            if (mi < p.M && ncol < p.N) {
              float v[8];
              f32x4 x0 = *(const f32x4*)(slab + row * 64 + ((ch ^ sw) << 2)), x1 = *(const f32x4*)(slab + row * 64 + (((ch + 1) ^ sw) << 2));
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
              g_epilogue_store8<OF>(p, mi, ncol, v, z);
            }
          }
        }
        W4_EPI_SYNC();
      }
    }
  }
}

// The dGELU epilogues multiply by a saved operand (GELU'(u) or u) that arrives as whole row segments, one 32-row block ahead of its use -- except the FIRST block of a
// tile, whose rows are requested when the epilogue begins: a lone wave then sits through one memory round trip per tile (cycle stamps, dfc2 of ViT-B/16: first block 8.1 k
// cycles, the others 3.1 k).  W4_AUX_PREFETCH = 1 requests that first block in front of the tile's LAST k-tile, with the lane / row mapping of w4_epilogue's fast form
// (NCT = 4).  Measured (tools/r6_gemm_quick.py, same box, two rounds): 322.6 / 319.9 us without, 324.0 / 321.8 us with -- the 32 registers it holds across the last
// k-tile cost what the round trip saved (11 spilled registers instead of 6): off.
__device__ __forceinline__ void w4_aux_prefetch(const GemmParams& p, u32x4 (&pre)[8], int lane, int wr, int wc, int m0, int n0) {
  const int rrow = lane >> 4, rc = lane & 15;
  const int mrow0 = m0 + wr * 128, ncol = n0 + wc * 128 + rc * 8;
  const unsigned lane_aux = ncol < p.N ? (unsigned)rrow * (unsigned)p.ldaux * 2u + (unsigned)ncol * 2u : W4_OOB;
  const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc((void*)p.aux, 0, (int)((unsigned)p.M * (unsigned)p.ldaux * 2u), 0x00020000);
  const unsigned srow = (unsigned)mrow0 * (unsigned)p.ldaux * 2u;
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) pre[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_aux, lane_aux, srow + (unsigned)(ps * 4) * (unsigned)p.ldaux * 2u, 0);
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------------------
// Linear tile id -> (tile row, tile column).  An XCD works through a contiguous range of ids; in plain row-major order that range sweeps ALL tile columns every few
// rows, i.e. the whole B operand (the weight matrix) again and again: at 4.7 MB (768 x 3072) it does not stay in the XCD's 4 MB L2 and streamed back from the Infinity
// Cache once per round (PMC: the fc1 GEMM fetched 469 MB for 82 MB of operands, the dGELU GEMM 1134 MB for 392 MB).  With the ids ordered by COLUMN BANDS of `cw`
// tile columns (row-major inside a band) an XCD stays inside one band whose slice of B fits its L2; the price is that every A row panel is read once per band.
__device__ __forceinline__ void w4_tile_rc(int tile, int ntn, int ntm, int cw, int& tm, int& tn) {
  if (cw <= 0 || cw >= ntn) { tm = tile / ntn; tn = tile - tm * ntn; return; }
  const int per = ntm * cw, band = tile / per, rem = tile - band * per;
  const int wdt = ntn - band * cw < cw ? ntn - band * cw : cw;          // (the last band may be narrower)
  tm = rem / wdt; tn = band * cw + rem - tm * wdt;
}
// PERSIST: gridDim.x workgroups (a multiple of 8, at most one per CU) walk the output tiles; workgroup b serves the tiles start(x) + s, + s + G/8, ... of its
// XCD's contiguous share (x = b & 7, s = b >> 3): at any time an XCD's workgroups multiply G/8 consecutive tiles, which share A row panels / the weight matrix
// in that XCD's L2.  Otherwise: one tile per workgroup (blockIdx.x, same XCD raster) and blockIdx.y = split-K slice.
// Every k-range holds an even number (>= 2) of whole k-tiles (launcher).
template <bool TN, int E, bool PERSIST, int OF>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[W4_SMEM];   // 128 KB operand ring + 32 KB epilogue staging: the CU's whole LDS, one object
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  const int nwg = ntn * ntm;
  // this workgroup's tiles: start + slot, + slot + stride, ... while < start + cnt
  int z = PERSIST ? 0 : (int)blockIdx.y;
  int t_start, t_cnt, t_stride, t_idx;
  if (!PERSIST && p.sk_xcd) {
    // split-K on a 1-D grid: the (split, tile) items in split-major order, XCD x (= workgroup id & 7, the dispatcher's round robin) takes a CONTIGUOUS share of them.
    // With tiles x splits ~ CUs an XCD's 32 workgroups then hold (most of) ONE k-slab: they walk the same rows in step and every operand panel of the slab reaches
    // that XCD's L2 once.  On the (tiles, splits) grid every XCD held ~4 tiles of EVERY slab: the ViT weight-gradient GEMMs fetched 2.8x their operands
    // (869 MB per launch against 310 MB, PMC) and ran at the fabric's 6.4 TB/s rather than at the MFMA rate.
    const int items = nwg * p.splitk, bid = blockIdx.x, q = items >> 3, r = items & 7, xcd = bid & 7;
    const int item = xcd * q + (xcd < r ? xcd : r) + (bid >> 3);
    if ((bid >> 3) >= q + (xcd < r ? 1 : 0)) return;      // (uniform: the whole workgroup leaves before any barrier)
    z = item / nwg;
    t_start = item - z * nwg; t_cnt = 1; t_idx = 0; t_stride = 1 << 30;
  } else {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t_start = xcd * q + (xcd < r ? xcd : r);
    t_cnt = q + (xcd < r ? 1 : 0);
    t_idx = bid >> 3;
    t_stride = PERSIST ? (int)(gridDim.x >> 3) : (1 << 30);
  }
  const int kbeg = z * p.k_per_split;
  int kend = kbeg + p.k_per_split; if (kend > p.K) kend = p.K;
  const int nk = (kend - kbeg) / 64;
  if (t_idx >= t_cnt || nk < 2) return;                   // (uniform: the whole workgroup leaves before any barrier)
#ifndef VDK_EMU
  // Start phase of the persistent walk (p.stagger shader cycles ~ one tile period; 0: off).  Equal tiles keep all CUs in lock-step: every workgroup multiplies, then every
  // workgroup runs its epilogue, so the HBM traffic of the epilogues (fp32 residual in / out: 512 KB per tile) comes in chip-wide bursts during which no MFMA runs, and the
  // main loops leave HBM idle.  Workgroups that walk one tile fewer than the busiest ones have a tile period of slack: they start a hashed fraction of it late
  // (stagger_lo = 1: every workgroup is delayed by a hashed fraction of half a period, slack or not -- experiments).
  if (PERSIST && p.stagger > 0) {
    const int max_tcnt = (nwg >> 3) + ((nwg & 7) ? 1 : 0);
    const int my_cnt = (t_cnt - t_idx + t_stride - 1) / t_stride, max_cnt = (max_tcnt + t_stride - 1) / t_stride;
    const unsigned hsh = ((unsigned)blockIdx.x * 0x9E3779B1u) >> 16;                    // uniform in [0, 65536)
    long delay = 0;
    // (round 6, measured null: the hash keyed by the A row panel of the workgroup's first tile instead of the workgroup id -- the workgroups on one panel's column tiles
    //  then stay in step and could share it in L2 while panels are spread: 35.49 against 35.48 ms)
    if (my_cnt < max_cnt) delay = ((long)p.stagger * (long)hsh) >> 16;
    else if (p.stagger_lo == 1) delay = ((long)p.stagger * (long)hsh) >> 17;
    if (delay > 0) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      while ((long)(__builtin_readcyclecounter() - t0) < delay) __builtin_amdgcn_s_sleep(16);
    }
  }
#endif

  W4Dma da, db;
  const bool convb = TN && p.conv_on;      // the implicit weight gradient: B gathered from the NHWC input, A's valid rows end at crows (K is crows rounded up)
  w4_dma_init<TN>(da, p.A, p.lda, p.M, convb ? p.crows : p.K, kbeg, w, lane);
  w4_dma_init<TN>(db, p.B, p.ldb, p.N, p.K, kbeg, w, lane);
  if (convb) {
    db.conv = 1;
    db.cH = p.cH; db.cW = p.cW; db.cOW = p.cOW; db.cHW = p.cOH * p.cOW; db.cCin = p.cCin; db.cKW = p.cKW; db.cstride = p.cstride; db.cpad = p.cpad; db.crows = p.crows; db.cN = p.N;
    db.inv_hw = 1.0f / (float)db.cHW; db.inv_w = 1.0f / (float)db.cOW;
    db.rs = w4_make_rsrc(p.B, (unsigned)((long)(p.crows / db.cHW) * p.cH * p.cW * p.cCin * 2));
    db.w16 = (unsigned)(16 * w);
    db.kr = (unsigned)lane >> 4;
    { const unsigned cp = (unsigned)lane & 15u, c = cp ^ (4u * (db.kr & 3u)); db.ncol0 = (c >> 3) * 128u + (c & 7u) * 8u; }
    db.kbeg = (unsigned)kbeg; db.kstep = 64u; db.wave_off = 0u;
  }
  W4Frag<TN> F;
  w4_frag_init<TN>(F, wr, wc, lane);

  // DMA cursor: the next k-tile of the stream to be issued (two k-tiles ahead of the MFMAs)
  int c_idx = t_idx, c_left = nk;
  unsigned c_ta, c_tb, c_ka = da.kbeg, c_kb = db.kbeg;
  {
    int tm_, tn_; w4_tile_rc(t_start + c_idx, ntn, ntm, p.band_cw, tm_, tn_);
    c_ta = w4_tile_base<TN>(da, tm_ * 256); c_tb = w4_tile_base<TN>(db, tn_ * 256);
    if (convb) { c_tb = 0u; w4_conv_set_tile(db, tn_ * 256); }
  }
#define W4_CURSOR_ADVANCE()                                                                                         \
  do {                                                                                                              \
    c_ka += da.kstep; c_kb += db.kstep;                                                                             \
    if (--c_left == 0) {                                                                                            \
      c_idx += t_stride; c_ka = da.kbeg; c_kb = db.kbeg;                                                            \
      if (c_idx < t_cnt) {                                                                                          \
        int tm_, tn_; w4_tile_rc(t_start + c_idx, ntn, ntm, p.band_cw, tm_, tn_);                                   \
        c_ta = w4_tile_base<TN>(da, tm_ * 256); c_tb = w4_tile_base<TN>(db, tn_ * 256);                             \
        if (convb) { c_tb = 0u; w4_conv_set_tile(db, tn_ * 256); }                                                  \
        c_left = nk;                                                                                                \
      } else { c_ta = W4_OOB; c_tb = W4_OOB; c_left = 0x7fffffff; }                                                 \
    }                                                                                                               \
  } while (0)

  f32x16 acc[4][4];
  s16x8 A0[2][4], A1[2][4], B1[2][4], BX[2][4], BY[2][4];

  // ---- prologue: the stream's k-tiles 0 and 1 completely, in the steady state's issue order (RA0, RB0, RB1, RA1) --------------------------------------
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    unsigned char* const buf = smem + b * W4_TILEBUF;
    w4_region(da, buf + W4_RA0, 0, w, c_ta + c_ka); w4_region(db, buf + W4_RB0, 0, w, c_tb + c_kb);
    w4_region(db, buf + W4_RB1, 1, w, c_tb + c_kb); w4_region(da, buf + W4_RA1, 1, w, c_ta + c_ka);
    W4_CURSOR_ADVANCE();
  }
  W4_WAIT_VM(4);                                          // deep wait: everything but the newest group

  int dbg_i = 0;
  for (;;) {
    unsigned long long t_top = 0, t_main = 0;
    if (p.dbg) t_top = __builtin_readcyclecounter();
    // ---- one output tile: nk k-tiles.  Its first fragments come from buffer 0 (landed: the deep wait above / in the previous epilogue, made workgroup-wide by the
    // barrier); the first k-tile starts the accumulators from zero and needs no DMA waits; the last one leaves the fragment registers to the epilogue ----------
    W4_BAR();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { A0[rt][ks] = w4_frag<TN>(smem + W4_RA0, F.a, rt, ks); BX[rt][ks] = w4_frag<TN>(smem + W4_RB0, F.b, rt, ks); }
    __builtin_amdgcn_sched_barrier(0);
    W4_WAIT_LGKM0();
    w4_ktile<TN, 0, -1, true, true, OF>(smem, F, da, db, c_ta + c_ka, c_tb + c_kb, w, acc, A0, A1, BX, B1, BY);
    W4_CURSOR_ADVANCE();
    for (int t = 2; t < nk; t += 2) {
      w4_ktile<TN, 1, 20, false, true, OF>(smem, F, da, db, c_ta + c_ka, c_tb + c_kb, w, acc, A0, A1, BY, B1, BX);
      W4_CURSOR_ADVANCE();
      w4_ktile<TN, 0, 20, false, true, OF>(smem, F, da, db, c_ta + c_ka, c_tb + c_kb, w, acc, A0, A1, BX, B1, BY);
      W4_CURSOR_ADVANCE();
    }
    int tm, tn; w4_tile_rc(t_start + t_idx, ntn, ntm, p.band_cw, tm, tn);
    constexpr bool AUXPRE = W4_AUX_PREFETCH && !TN && E != E_GENERIC && (E & E_DGELU) && ((E & ~(E_AUXD | E_OCS)) == E_DGELU);
    u32x4 auxpre[8];
    if constexpr (AUXPRE) w4_aux_prefetch(p, auxpre, lane, wr, wc, tm * 256, tn * 256);
    w4_ktile<TN, 1, 20, false, false, OF>(smem, F, da, db, c_ta + c_ka, c_tb + c_kb, w, acc, A0, A1, BY, B1, BX);
    W4_CURSOR_ADVANCE();
    if (p.dbg) t_main = __builtin_readcyclecounter();
    // The accumulators become opaque here: without it the compiler reads (and shuffles) them from inside the last MFMA phase, a few wait states behind the MFMA
    // that writes them, and on the MI355X some (lane, register) pairs then carry whatever the register held before (measured: the columns 4 g + 2, 4 g + 3 of
    // one 32-column block, lanes with bit 2 set).  The s_nop covers the last MFMA's passes.
#ifndef VDK_EMU
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                   "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]));
#endif
    unsigned long long ts[5] = {0, 0, 0, 0, 0};
    w4_epilogue<E, 4, 4, OF>(p, smem + W4_STAGE + w * 8192, acc, lane, wr, wc, tm * 256, tn * 256, z, tm, ts, AUXPRE ? auxpre : nullptr);
    if (p.dbg && tid == 0 && dbg_i < 8) {   // debug only: shader-cycle stamps of this workgroup's first 8 tiles: top, main loop done, epilogue issued
      unsigned long long* o = p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + dbg_i) * 8;   /* (debug stamps index the launch grid, not the item order) */
      o[0] = t_top; o[1] = t_main; o[2] = __builtin_readcyclecounter(); o[3] = (unsigned long long)(t_start + t_idx);
      o[4] = ts[0]; o[5] = ts[1]; o[6] = ts[2]; o[7] = ts[3];
    }
    ++dbg_i;
    t_idx += t_stride;
    if (t_idx >= t_cnt) break;
  }
  W4_WAIT_VM(0);                                          // the cursor's last (out-of-range) pieces still write zeros into this workgroup's LDS
#undef W4_CURSOR_ADVANCE
}

// (round 6, measured null: s_setprio 3 in this kernel's main loop and 0 in its epilogue -- the multiplying workgroup ahead of the other one's epilogue arithmetic in the issue
//  arbitration -- changed no ViT shape by more than +-1 % (fc1 + GELU 329-333 us either way); the same in the CBIR scan's two wave groups: 3.167 / 3.173 ms)
// =====================================================================================================================================
// gemm_w4h_kernel — the same building blocks as a 256 x 128 tile with TWO workgroups per CU (80 KB of LDS and 256 registers per wave each): every SIMD hosts
// one wave of each workgroup, so while one workgroup runs its epilogue (the GELU / dGELU arithmetic, the fp32 residual traffic: 30-70 k cycles in which a
// single resident workgroup leaves the matrix pipe idle) the other one multiplies.  Per wave: 128 x 64 = 4 x 2 accumulators.
//   * a k-tile is 3 regions of 16 KB: RA0 / RA1 (first / second 64 rows of both wave-rows) and RB (the 128 columns); 5 region slots form the ring, the stream's
//     region q = 3 t + r lives in slot q % 5 and is refilled with region q + 5 once it has been read;
//   * 2 phases of 16 MFMAs per k-tile: P1 = A0 x B (reads A1(t); refills the slots of RA0(t) and RB(t) with RA1(t+1) and RA0(t+2)), P2 = A1 x B (reads A0(t+1),
//     and B(t+1) IN PLACE: a B fragment register is reloaded right behind the last MFMA of the phase that uses it; refills RA1(t)'s slot with RB(t+2));
//   * every phase end waits vmcnt(8): the group read next was issued two phases earlier, only the 8 newest pieces may be in flight.
#define W4H_SMEM 81920
#define W4H_SLOT 16384

template <bool TN, bool FIRST, int OF>
__device__ __forceinline__ void w4h_ktile(unsigned char* smem, const W4Frag<TN>& F, const W4Dma& da, const W4Dma& db, unsigned s_a1n /* RA1(t+1) */, unsigned s_a0nn /* RA0(t+2) */,
                                          unsigned s_bnn /* RB(t+2) */, int w, unsigned oA0, unsigned oB, unsigned oA1, unsigned oA0n, unsigned oBn,
                                          f32x16 (&acc)[4][2], s16x8 (&A0)[2][4], s16x8 (&A1)[2][4], s16x8 (&B)[2][4]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // ---- P1: A0 x B ----
  W4_BAR();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    acc[0][0] = vdk_mfma32<OF>(B[0][ks], A0[0][ks], (FIRST && ks == 0) ? zero : acc[0][0]);
    A1[0][ks] = w4_frag<TN>(smem + oA1, F.a, 0, ks);
    acc[0][1] = vdk_mfma32<OF>(B[1][ks], A0[0][ks], (FIRST && ks == 0) ? zero : acc[0][1]);
    A1[1][ks] = w4_frag<TN>(smem + oA1, F.a, 1, ks);
    acc[1][0] = vdk_mfma32<OF>(B[0][ks], A0[1][ks], (FIRST && ks == 0) ? zero : acc[1][0]);
    // issue order inside the phase: the 4 pieces of RA1(t+1) first, then the 4 of RA0(t+2) (the vmcnt(8) of the phase ends counts on it)
    if (ks < 2) w4_piece(da, smem + oA0 + (4 * w + 2 * ks) * 1024, 1, 2 * ks, s_a1n); else w4_piece(da, smem + oB + (4 * w + 2 * (ks - 2)) * 1024, 0, 2 * (ks - 2), s_a0nn);
    acc[1][1] = vdk_mfma32<OF>(B[1][ks], A0[1][ks], (FIRST && ks == 0) ? zero : acc[1][1]);
    if (ks < 2) w4_piece(da, smem + oA0 + (4 * w + 2 * ks + 1) * 1024, 1, 2 * ks + 1, s_a1n); else w4_piece(da, smem + oB + (4 * w + 2 * (ks - 2) + 1) * 1024, 0, 2 * (ks - 2) + 1, s_a0nn);
  }
#ifndef VDK_EMU
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
  }
#endif
  w4_phase_end<8>();
  // ---- P2: A1 x B ----
  W4_BAR();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    acc[2][0] = vdk_mfma32<OF>(B[0][ks], A1[0][ks], (FIRST && ks == 0) ? zero : acc[2][0]);
    A0[0][ks] = w4_frag<TN>(smem + oA0n, F.a, 0, ks);
    acc[2][1] = vdk_mfma32<OF>(B[1][ks], A1[0][ks], (FIRST && ks == 0) ? zero : acc[2][1]);
    A0[1][ks] = w4_frag<TN>(smem + oA0n, F.a, 1, ks);
    acc[3][0] = vdk_mfma32<OF>(B[0][ks], A1[1][ks], (FIRST && ks == 0) ? zero : acc[3][0]);
    w4_piece(db, smem + oA1 + (4 * w + ks) * 1024, 0, ks, s_bnn);
    acc[3][1] = vdk_mfma32<OF>(B[1][ks], A1[1][ks], (FIRST && ks == 0) ? zero : acc[3][1]);
    B[0][ks] = w4_frag<TN>(smem + oBn, F.b, 0, ks);      // (k-tile t+1's fragments, into the registers the four MFMAs above have just read)
    B[1][ks] = w4_frag<TN>(smem + oBn, F.b, 1, ks);
  }
#ifndef VDK_EMU
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, TN ? 4 : 2, 0);
  }
#endif
  w4_phase_end<8>();
}

template <bool TN, int E, int OF>
__global__ __launch_bounds__(256, 2) void gemm_w4h_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[W4H_SMEM];   // 5 region slots; the epilogue staging (4 x 8 KB) reuses the first two
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int ntn = (p.N + 127) / 128, ntm = (p.M + 255) / 256;
  const int nwg = ntn * ntm;
  const int z = blockIdx.y;
  int tile;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn; w4_tile_rc(tile, ntn, ntm, p.band_cw, tm, tn);
  const int m0 = tm * 256, n0 = tn * 128;
  const int kbeg = z * p.k_per_split;
  int kend = kbeg + p.k_per_split; if (kend > p.K) kend = p.K;
  const int nk = (kend - kbeg) / 64;                      // launcher guarantees whole k-tiles, nk >= 1

  W4Dma da, db;
  w4_dma_init<TN>(da, p.A, p.lda, p.M, p.K, kbeg, w, lane, 128u);
  w4_dma_init<TN>(db, p.B, p.ldb, p.N, p.K, kbeg, w, lane, 64u);
  W4Frag<TN> F;
  w4_frag_init<TN>(F, wr, wc, lane);
  const unsigned ta = w4_tile_base<TN>(da, m0) + da.kbeg, tb = w4_tile_base<TN>(db, n0) + db.kbeg;
#define W4H_KA(t) ((t) < nk ? ta + (unsigned)(t) * da.kstep : W4_OOB)
#define W4H_KB(t) ((t) < nk ? tb + (unsigned)(t) * db.kstep : W4_OOB)

  f32x16 acc[4][2];
  s16x8 A0[2][4], A1[2][4], B[2][4];
  unsigned long long t_top = 0, t_land = 0, t_main = 0, rt_top = 0;
  if (p.dbg) {
    t_top = __builtin_readcyclecounter();
#ifndef VDK_EMU
    rt_top = __builtin_amdgcn_s_memrealtime();
#endif
  }
  // Optional start delay for the second slot's first workgroups (p.stagger cycles; VDK_GEMM_W4H_STAGGER).  The idea: the dispatcher fills both slots of every
  // CU at the same moment and equal tiles could keep the pair in lock-step (both in the prologue, both in the main loop, both in the epilogue).  Measured
  // (tools/w4h_stamps.py, per-CU time lines from s_memrealtime + HW_ID): the pair is 8-10 us apart after its first workgroups anyway, and a forced half-period
  // offset changed the long-epilogue GEMMs by 0 to -2 %.  Off by default.
#ifndef VDK_EMU
  if (p.stagger > 0 && blockIdx.y == 0 && blockIdx.x >= (unsigned)p.stagger_lo && blockIdx.x < (unsigned)(2 * p.stagger_lo)) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(16);
  }
#endif
  // ---- prologue: the stream's regions 0..4 (RA0(0), RB(0), RA1(0), RA0(1), RB(1)) into slots 0..4 ----
  w4_region(da, smem + 0 * W4H_SLOT, 0, w, W4H_KA(0)); w4_region(db, smem + 1 * W4H_SLOT, 0, w, W4H_KB(0)); w4_region(da, smem + 2 * W4H_SLOT, 1, w, W4H_KA(0));
  w4_region(da, smem + 3 * W4H_SLOT, 0, w, W4H_KA(1)); w4_region(db, smem + 4 * W4H_SLOT, 0, w, W4H_KB(1));
  W4_WAIT_VM(12);                                         // RA0(0), RB(0)
  W4_BAR();
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { A0[rt][ks] = w4_frag<TN>(smem, F.a, rt, ks); B[rt][ks] = w4_frag<TN>(smem + W4H_SLOT, F.b, rt, ks); }
  __builtin_amdgcn_sched_barrier(0);
  W4_WAIT_VM(8);                                          // RA1(0), read in phase 1
  W4_WAIT_LGKM0();

  if (p.dbg) t_land = __builtin_readcyclecounter();
  unsigned q5 = 0;                                        // (3 t) mod 5
#define W4H_SLOT_OF(i) ((((q5 + (i)) >= 5u) ? (q5 + (i) - 5u) : (q5 + (i))) * (unsigned)W4H_SLOT)
  w4h_ktile<TN, true, OF>(smem, F, da, db, W4H_KA(1), W4H_KA(2), W4H_KB(2), w, W4H_SLOT_OF(0), W4H_SLOT_OF(1), W4H_SLOT_OF(2), W4H_SLOT_OF(3), W4H_SLOT_OF(4), acc, A0, A1, B);
  q5 = 3;
  for (int t = 1; t < nk; ++t) {
    w4h_ktile<TN, false, OF>(smem, F, da, db, W4H_KA(t + 1), W4H_KA(t + 2), W4H_KB(t + 2), w, W4H_SLOT_OF(0), W4H_SLOT_OF(1), W4H_SLOT_OF(2), W4H_SLOT_OF(3), W4H_SLOT_OF(4), acc, A0, A1, B);
    q5 = q5 + 3 >= 5 ? q5 - 2 : q5 + 3;
  }
#undef W4H_SLOT_OF
#undef W4H_KA
#undef W4H_KB
  if (p.dbg) t_main = __builtin_readcyclecounter();
  W4_WAIT_VM(0);                                          // the out-of-range pieces behind the last k-tile still write zeros into the ring the epilogue is about to reuse
  W4_BAR();
#ifndef VDK_EMU
  asm volatile("s_nop 15\n\ts_nop 15"
               : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
#endif
  unsigned long long ts[5] = {0, 0, 0, 0, 0};
  w4_epilogue<E, 2, -1, OF>(p, smem + w * 8192, acc, lane, wr, wc, m0, n0, z, tm, ts);
  if (p.dbg && tid == 0) {   // debug only: shader-cycle stamps (start, first operands landed, main loop done, epilogue entered, its 32-row blocks 1..3, end)
    unsigned long long* o = p.dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    o[0] = t_top; o[1] = t_main; o[2] = __builtin_readcyclecounter(); o[3] = t_land; o[4] = ts[0];
#ifndef VDK_EMU
    o[5] = rt_top; o[6] = __builtin_amdgcn_s_memrealtime();                       // the 100 MHz counter is common to the chip: a time line across CUs
    o[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID, HW_ID
#endif
  }
}

// ---- launcher (called by vdk_gemm_bf16_nt / vdk_margin_cos_pass in gemm.hip) ------------------------------------------------------------------------
// Serves what the 8-wave 256x256 kernel serves except its a_colsum by-product, the token-row remap of the A operand and stream-K.  Operand byte sizes stay
// below 2 GB (32-bit buffer offsets, W4_OOB beyond them); every split holds an even number of k-tiles.
#if VDK_W4_OF == 0
bool vdk_gemm_w4_serves(const GemmParams& p, bool trans) {
  const double lim = 2147483648.0 - 65536.0;
  if (p.colsum_part || p.sk_cnt || p.a_row_group > 0) return false;
  if (p.conv_on) {      // only the implicit weight gradient (TN, B gathered): input tensor and dY below 2 GB
    if (!trans || (p.K % 128) || (p.k_per_split % 128) || p.K < 128) return false;
    const double in_bytes = (double)(p.crows / (p.cOH * p.cOW)) * p.cH * p.cW * p.cCin * 2.0;
    return in_bytes < lim && ((double)p.crows + 64.0) * (double)p.lda * 2.0 < lim && ((double)p.M + 256.0) * (double)p.ldc * 4.0 < lim;
  }
  if ((p.K % 128) || (p.k_per_split % 128) || p.K < 128) return false;
  if (((double)p.M + 256.0) * (double)p.ldc * (p.c_dtype == VDK_F32 ? 4.0 : 2.0) >= lim || (p.aux && ((double)p.M + 256.0) * (double)p.ldaux * 2.0 >= lim) ||
      (p.residual && ((double)p.M + 256.0) * (double)p.ldr * 4.0 >= lim)) return false;   // rows leave through 32-bit buffer offsets whose upper half marks "out of range"
  if (!trans) return ((double)p.M + 256.0) * (double)p.lda * 2.0 < lim && ((double)p.N + 256.0) * (double)p.ldb * 2.0 < lim;
  return ((double)p.K + 64.0) * (double)p.lda * 2.0 < lim && ((double)p.K + 64.0) * (double)p.ldb * 2.0 < lim;
}

// the 256x128 / two-workgroups-per-CU form: any whole number of k-tiles; same size limits
bool vdk_gemm_w4h_serves(const GemmParams& p, bool trans) {
  const double lim = 2147483648.0 - 65536.0;
  if (p.colsum_part || p.sk_cnt || p.a_row_group > 0 || p.conv_on) return false;
  if ((p.K % 64) || (p.k_per_split % 64) || p.K < 64) return false;
  if (((double)p.M + 256.0) * (double)p.ldc * (p.c_dtype == VDK_F32 ? 4.0 : 2.0) >= lim || (p.aux && ((double)p.M + 256.0) * (double)p.ldaux * 2.0 >= lim) ||
      (p.residual && ((double)p.M + 256.0) * (double)p.ldr * 4.0 >= lim)) return false;
  if (!trans) return ((double)p.M + 256.0) * (double)p.lda * 2.0 < lim && ((double)p.N + 256.0) * (double)p.ldb * 2.0 < lim;
  return ((double)p.K + 64.0) * (double)p.lda * 2.0 < lim && ((double)p.K + 64.0) * (double)p.ldb * 2.0 < lim;
}

// CUs the persistent walk leaves free (vdk_gemm_reserve_cus; environment VDK_GEMM_RESERVE_CUS overrides).  A persistent grid that covers every CU with a
// workgroup owning the CU's whole LDS cannot share the chip: a collective's kernel on another stream (RCCL: one workgroup per channel) that occupies R CUs when a
// GEMM starts keeps R of its workgroups -- and the tiles the static walk gave them -- waiting until the collective ends.  With R CUs left out of the walk both fit.
static std::atomic<int> g_w4_reserve{0};

extern "C" int vdk_gemm_reserve_cus(int32_t n) {
  if (n < 0 || n > 128) return vdk_fail(VDK_EINVAL, "vdk_gemm_reserve_cus: 0 <= n <= 128");
  g_w4_reserve.store(n);
  return VDK_OK;
}
extern "C" int vdk_gemm_reserved_cus(void) { return g_w4_reserve.load(); }

// diagnostic (tools/w4_contention.py): `workgroups` workgroups that each hold a CU's LDS share of a collective's kernel for `microseconds`, on `stream` -- stands in for an
// RCCL all-reduce in flight when the effect of vdk_gemm_reserve_cus is measured on one GPU
__global__ __launch_bounds__(256) void w4_occupy_kernel(long ticks, int* sink) {
  __shared__ volatile int pad[8192];                       // 32 KB: cannot share a CU's LDS with a workgroup that owns all of it
  pad[threadIdx.x] = (int)threadIdx.x;
#ifndef VDK_EMU
  asm volatile("v_mov_b32 v127, 0" ::: "v127");            // a 128-register wave, as a collective's kernel is: does not fit beside a wave that owns the SIMD's register file
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(64);
#endif
  if (sink && pad[(threadIdx.x + 1) & 255] < 0) *sink = 1;
}

extern "C" int vdk_debug_occupy_cus(int32_t workgroups, int64_t microseconds, void* stream) {
  if (workgroups <= 0 || microseconds < 0 || microseconds > 1000000) return vdk_fail(VDK_EINVAL, "vdk_debug_occupy_cus: bad argument");
  hipLaunchKernelGGL(w4_occupy_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (long)microseconds * 100, (int*)nullptr);
  return hipGetLastError() == hipSuccess ? VDK_OK : vdk_fail(VDK_ELAUNCH, "vdk_debug_occupy_cus: launch failed");
}

int vdk_gemm_w4_cus() {
  static int cus = 0;
  static int env_reserve = getenv("VDK_GEMM_RESERVE_CUS") ? atoi(getenv("VDK_GEMM_RESERVE_CUS")) : -1;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    cus = n & ~7;
  }
  const int r = env_reserve >= 0 ? env_reserve : g_w4_reserve.load();
  int g = (cus - r) & ~7;                                  // (the walk wants a multiple of 8: one share per XCD)
  return g < 8 ? 8 : g;
}
int g_w4_force_band_cw = -1;
extern "C" int vdk_gemm_force_band_cw(int32_t cw) { g_w4_force_band_cw = cw; return VDK_OK; }   /* tests: the band order on small problems (-1: the size rule) */
#else
extern int g_w4_force_band_cw;
#endif
static int w4_cus() { return vdk_gemm_w4_cus(); }

#define W4_LAUNCH(TNF, EE)                                                                                                                    \
  do {                                                                                                                                        \
    if (persist) {                                                                                                                            \
      if (ev0 || ev1) hipExtLaunchKernelGGL((gemm_w4_kernel<TNF, EE, true, VDK_W4_OF>), pgrid, dim3(256), 0, stream, (hipEvent_t)ev0, (hipEvent_t)ev1, 0, p);   \
      else hipLaunchKernelGGL((gemm_w4_kernel<TNF, EE, true, VDK_W4_OF>), pgrid, dim3(256), 0, stream, p);                                               \
    } else {                                                                                                                                  \
      if (ev0 || ev1) hipExtLaunchKernelGGL((gemm_w4_kernel<TNF, EE, false, VDK_W4_OF>), grid, dim3(256), 0, stream, (hipEvent_t)ev0, (hipEvent_t)ev1, 0, p);   \
      else hipLaunchKernelGGL((gemm_w4_kernel<TNF, EE, false, VDK_W4_OF>), grid, dim3(256), 0, stream, p);                                               \
    }                                                                                                                                         \
    return true;                                                                                                                              \
  } while (0)

static bool w4_launch_one(const GemmParams& p, bool trans, int E, unsigned tiles, unsigned splitk, hipStream_t stream, void* ev0, void* ev1);

bool W4_SYM(vdk_gemm_w4_launch)(const GemmParams& p, bool trans, int E, unsigned tiles, unsigned splitk, void* stream_, void* ev0, void* ev1) {
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned G = (unsigned)w4_cus();
  // The ragged last round.  T tiles on G CUs take ceil(T / G) rounds of whole tiles (591 on 256: 2.31 -> 3).  When the tile rows beyond the last whole round are
  // few (at most G / 2 tiles) and the k-range is long enough to pay for a second launch, the whole rounds go to the persistent kernel and the remaining ROWS to the
  // 256x128 kernel, whose workgroups then each have a CU to themselves: 2.31 -> ~2.5 rounds.  Both launches are ordinary GEMMs over a row range of the operands.
  // (round 6: off by default -- with the start phase of the walk the unsplit persistent launch is the faster one: fc2 247.7 against 253.6 us, dfc1 213.6 / 216.8, dqkv 161.7 / 164.6;
  //  VDK_GEMM_W4_SPLIT=1 restores the two-launch form; re-read per launch: A/B runs in one process)
  const bool split_on = getenv("VDK_GEMM_W4_SPLIT") && atoi(getenv("VDK_GEMM_W4_SPLIT")) == 1;
  const int split_min_k = getenv("VDK_GEMM_W4_SPLIT_K") ? atoi(getenv("VDK_GEMM_W4_SPLIT_K")) : 1536;
  // (round 6, measured and removed: the rows beyond the whole rounds as a 3-way split-K launch of this kernel + a slab reduce / epilogue pass -- fc2 277.4 -> 273.2 / 282.0 us,
  //  dfc1 245.2 -> 239.5 / 244.6, dqkv 165.6 -> 177.1 / 177.8, same box, two rounds: the two launch boundaries and 64 MB of slab traffic cost what the idle CUs gained)
  if (split_on && !trans && splitk == 1 && tiles > G && p.K >= split_min_k && E != E_GENERIC && !(E & (E_ROWGRP | E_MSTAT | E_MGRAD | E_SPLITK))) {
    const unsigned ntn = (unsigned)((p.N + 255) / 256), ntm = (unsigned)((p.M + 255) / 256);
    const unsigned rows1 = (tiles / G) * G / ntn;         // tile rows covered by whole rounds
    const unsigned rem_tiles = (ntm - rows1) * ntn;
    if (rows1 > 0 && rem_tiles > 0 && 2 * rem_tiles <= G) {
      GemmParams p1 = p, p2 = p;
      const long m1 = (long)rows1 * 256;
      p1.M = (int)m1;
      p2.M = p.M - (int)m1;
      p2.A = p.A + m1 * p.lda;
      p2.C = (char*)p.C + m1 * p.ldc * ((E & E_F32) ? 4 : 2);
      if (p.residual) p2.residual = p.residual + m1 * p.ldr;
      if (p.aux) p2.aux = p.aux + m1 * p.ldaux;
      if (p.ocs_part) p2.ocs_part = p.ocs_part + (size_t)rows1 * 2 * p.N;
      if (vdk_gemm_w4h_serves(p2, false)) {
        w4_launch_one(p1, false, E, rows1 * ntn, 1u, stream, ev0, nullptr);
        return W4_SYM(vdk_gemm_w4h_launch)(p2, false, E, 1u, stream_, nullptr, ev1);
      }
    }
  }
  return w4_launch_one(p, trans, E, tiles, splitk, stream, ev0, ev1);
}

// column-band width of the tile order (0: plain row-major): band when the re-streamed B operand costs more than the repeated A panels.  tile_n: 256 or 128;
// G: workgroups in flight (one round)
static int w4_band_cw(const GemmParams& p, bool trans, unsigned tiles, unsigned G, int tile_n) {
  if (g_w4_force_band_cw >= 0) return (trans || p.conv_on) ? 0 : g_w4_force_band_cw;
  static const bool on = [] { const char* e = getenv("VDK_GEMM_BANDS"); return !(e && e[0] == '0'); }();
  if (!on || trans || p.conv_on) return 0;
  const double bbytes = (double)p.N * p.K * 2.0, abytes = (double)p.M * p.K * 2.0;
  if (bbytes <= 2.0e6) return 0;                                       // B stays in an XCD's L2 anyway
  const int ntn = (p.N + tile_n - 1) / tile_n;
  int nb = (int)((bbytes + 2.0e6 - 1.0) / 2.0e6); if (nb > ntn) nb = ntn;
  const double rounds = (double)tiles / (double)G;
  if (nb < 2 || 8.0 * bbytes * (rounds - 1.0) <= (nb - 1) * abytes) return 0;
  return (ntn + nb - 1) / nb;
}
static bool w4_sk_xcd() { static const bool v = [] { const char* e = getenv("VDK_GEMM_SK_XCD"); return !(e && e[0] == '0'); }(); return v; }
static bool w4_launch_one(const GemmParams& p_, bool trans, int E, unsigned tiles, unsigned splitk, hipStream_t stream, void* ev0, void* ev1) {
  GemmParams p = p_;
  p.sk_xcd = (splitk > 1 && E == E_SPLITK && w4_sk_xcd()) ? 1 : 0;
  p.band_cw = splitk == 1 ? w4_band_cw(p, trans, tiles, (unsigned)w4_cus(), 256) : 0;
  const unsigned items = tiles * splitk;
  const dim3 grid = p.sk_xcd ? dim3(8u * ((items + 7u) / 8u), 1u) : dim3(tiles, splitk);
  const unsigned G = (unsigned)w4_cus();
  const bool persist = splitk == 1 && tiles > G;          // more tiles than CUs: walk them (with one tile per workgroup there is nothing to prefetch)
  const dim3 pgrid(G, 1u);
  p.stagger = 0; p.stagger_lo = 0;
  if (persist) {
    // start phase of the walk (see the kernel): percent of an estimated tile period; VDK_GEMM_W4_STAGGER = "<percent>[,all]" is re-read per launch (A/B runs in one process)
    // Measured (tools/r6_stagger_ab.py, ViT-B/16 shapes at T = 50 432, fp16, us, off / 50 % / 90 % / 100 % for everybody): proj + fp32 residual 116.5 / 104.7 / 102.4 / 109.7,
    // fc2 + residual 261.5 / 247.7 / 247.7 / 270.9, dfc2 x saved derivative 284.5 / 275.0 / 280.8 / 286.4, dfc1 220.4 / 214.8 / 213.6 / 234.8, dqkv 169.0 / 163.8 / 161.7 / 176.8.
    int pct = 75, all = 0;
    if (const char* e = getenv("VDK_GEMM_W4_STAGGER")) { pct = atoi(e); all = strstr(e, "all") != nullptr; }
    if (pct > 0) {
      const int nk = p.K / 64;
      int epi = 6000;
      if (E != E_GENERIC && (E & E_GELU)) epi = 20000; else if (E != E_GENERIC && (E & E_DGELU)) epi = 20000; else if (E == E_GENERIC || (E & (E_RES | E_F32))) epi = 25000;
      p.stagger = (int)((long)(nk * 2300 + epi) * pct / 100);
      p.stagger_lo = all;
    }
  }
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
    case E_BIAS | E_GELU | E_AUXD: W4_LAUNCH(false, E_BIAS | E_GELU | E_AUXD);
    case E_DGELU | E_AUXD: W4_LAUNCH(false, E_DGELU | E_AUXD);
    case E_DGELU | E_OCS | E_AUXD: W4_LAUNCH(false, E_DGELU | E_OCS | E_AUXD);
    case E_BIAS | E_RES | E_F32: W4_LAUNCH(false, E_BIAS | E_RES | E_F32);
    case E_BIAS | E_F32: W4_LAUNCH(false, E_BIAS | E_F32);
    case E_BIAS | E_RES | E_F32 | E_ROWGRP: W4_LAUNCH(false, E_BIAS | E_RES | E_F32 | E_ROWGRP);
    case E_SPLITK: W4_LAUNCH(false, E_SPLITK);
    case E_F32: W4_LAUNCH(false, E_F32);
    default: W4_LAUNCH(false, E_GENERIC);
  }
  return false;
}

#define W4H_LAUNCH(TNF, EE)                                                                                                             \
  do {                                                                                                                                  \
    if (ev0 || ev1) hipExtLaunchKernelGGL((gemm_w4h_kernel<TNF, EE, VDK_W4_OF>), grid, dim3(256), 0, stream, (hipEvent_t)ev0, (hipEvent_t)ev1, 0, p);     \
    else hipLaunchKernelGGL((gemm_w4h_kernel<TNF, EE, VDK_W4_OF>), grid, dim3(256), 0, stream, p);                                                 \
    return true;                                                                                                                        \
  } while (0)

bool W4_SYM(vdk_gemm_w4h_launch)(const GemmParams& p_, bool trans, int E, unsigned splitk, void* stream_, void* ev0, void* ev1) {
  hipStream_t stream = (hipStream_t)stream_;
  GemmParams p = p_;
  const dim3 grid((unsigned)(((p.M + 255) / 256) * ((p.N + 127) / 128)), splitk);
  p.band_cw = splitk == 1 ? w4_band_cw(p, trans, grid.x, 2u * (unsigned)w4_cus(), 128) : 0;
  {
    // start delay of the second slot's first workgroups: half of (prologue + main loop at ~2400 cycles per k-tile with a shared pipe + epilogue estimate)
    static const int stag_pct = getenv("VDK_GEMM_W4H_STAGGER") ? atoi(getenv("VDK_GEMM_W4H_STAGGER")) : 0;   // measured null to -2 % at 50 / 100 / 150 % (tools/w4h_stagger_ab.py): the pair drifts apart by itself; opt-in
    const int cus = w4_cus();
    const int nk = (p.k_per_split < p.K ? p.k_per_split : p.K) / 64;
    int epi = 7000;
    if (E != E_GENERIC && (E & E_GELU)) epi = 20000; else if (E != E_GENERIC && (E & E_DGELU)) epi = 30000; else if (E == E_GENERIC || (E & (E_RES | E_F32 | E_SPLITK))) epi = 30000;
    p.stagger_lo = cus;
    p.stagger = ((int)grid.x >= 2 * cus && splitk == 1) ? (int)((long)(4000 + nk * 2400 + epi) / 2 * stag_pct / 100) : 0;
  }
  if (trans) {
    switch (E) {
      case E_SPLITK: W4H_LAUNCH(true, E_SPLITK);
      case E_F32: W4H_LAUNCH(true, E_F32);
      case 0: W4H_LAUNCH(true, 0);
      default: W4H_LAUNCH(true, E_GENERIC);
    }
  }
  switch (E) {
    case 0: W4H_LAUNCH(false, 0);
    case E_OCS: W4H_LAUNCH(false, E_OCS);
    case E_DGELU | E_OCS: W4H_LAUNCH(false, E_DGELU | E_OCS);
    case E_BIAS: W4H_LAUNCH(false, E_BIAS);
    case E_BIAS | E_GELU: W4H_LAUNCH(false, E_BIAS | E_GELU);
    case E_DGELU: W4H_LAUNCH(false, E_DGELU);
    case E_BIAS | E_GELU | E_AUXD: W4H_LAUNCH(false, E_BIAS | E_GELU | E_AUXD);
    case E_DGELU | E_AUXD: W4H_LAUNCH(false, E_DGELU | E_AUXD);
    case E_DGELU | E_OCS | E_AUXD: W4H_LAUNCH(false, E_DGELU | E_OCS | E_AUXD);
    case E_BIAS | E_RES | E_F32: W4H_LAUNCH(false, E_BIAS | E_RES | E_F32);
    case E_BIAS | E_F32: W4H_LAUNCH(false, E_BIAS | E_F32);
    case E_BIAS | E_RES | E_F32 | E_ROWGRP: W4H_LAUNCH(false, E_BIAS | E_RES | E_F32 | E_ROWGRP);
    case E_SPLITK: W4H_LAUNCH(false, E_SPLITK);
    case E_F32: W4H_LAUNCH(false, E_F32);
    default: W4H_LAUNCH(false, E_GENERIC);
  }
  return false;
}

// attention_small.hip — K3 for short sequences (N <= 256 keys: ViT-B/16 at 224 has N = 197): the whole (batch, head) problem lives in one CU's LDS.
// Same operator as attention.hip (timm `Attention`: softmax(q k^T / sqrt(hd)) v behind models/classifier/classify_model.py:49-54 and
// models/faceX/backbone/timm_wrapper.py:16-21), different structure:
//
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4, 8 rows of 128 B per wave-instruction): no staging registers, so the kernels stay
//     under 256 VGPRs and keep two waves per SIMD (the register-prefetching forward of attention.hip sat at 270 registers = ONE wave per SIMD, i.e.
//     no overlap at all between one (batch, head) item's loads and another's MFMAs: 153 us per layer against a 62 us HBM floor);
//   * the 16-byte chunk c of row r is stored at chunk position c ^ f(r), f(r) = 4*bit1(r) + bits3:2(r) (swizzle on the DMA's SOURCE address,
//     cdna_hip_programming.md rule 21): conflict-free both for ds_read_b128 row fragments and for ds_read_b64_tr_b16 transposed fragments
//     (tools/lds_banks.py prints the bank multiplicities);
//   * FORWARD keeps all of S^T = K Q^T for a 32-query tile in registers (16 * NKT accumulators), so the softmax is the textbook one:
//     exact row maximum, P normalised in fp32 and rounded to bf16 ONCE as the operand of P V — the rounding point autocast has in the reference
//     (engine/procedure/train.py:118) and the one oracle/bf16ops.py restates; no online rescale, no data-dependent branch;
//   * BACKWARD is ONE kernel with 5 GEMM-equivalents per (query tile, key tile) block instead of two kernels with 7: a wave owns a key tile (dK, dV
//     accumulate in its registers over the query tiles); dS is handed through a wave-private 2 KB LDS tile to become the B operand of
//     dQ^T += K^T dS^T, and dQ accumulates in an fp32 LDS tile per query tile.  The waves walk the query tiles staggered (wave w takes tile
//     (w + t) mod n in step t), so no two waves touch the same dQ tile inside a step: no atomics, a fixed summation order, bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "vdk_device.h"
#include "vdk_host.h"

#include "vdk_attn_tile.h"


// =====================================================================================  forward
// LDS (dynamic): Q [R8 rows] | K [R8] | V [R8] | zero rows up to 32*NKT of the V array.  R8 = N rounded up to 8.  Tile reads beyond R8 fall into the
// next array (finite data whose contribution is masked) or into the zero rows (V: P is exactly 0 there).  A wave's O tile is staged in its own Q rows.
template <int NKT, int OF = 0>
__global__ __launch_bounds__(256, (NKT <= 7 ? 2 : 1)) void attn_s_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                                              bf16_t* __restrict__ o, long ldo, float* __restrict__ lse, int N, int H, float scale, int nitems) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R8 = (N + 7) & ~7;
  unsigned char* const Qs = smem;
  unsigned char* const Ks = smem + R8 * AS_ROW;
  unsigned char* const Vs = smem + 2 * R8 * AS_ROW;
  for (int i = tid * 16; i < (32 * NKT - R8) * AS_ROW; i += 256 * 16) *(u32x4*)(Vs + R8 * AS_ROW + i) = (u32x4){0u, 0u, 0u, 0u};
  const int nqt = (N + 31) >> 5;
  const float scale2 = scale * VDK_LOG2E;
  const AsLane al = as_lane(lane);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64;
    __syncthreads();                                   // the previous item's readers are done (first pass: the zero rows are written)
    as_dma_rows(Qs, q + off, ld, N, R8, w, 4, lane);
    as_dma_rows(Ks, k + off, ld, N, R8, w, 4, lane);
    as_dma_rows(Vs, v + off, ld, N, R8, w, 4, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's DMAs have landed
    __syncthreads();
    for (int qt = w; qt < nqt; qt += 4) {
      const int qrow = qt * 32 + l31;
      s16x8 qf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[ks] = as_row_frag(Qs, qrow, ks, hi);
      f32x16 st[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        st[kt] = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st[kt] = vdk_mfma32<OF>(as_row_frag_l(Ks + kt * 32 * AS_ROW, al, ks), qf[ks], st[kt]);
      }
      if (N & 31) {                                    // ragged last key tile (wave-uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) st[NKT - 1][r] = -INFINITY;
      }
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, st[kt][r]);
      m = fmaxf(m, __shfl_xor(m, 32));                 // the two half-waves hold the same queries, different keys
      const float m2 = m * scale2;
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float p = fast_exp2(fmaf(st[kt][r], scale2, -m2)); st[kt][r] = p; l += p; }
      l += __shfl_xor(l, 32);
      const float inv = 1.0f / l;
      f32x16 o0 = as_zero16(), o1 = as_zero16();
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f32x16 pn;
#pragma unroll
        for (int r = 0; r < 16; ++r) pn[r] = st[kt][r] * inv;
        s16x8 pf[2];
        as_pack_b<OF>(pn, pf);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          o0 = vdk_mfma32<OF>(as_tr_frag_l(Vs + (kt * 32 + 16 * s) * AS_ROW, al, 0), pf[s], o0);
          o1 = vdk_mfma32<OF>(as_tr_frag_l(Vs + (kt * 32 + 16 * s) * AS_ROW, al, 1), pf[s], o1);
        }
      }
      as_store_tile<OF>(Qs + qt * 32 * AS_ROW, o0, o1, 1.0f, o + (long)b * N * ldo + h * 64, ldo, qt * 32, N, lane);
      if (lse && hi == 0 && qrow < N) lse[((long)b * H + h) * N + qrow] = (m2 + log2f(l)) * 0.6931471805599453f;
    }
  }
}

// =====================================================================================  backward (one pass, dS exchanged between the waves)  [form 5]
// The one-pass structure of form 4 (wave = key tile, query tiles in the outer loop: S, dP, P, dS once; dV^T / dK^T in registers, bit for bit the kv kernel's arithmetic)
// with a different answer to "dQ_j contracts over the key, i.e. over waves".  Form 4 let every key wave form a PARTIAL dQ_j tile (8 KB fp32) and a reducer wave add the
// seven up: 112 KB of LDS traffic and ~1.5 k serial cycles per query tile.  Here the waves exchange dS itself -- 2 KB of 16-bit values per wave and tile, [key][query] rows in
// a double-buffered 2 x 16 KB -- and the dQ_j^T tile (64 x 32) is split by OUTPUT block: wave w owns the 16 x 16 block (d block w >> 1, query block w & 1) and contracts it
// over ALL keys with seven v_mfma_f32_16x16x32, so nobody sums partials, the result is deterministic, every wave carries the same load, and ONE barrier per query tile is
// enough (it publishes tile j's dS and the landed Q / dO of tile j + 1; the dS buffer of tile j is rewritten two tiles later, behind the next barrier).
// MFMA work per pair: 16 x 32x32x16 + 7/8 x 16x16x32 against 28 in the two-kernel form; one exp per score instead of two.
// One workgroup per CU has nobody to hide a memory round trip behind, so nothing inside the tile loop waits for one: Q_j / dO_j tiles travel TWO tiles ahead through a ring
// of three buffers (the only vector-memory operations in the loop, so the single s_waitcnt before the barrier is a counted one: "all but the newest request"), and dQ is
// staged in LDS and leaves once per item as whole 128-byte rows.
// The Q_j / dO_j requests are RAW LDS-DMA (inline asm, vdk_attn_tile.h).  Round 6 finding: with the builtin form hipcc kept the DMA on its vmcnt scoreboard and put
// s_waitcnt vmcnt(0) in front of the tile's first transposing LDS read and in front of every barrier, i.e. the tile requested at the top of tile j (meant for tile j + 2) was
// waited for inside tile j: one exposed memory round trip per query tile.
// K^T fragments of the dQ role and the K / V row fragments of the key role are re-read from the K / V tiles (held in registers they push the kernel past 256: spills).
// LDS: 24 KB (Q_j, dO_j x 3) + 2 x 28 KB (K, V tiles) + 2 x 15.75 KB (dS; at an item's end the waves' store tiles) + 31.5 KB (dQ rows) + lse / D + 6 KB = 151 KB; 8 waves x 256 registers.
#define A5_PITCH 72                       /* bytes per key row of the dS exchange: 32 queries x 2 B + 8 (the 8-byte writes of 32 key lanes then hit 32 distinct bank pairs) */
#ifndef A5_KVREG
#define A5_KVREG 1                       /* the wave's K / V row fragments: 1 = read once per item into registers, 0 = re-read from the LDS tiles in every key phase */
#endif
#ifndef A5_SWAP
#define A5_SWAP 1                        /* 1: waves 0..3 run key phase j + 1 before dQ phase j, waves 4..7 after it */
#endif
#ifndef A5_KP
#define A5_KP 1                          /* key-phase order: 0 = the scheduler's (MFMAs, VALU, MFMAs), 1 = MFMAs interleaved with the softmax arithmetic by hand */
#endif
#define A5_QPITCH 144                     /* bytes per query row of the dQ staging: 64 d x 2 B + 16 */
template <int NKT, int OF>
__global__ __launch_bounds__(512) void attn_s_bwd5_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                          const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long ldd, int N, int H, float scale,
                                                          int nitems, int dbg /* timing experiments only (VDK_ATTN5_DBG): 1 skip the key phase, 2 the dQ phase, 4 D, 8 the stores */,
                                                          float* __restrict__ cspart /* optional f32 [B][3][H][64]: column sums over the item's tokens of the dq / dk / dv rows AS STORED
                                                                                        (the qkv.bias gradient's partial per image: no second pass over dqkv) */) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NP = 32 * NKT;
  constexpr int NPC = (NP * 8 + 511) / 512;
  constexpr int DSB = NP * A5_PITCH;                                  // one dS buffer
  unsigned char* const QO = smem;                                     // [buffer 3][Q | dO][32 rows x 128 B]
  unsigned char* const Kall = smem + 3 * 8192;                        // [NKT][4 KB] K tiles: wave w's row fragments (S) and every wave's K^T fragments (dQ role)
  unsigned char* const Kt = Kall + w * 4096;
  unsigned char* const Vt = Kall + NKT * 4096 + w * 4096;             // [NKT][4 KB] V tiles: wave w's row fragments (dP)
  unsigned char* const Ds = Kall + 2 * NKT * 4096;                    // [2][NP rows x A5_PITCH] dS [key][query]; behind an item's last barrier: the waves' 4 KB store tiles
  unsigned char* const DQs = Ds + 2 * DSB;                            // [NP rows x A5_QPITCH] dQ [query][d], 16-bit, already scaled
  float* const lse2 = (float*)(DQs + NP * A5_QPITCH);
  float* const Dv = lse2 + NP;
  float* const CSs = Dv + NP;                                         // [3: q, k, v][8 waves][64 d] column-sum partials of the item's stored rows (cspart != nullptr)
  constexpr int nt = NKT;                                             // (the launcher instantiates NKT = ceil(N / 32): every tile loop is a compile-time loop)
  const float scale2 = scale * VDK_LOG2E;
  const bool ragged = (N & 31) != 0;
  const AsLane al = as_lane(lane);
  const bool keyw = w < nt;                                           // (wave-uniform)
  // dQ role: block (d block db, query block qb) of the 64 x 32 tile dQ_j^T; operand fragments of v_mfma_f32_16x16x32: lane l -> row / column l & 15, k = 8 (l >> 4) + e.
  // A transposing read hands lane i (of a 16-lane group g) element i & 3 of the 8-byte chunks addressed by lanes (i >> 2) + {0, 4, 8, 12}: source lane a points at key row
  // 8 g + (a >> 2) (+ 4 for slots 4..7), chunk a & 3 of the block's 16 columns.
  const int db = w >> 1, qb = w & 1;
  const int a16 = lane & 15, g4 = lane >> 4;
  const int trow = 8 * g4 + (a16 >> 2);                               // key row inside a 32-key step
  const int ds_off = trow * A5_PITCH + qb * 32 + 8 * (a16 & 3);       // dS^T fragment (B): + (32 t) * A5_PITCH, second read + 4 * A5_PITCH
  int kt_off[2];                                                      // K^T fragment (A) inside a 4 KB K tile (swizzled 16-byte chunks): first / second read
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int r = trow + 4 * h2, byte = 32 * db + 8 * (a16 & 3);
    kt_off[h2] = r * AS_ROW + (((byte >> 4) ^ as_f(r)) << 4) + (byte & 8);
  }
  // one DMA instruction per wave fills a quarter of one operand's 32-row tile: waves 0..3 Q, 4..7 dO.  Raw LDS-DMA (vdk_attn_tile.h): the compiler neither waits for it in
  // front of the tile's transposing LDS reads nor drains it at the tile's barrier, so a tile requested two tiles ahead really travels for two tiles.
  auto request_tile = [&](int j, int buf, long off, long offo, int lane_) {
    const int q0 = j * 32, part = w & 3;
    if (w < 4) as_dma_rows_raw(QO + buf * 8192, q + off + (long)q0 * ld, ld, N - q0, 32, part, 4, lane_);
    else as_dma_rows_raw(QO + buf * 8192 + 4096, dout + offo + (long)q0 * ldo, ldo, N - q0, 32, part, 4, lane_);
  };
  // What an item needs before its first tile: its first two Q / dO tiles and the wave's K and V tiles, all by LDS-DMA (the K / V row fragments of the S and dP products are
  // re-read from the tiles in every key phase: 8 LDS reads against 32 registers that the kernel does not have -- with the fragments in registers the instantiation for 7 key
  // tiles sat at 251 registers and every value added to the tile loop spilled), and the rows of dO and O for D = rowsum(dO * O) plus lse in registers (d_issue).  Requested
  // for the NEXT item right behind the current item's last barrier -- every reader of the ring, of the K / V tiles, of lse2 / D has passed it, and the store staging lives in
  // the dS buffers -- so the requests travel while dK / dV / dQ are written out: per item, one memory round trip less in the open.
  // (lane_ / tid_: the caller's copy of the lane / thread index.  Behind an item's tile loop it is a LAUNDERED copy -- an empty asm makes it opaque -- so that the
  //  lane-constant source and destination addresses of the item-level requests and stores are recomputed per item instead of being hoisted to the kernel's entry, kept
  //  across the tile loop and spilled: each reload of such an address carries an s_waitcnt vmcnt(0) that also waits for the DMAs just requested.)
  auto request_item = [&](long off, long offo, int lane_) {
    request_tile(0, 0, off, offo, lane_);
    if (nt > 1) request_tile(1, 1, off, offo, lane_);
    if (keyw) {
      as_dma_rows_raw(Kt, k + off + (long)(w * 32) * ld, ld, N - w * 32, 32, 0, 1, lane_);
      as_dma_rows_raw(Vt, v + off + (long)(w * 32) * ld, ld, N - w * 32, 32, 0, 1, lane_);
    }
  };
  // D = rowsum(dO * O) on the rounded tensors, straight from global memory: 8 lanes per row
  u32x4 d_a[NPC], d_c[NPC];
  float lse_r = 0.f;
  auto d_issue = [&](long offo, long lrow, int tid_) {
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid_ + 512 * p, row = id >> 3, cp = id & 7;
      d_a[p] = (u32x4){0u, 0u, 0u, 0u}; d_c[p] = (u32x4){0u, 0u, 0u, 0u};
      if (row < N && !(dbg & 4)) { d_a[p] = *(const u32x4*)(dout + offo + (long)row * ldo + cp * 8); d_c[p] = *(const u32x4*)(o + offo + (long)row * ldo + cp * 8); }
    }
    lse_r = tid_ < N ? lse[lrow + tid_] : 0.f;                        // (NP <= 256 < 512 threads)
  };
  auto d_finish = [&]() {
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 512 * p, row = id >> 3, cp = id & 7;
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { d = fmaf(op_lo<OF>(d_a[p][e]), op_lo<OF>(d_c[p][e]), d); d = fmaf(op_hi<OF>(d_a[p][e]), op_hi<OF>(d_c[p][e]), d); }
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
      if (row < NP && cp == 0) Dv[row] = row < N ? d : 0.f;
    }
    if (tid < NP) lse2[tid] = lse_r * VDK_LOG2E;
  };
  int pb = -1, ph = 0;                                                // the item whose column-sum partials wait in CSs
  auto cs_flush = [&]() {
    if (tid < 192) {
      const int which = tid >> 6, col = tid & 63, nw = which == 0 ? 8 : nt;
      float t = 0.f;
      for (int ww = 0; ww < nw; ++ww) t += CSs[(which * 8 + ww) * 64 + col];
      cspart[(((long)pb * 3 + which) * H + ph) * 64 + col] = t;
    }
  };
  unsigned char* const St = Ds + w * 4096;                            // this wave's 4 KB store tile at the end of an item (the dS buffers are free behind the last barrier)
  if ((int)blockIdx.x < nitems) {
    const int b0 = blockIdx.x / H, h0 = blockIdx.x - b0 * H;
    request_item((long)b0 * N * ld + h0 * 64, (long)b0 * N * ldo + h0 * 64, lane);
    d_issue((long)b0 * N * ldo + h0 * 64, ((long)b0 * H + h0) * N, tid);
  }
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    // (no barrier here: lse2 / Dv were last read in front of the previous item's last barrier; its stores read the dS buffers and the dQ rows, which this item first
    //  writes behind the barrier below)
    d_finish();                                                       // this item's D and lse rows (requested with the item)
    f32x16 gk0 = as_zero16(), gk1 = as_zero16(), gv0 = as_zero16(), gv1 = as_zero16();
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this item's requests (and the previous item's stores)
    __syncthreads();
    if (cspart && pb >= 0) cs_flush();                                // (CSs is written again at the END of this item, several barriers from here)
    s16x8 kfr[4], vfr[4];                                             // the wave's K / V row fragments for the whole item, from its landed tiles
#if A5_KVREG
    if (keyw) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kfr[ks] = as_row_frag_l(Kt, al, ks); vfr[ks] = as_row_frag_l(Vt, al, ks); }
    }
#endif
    // One query tile = a key phase (S, dP, P, dS; dV / dK accumulate; dS to the exchange buffer) and a dQ phase (the wave's 16 x 16 block of dQ_j^T over all keys) with the
    // tile's one barrier between them.  Between barrier j and barrier j + 1 a wave owes the dQ phase of tile j and the key phase of tile j + 1, in either order (different dS
    // buffers, different ring buffers).  A5_SWAP: the first wave of every SIMD (waves 0..3) takes the key phase first, the second one (waves 4..7) the dQ phase, so the two
    // do not run the same pipe at the same time (released by the same barrier, both began with the 8 MFMAs of the key phase).
    auto key_phase = [&](int j) {
      const int q0 = j * 32, buf = j % 3, dbuf = j & 1;
      const unsigned char* Qs = QO + buf * 8192;
      const unsigned char* Os = Qs + 4096;
      unsigned char* const Dsj = Ds + dbuf * DSB;
      if (j + 2 < nt) request_tile(j + 2, (j + 2) % 3, off, offo, lane);    // (that buffer held tile j - 1: its readers passed the barrier that closed tile j - 1)
      if (keyw && !(dbg & 1)) {
        f32x16 st = as_zero16(), dp = as_zero16();
        f32x16 pv, ds;
        s16x8 pf[2], df[2];
        const bool edge = ragged && j == nt - 1;                      // rows of the last query tile beyond N (the DMA filled them with row N - 1): silenced
#if A5_KP == 0
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = vdk_mfma32<OF>(as_row_frag_l(Qs, al, ks), as_row_frag_l(Kt, al, ks), st);   // S[q][key]: lane = key, registers = queries
          dp = vdk_mfma32<OF>(as_row_frag_l(Os, al, ks), as_row_frag_l(Vt, al, ks), dp);   // dP[q][key]
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 lv = *(const f32x4*)(lse2 + q0 + 8 * g + 4 * hi);
          const f32x4 dd = *(const f32x4*)(Dv + q0 + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float p = fast_exp2(fmaf(st[r], scale2, -lv[e]));
            if (edge && q0 + 8 * g + 4 * hi + e >= N) p = 0.f;
            pv[r] = p;
            ds[r] = p * (dp[r] - dd[e]);
          }
        }
        as_pack_b<OF>(pv, pf);
        as_pack_b<OF>(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gv0 = vdk_mfma32<OF>(as_tr_frag_l(Os + 16 * s2 * AS_ROW, al, 0), pf[s2], gv0);     // dV^T[d][key] += dO^T P
          gv1 = vdk_mfma32<OF>(as_tr_frag_l(Os + 16 * s2 * AS_ROW, al, 1), pf[s2], gv1);
          gk0 = vdk_mfma32<OF>(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 0), df[s2], gk0);     // dK^T[d][key] += Q^T dS
          gk1 = vdk_mfma32<OF>(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 1), df[s2], gk1);
        }
#else
        // The same arithmetic in an order that gives every MFMA something to run beside.  Left to the scheduler the phase is 8 MFMAs, ~150 VALU instructions, 8 MFMAs,
        // and the two waves of a SIMD -- released by the same barrier -- do each part at the same time: matrix pipe against matrix pipe, then VALU against VALU
        // (measured: ~2600 cycles per key phase where the MFMAs of both waves are ~1000).  Here: the 4 S MFMAs, then each dP MFMA followed by the exponentials of 4
        // rows of S (independent of dP), each dV MFMA followed by 4 rows of dS, then the dK MFMAs; sched_barrier keeps the pieces in this order.
#define A5_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef VDK_EMU
#define A5_PIN(x) ((void)0)
#else
#define A5_PIN(x) asm volatile("" : "+v"(x))
#endif
        s16x8 qf[4], of_[4], kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = as_row_frag_l(Qs, al, ks); kf[ks] = A5_KVREG ? kfr[ks] : as_row_frag_l(Kt, al, ks); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st = vdk_mfma32<OF>(qf[ks], kf[ks], st);               // S[q][key]: lane = key, registers = queries
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { of_[ks] = as_row_frag_l(Os, al, ks); vf[ks] = A5_KVREG ? vfr[ks] : as_row_frag_l(Vt, al, ks); }
        A5_SB();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          dp = vdk_mfma32<OF>(of_[g], vf[g], dp);                                              // dP[q][key]
          const f32x4 lv = *(const f32x4*)(lse2 + q0 + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) pv[4 * g + e] = fast_exp2(fmaf(st[4 * g + e], scale2, -lv[e]));
          A5_SB();
        }
        if (edge) {                                                   // (wave-uniform)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (q0 + 8 * (r >> 2) + 4 * hi + (r & 3) >= N) pv[r] = 0.f;
        }
        as_pack_b<OF>(pv, pf);
        A5_SB();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dh = 0; dh < 2; ++dh) {
            const s16x8 tf = as_tr_frag_l(Os + 16 * s2 * AS_ROW, al, dh);
            if (dh == 0) gv0 = vdk_mfma32<OF>(tf, pf[s2], gv0); else gv1 = vdk_mfma32<OF>(tf, pf[s2], gv1);      // dV^T[d][key] += dO^T P
            const int g = 2 * s2 + dh;
            const f32x4 dd = *(const f32x4*)(Dv + q0 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ds[4 * g + e] = pv[4 * g + e] * (dp[4 * g + e] - dd[e]); A5_PIN(ds[4 * g + e]); }      // (pinned: left alone the optimiser sinks these behind the last MFMA)
            A5_SB();
          }
        as_pack_b<OF>(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gk0 = vdk_mfma32<OF>(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 0), df[s2], gk0);     // dK^T[d][key] += Q^T dS
          gk1 = vdk_mfma32<OF>(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 1), df[s2], gk1);
        }
#undef A5_SB
#undef A5_PIN
#endif
        // dS [q][key] (lane = key, registers = queries) -> the exchange buffer as [key][q] rows.  Lanes of keys beyond N hold dS of a duplicated key row: zero.
        const bool kval = w * 32 + l31 < N;
        unsigned char* const drow = Dsj + (w * 32 + l31) * A5_PITCH + 8 * hi;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          u32x4 u = *(const u32x4*)&df[s2];
          if (!kval) u = (u32x4){0u, 0u, 0u, 0u};
          *(u32x2*)(drow + 32 * s2) = (u32x2){u[0], u[1]};           // queries 16 s2 + 4 hi + 0..3
          *(u32x2*)(drow + 32 * s2 + 16) = (u32x2){u[2], u[3]};      // queries 16 s2 + 8 + 4 hi + 0..3
        }
      }
    };
    auto dq_phase = [&](int j) {
      const int q0 = j * 32, dbuf = j & 1;
      unsigned char* const Dsj = Ds + dbuf * DSB;
      // dQ_j^T block (16 d x 16 q) = sum over the key steps of K^T[16 d x 32 keys] . dS^T[32 keys x 16 q]
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (!(dbg & 2))
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        const unsigned char* bp = Dsj + t * 32 * A5_PITCH + ds_off;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(bp));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(bp + 4 * A5_PITCH));
        const s16x8 bfrag = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
        const s16x4 klo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Kall + t * 4096 + kt_off[0]));
        const s16x4 kup = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Kall + t * 4096 + kt_off[1]));
        acc = vdk_mfma16<OF>((s16x8){klo[0], klo[1], klo[2], klo[3], kup[0], kup[1], kup[2], kup[3]}, bfrag, acc);
      }
      // C layout: lane -> query column l & 15, rows d = 4 (l >> 4) + 0..3: four consecutive d of one query row = 8 bytes of its staged row
      *(u32x2*)(DQs + (q0 + 16 * qb + a16) * A5_QPITCH + 32 * db + 8 * g4) = (u32x2){pack_op2<OF>(acc[0] * scale, acc[1] * scale), pack_op2<OF>(acc[2] * scale, acc[3] * scale)};
    };
    key_phase(0);
#pragma unroll 1
    for (int j = 0; j < nt; ++j) {
      // tile j + 1 (requested one tile ago) must have landed before the barrier publishes it; tile j + 2's request -- this wave's newest, one instruction -- may still travel
      if (j + 2 < nt) __builtin_amdgcn_s_waitcnt(0x0F71); else __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(1) | vmcnt(0)
      __syncthreads();                                                // tile j's dS complete, tile j + 1's operands in place
      const bool keyfirst = A5_SWAP && w < 4;                         // (wave-uniform)
      if (keyfirst && j + 1 < nt) key_phase(j + 1);
      dq_phase(j);
      if (!keyfirst && j + 1 < nt) key_phase(j + 1);
    }
    __syncthreads();                                                  // every dQ block is staged; the last tile's dQ phase is done with the K tiles and the dS buffers
    int ln = lane, td = tid;
#ifndef VDK_EMU
    asm volatile("" : "+v"(ln), "+v"(td));                           // (laundered: see request_item)
#endif
    const int lane = ln, tid = td, l31 = ln & 31, hi = ln >> 5;       // (shadow the kernel-wide copies for the rest of the item)
    if (item + (int)gridDim.x < nitems) {
      const int nb = (item + (int)gridDim.x) / H, nh = (item + (int)gridDim.x) - nb * H;
      request_item((long)nb * N * ld + nh * 64, (long)nb * N * ldo + nh * 64, lane);
      d_issue((long)nb * N * ldo + nh * 64, ((long)nb * H + nh) * N, tid);
    }
    if (dbg & 8) continue;
    if (!cspart) {
      if (keyw) {
        as_store_tile<OF>(St, gk0, gk1, scale, dk + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
        as_store_tile<OF>(St, gv0, gv1, 1.0f, dv + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
      }
      // dQ rows leave as whole 128-byte rows: 8 threads per row
#pragma unroll
      for (int p = 0; p < NPC; ++p) {
        const int id = tid + 512 * p, row = id >> 3, cp = id & 7;
        if (row < N) *(u32x4*)(dq + (long)b * N * ldd + h * 64 + (long)row * ldd + cp * 8) = *(const u32x4*)(DQs + row * A5_QPITCH + cp * 16);
      }
      continue;
    }
    // the same stores, and on the way the column sums of what is stored: a lane adds up the 8 columns of its 16-byte chunk over its rows, the 8 lanes of a wave that own
    // the same chunk (lane & 7) are folded with three shuffles, lanes 0..7 then hold the wave's 64 sums; a last pass adds the waves up in wave order (deterministic)
    auto cs_add = [&](float (&cs)[8], const u32x4& v) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs[2 * e] += op_lo<OF>(v[e]); cs[2 * e + 1] += op_hi<OF>(v[e]); }
    };
    auto cs_put = [&](float (&cs)[8], int which) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { float t = cs[e]; t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32); cs[e] = t; }
      if (lane < 8) {
        float* dst = CSs + (which * 8 + w) * 64 + lane * 8;
        *(f32x4*)dst = (f32x4){cs[0], cs[1], cs[2], cs[3]}; *(f32x4*)(dst + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};
      }
    };
    if (keyw) {
#pragma unroll
      for (int which = 1; which <= 2; ++which) {
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bf16_t* const dst = (which == 1 ? dk : dv) + (long)b * N * ldd + h * 64;
        const f32x16& x0 = which == 1 ? gk0 : gv0; const f32x16& x1 = which == 1 ? gk1 : gv1;
        const float mul = which == 1 ? scale : 1.0f;
        const int row0 = w * 32;
        if (row0 + l31 < N) {      // (as_store_tile, with the rows read back for the sums)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = g ^ (l31 & 7), ch1 = (4 + g) ^ (l31 & 7);
            *(u32x2*)(St + l31 * AS_ROW + (ch << 4) + 8 * hi) = (u32x2){pack_op2<OF>(x0[4 * g] * mul, x0[4 * g + 1] * mul), pack_op2<OF>(x0[4 * g + 2] * mul, x0[4 * g + 3] * mul)};
            *(u32x2*)(St + l31 * AS_ROW + (ch1 << 4) + 8 * hi) = (u32x2){pack_op2<OF>(x1[4 * g] * mul, x1[4 * g + 1] * mul), pack_op2<OF>(x1[4 * g + 2] * mul, x1[4 * g + 3] * mul)};
          }
        }
        VDK_WAVE_LDS_SYNC();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = 8 * p + (lane >> 3), cp = lane & 7;
          if (row0 + r < N) {
            const u32x4 v = *(const u32x4*)(St + r * AS_ROW + ((cp ^ (r & 7)) << 4));
            *(u32x4*)(dst + (long)(row0 + r) * ldd + cp * 8) = v;
            cs_add(cs, v);
          }
        }
        VDK_WAVE_LDS_SYNC();
        cs_put(cs, which);
      }
    }
    {
      float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < NPC; ++p) {
        const int id = tid + 512 * p, row = id >> 3, cp = id & 7;
        if (row < N) {
          const u32x4 v = *(const u32x4*)(DQs + row * A5_QPITCH + cp * 16);
          *(u32x4*)(dq + (long)b * N * ldd + h * 64 + (long)row * ldd + cp * 8) = v;
          cs_add(cs, v);
        }
      }
      cs_put(cs, 0);
    }
    pb = b; ph = h;      // the waves' partials are added up behind the NEXT item's first barrier (or behind the loop): no barrier of its own
  }
  if (cspart && pb >= 0) { __syncthreads(); cs_flush(); }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
template <int NKT, int OF>
static int launch_fwd(const bf16_t* base, long D, long ld, bf16_t* o, long ldo, float* lse, int B, int N, int H, float scale, int grid, hipStream_t s) {
  const int R8 = (N + 7) & ~7;
  const size_t lds = (size_t)(2 * R8 + 32 * NKT) * AS_ROW;
  if (hipFuncSetAttribute((const void*)attn_s_fwd_kernel<NKT, OF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_fwd: LDS attribute");
  hipLaunchKernelGGL((attn_s_fwd_kernel<NKT, OF>), dim3((unsigned)grid), dim3(256), lds, s, base, base + D, base + 2 * D, ld, o, ldo, lse, N, H, scale, B * H);
  return VDK_OK;
}
// the caller's wish for the column-sum by-product of the next backward (form 5 only) and whether the launch produced it: vdk_attention_bwd_cs (attention.hip)
static thread_local float* t_cspart = nullptr;
static thread_local int t_cs_produced = 0;
void vdk_attention_small_want_colsum(float* cspart) { t_cspart = cspart; t_cs_produced = 0; }
int vdk_attention_small_colsum_produced() { return t_cs_produced; }
template <int NKT, int OF>
static int launch_bwd5(const bf16_t* base, long D, long ld, const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, bf16_t* dbase, long ldd, int B, int N, int H,
                       float scale, int grid, hipStream_t s) {
  const size_t lds = 3 * 8192 + 2 * (size_t)NKT * 4096 + 2 * (size_t)(32 * NKT) * A5_PITCH + (size_t)(32 * NKT) * A5_QPITCH + (size_t)NKT * 32 * 8 + 3 * 8 * 64 * 4;
  float* const cspart = t_cspart;
  if (cspart) t_cs_produced = 1;
  const char* edbg = getenv("VDK_ATTN5_DBG");
  const int dbg = edbg ? atoi(edbg) : 0;
  if (hipFuncSetAttribute((const void*)attn_s_bwd5_kernel<NKT, OF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
  hipLaunchKernelGGL((attn_s_bwd5_kernel<NKT, OF>), dim3((unsigned)grid), dim3(512), lds, s, base, base + D, base + 2 * D, ld, o, dout, ldo, lse, dbase, dbase + D, dbase + 2 * D,
                     ldd, N, H, scale, B * H, dbg, cspart);
  return VDK_OK;
}
static int grid_cap(int dflt) {
  if (const char* e = getenv("VDK_ATTN_GRID")) { const int v = atoi(e); if (v > 0) return v; }   // tests: force several items per workgroup
  return dflt;
}

// in-library entry points (attention.hip routes N <= 256 / N <= 224 here); return VDK_EUNSUPPORTED to let the caller fall back
int vdk_attention_small_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H, float scale, int opf, void* stream) {
  const int nkt = (N + 31) / 32;
  if (nkt < 1 || nkt > 8) return VDK_EUNSUPPORTED;
  const bf16_t* base = (const bf16_t*)qkv;
  const long D = (long)H * 64;
  int grid = B * H;
  const int cap = grid_cap(512);
  if (grid > cap) grid = cap;
  hipStream_t s = (hipStream_t)stream;
  switch (nkt) {
#define FW(n) case n: return opf ? launch_fwd<n, VDK_OPF_F16>(base, D, ld, (bf16_t*)o, ldo, lse, B, N, H, scale, grid, s) : launch_fwd<n, 0>(base, D, ld, (bf16_t*)o, ldo, lse, B, N, H, scale, grid, s);
    FW(1) FW(2) FW(3) FW(4) FW(5) FW(6) FW(7)
    default: return opf ? launch_fwd<8, VDK_OPF_F16>(base, D, ld, (bf16_t*)o, ldo, lse, B, N, H, scale, grid, s) : launch_fwd<8, 0>(base, D, ld, (bf16_t*)o, ldo, lse, B, N, H, scale, grid, s);
#undef FW
  }
}

int vdk_attention_small_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv, int64_t ldd, float* dvec, int32_t B, int32_t N,
                            int32_t H, float scale, int opf, void* stream) {
  const int nkt = (N + 31) / 32;
  if (nkt < 1 || nkt > 7) return VDK_EUNSUPPORTED;
  const bf16_t* base = (const bf16_t*)qkv;
  const long D = (long)H * 64;
  int grid = B * H;
  const int cap = grid_cap(256);
  if (grid > cap) grid = cap;
  hipStream_t s = (hipStream_t)stream;
  switch (nkt) {
#define B5(n) case n: return opf ? launch_bwd5<n, VDK_OPF_F16>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s) \
                                 : launch_bwd5<n, 0>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s);
    B5(1) B5(2) B5(3) B5(4) B5(5) B5(6)
    default: return opf ? launch_bwd5<7, VDK_OPF_F16>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s)
                        : launch_bwd5<7, 0>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s);
#undef B5
  }
}

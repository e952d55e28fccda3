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

// =====================================================================================  backward (fused)
// 512 threads, one workgroup per CU, persistent over (batch, head) items.  LDS (dynamic): Q [R8] | dO [R8] | zero rows up to 32*NKT of the dO array |
// 8 wave tiles of 4 KB (a wave's K tile, later its store tile) | 8 dS^T hand-off tiles of 2 KB | dQ f32 [32*NKT][64] | lse2 [32*NKT] | D [32*NKT].
// Register budget (two waves per SIMD = 256): the dK / dV accumulators (64) and the V fragments (16) stay resident; the K row fragments and the K^T
// fragments are re-read from the wave's K tile every step (12 LDS reads against 32 registers that had pushed lane-constant addresses into scratch).
// dQ accumulates in fp32 LDS rows (16-byte chunk ^ (q & 15): conflict-free ds_read_b128 / ds_write_b128 with lane = query); step 0 stores (every query tile
// is visited by exactly one wave per step), later steps read-add-write -- no zeroing pass.  (ds_add_f32 was measured and dropped: 164 cycles per
// wave-instruction on gfx950, 1.55 ms for the kernel against 0.3 ms with the explicit read-modify-write.)
// The next item's operands are requested as soon as the last step's barrier has passed (Q / dO by LDS-DMA, the K / V fragments and the O pieces for D into
// registers), so they travel while this item's dK / dV / dQ are written out; a one-workgroup-per-CU kernel has nobody else to hide that latency behind.

template <int NKT>
__global__ __launch_bounds__(512, 2) void attn_s_bwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                            const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long ldd, int N, int H, float scale,
                                                            int nitems, int dbg /* timing-only ablation mask (VDK_ATTN_DBG), 0 in production */) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R8 = (N + 7) & ~7;
  constexpr int NP = 32 * NKT;
  constexpr int NPC = (NP * 8 + 511) / 512;                           // 16-byte O pieces per thread for D = rowsum(dO * O)
  unsigned char* const Qs = smem;
  unsigned char* const Os = smem + R8 * AS_ROW;                       // dO rows, then zero rows up to NP
  unsigned char* const Wt = smem + (R8 + NP) * AS_ROW + w * 4096;     // this wave's 4 KB tile: its K rows during the steps, its store tile afterwards
  unsigned char* const Ht = smem + (R8 + NP) * AS_ROW + 8 * 4096 + w * 2048;   // this wave's 2 KB dS^T hand-off tile
  float* const dQt = (float*)(smem + (R8 + NP) * AS_ROW + 8 * 6144);  // f32 [NP][64]
  float* const lse2 = dQt + NP * 64;
  float* const Dv = lse2 + NP;
  for (int i = tid * 16; i < (NP - R8) * AS_ROW; i += 512 * 16) *(u32x4*)(Os + R8 * AS_ROW + i) = (u32x4){0u, 0u, 0u, 0u};
  const int nt = (N + 31) >> 5;                                       // query tiles == key tiles
  const bool act = w < nt;                                            // wave w owns key tile w
  const float scale2 = scale * VDK_LOG2E;
  const int krow = w * 32 + l31;
  const bool ragged = (N & 31) != 0;

  s16x8 vf[4];
  u32x4 opiece[NPC];
  // everything of item `it` that does not need this workgroup's wave tiles: Q / dO by DMA, K / V fragments, O pieces, lse
  auto request = [&](int it) {
    const int b = it / H, h = it - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    as_dma_rows(Qs, q + off, ld, N, R8, w, 8, lane);
    as_dma_rows(Os, dout + offo, ldo, N, R8, w, 8, lane);
    if (act) {
      const long kr = (long)(krow < N ? krow : N - 1) * ld;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) vf[ks] = *(const s16x8*)(v + off + kr + ks * 16 + hi * 8);
    }
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 512 * p, row = id >> 3;
      opiece[p] = (u32x4){0u, 0u, 0u, 0u};
      if (row < N && !(dbg & 64)) opiece[p] = *(const u32x4*)(o + offo + (long)row * ldo + (id & 7) * 8);
    }
    for (int i = tid; i < NP; i += 512) lse2[i] = i < N ? lse[((long)b * H + h) * N + i] * VDK_LOG2E : 0.f;
  };
  auto request_ktile = [&](int it) {                                  // own K tile: 32 rows = 4 DMA instructions into the wave tile
    if (!act) return;
    const int b = it / H, h = it - b * H;
    const long off = (long)b * N * ld + h * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = w * 32 + 8 * j + (lane >> 3), lrow = 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ as_f(lrow);
      const bf16_t* g = k + off + (long)(row < N ? row : N - 1) * ld + c * 8;
      __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(g), VDK_LDS_PTR(Wt + j * 1024), 16, 0, 0);
    }
  };

  __syncthreads();                                                    // the zero rows are written
  if ((int)blockIdx.x < nitems) { request(blockIdx.x); request_ktile(blockIdx.x); }
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this wave's DMAs (and the previous item's stores) are done
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 512 * p, row = id >> 3, cp = id & 7;
      float d = 0.f;
      if (row < NP) {
        const u32x4 a = *(const u32x4*)(Os + row * AS_ROW + ((cp ^ as_f(row)) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) { d = fmaf(bf_lo(a[e]), bf_lo(opiece[p][e]), d); d = fmaf(bf_hi(a[e]), bf_hi(opiece[p][e]), d); }
      }
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
      if (row < NP && cp == 0) Dv[row] = row < N ? d : 0.f;
    }
    __syncthreads();                                                  // D is complete
    f32x16 gk0 = as_zero16(), gk1 = as_zero16(), gv0 = as_zero16(), gv1 = as_zero16();
    for (int t = 0; t < nt; ++t) {
      if (act && !(dbg & 8)) {
        int qt = w + t; if (qt >= nt) qt -= nt;
        const int q0 = qt * 32;
        f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag(Qs, q0 + l31, ks, hi), as_row_frag(Wt, l31, ks, hi), st, 0, 0, 0);   // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag(Os, q0 + l31, ks, hi), vf[ks], dp, 0, 0, 0);   // dP[q][key]
        }
        f32x16 pv, ds;
        const bool edge = ragged && (qt == nt - 1 || w == nt - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 lv = *(const f32x4*)(lse2 + q0 + 8 * g + 4 * hi);
          const f32x4 dd = *(const f32x4*)(Dv + q0 + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float p = fast_exp2(fmaf(st[r], scale2, -lv[e]));
            if (edge && (q0 + 8 * g + 4 * hi + e >= N || krow >= N)) p = 0.f;
            pv[r] = p;
            ds[r] = p * (dp[r] - dd[e]);
          }
        }
        s16x8 pf[2], df[2];
        as_pack_b(pv, pf);
        as_pack_b(ds, df);
        // dS^T hand-off tile [key][q], 64-byte rows, 16-byte chunk ^ ((key >> 1) & 3): lane = key row, 4 consecutive queries per 8-byte store
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x4 u0 = *(const u32x4*)&df[g >> 1];
          const int e0 = (g & 1) * 2;
          *(u32x2*)(Ht + l31 * 64 + ((g ^ ((l31 >> 1) & 3)) << 4) + 8 * hi) = (u32x2){u0[e0], u0[e0 + 1]};
        }
        if (!(dbg & 4))
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          gv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Os, q0 + 16 * s, 0, lane), pf[s], gv0, 0, 0, 0);     // dV^T[d][key] += dO^T P
          gv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Os, q0 + 16 * s, 32, lane), pf[s], gv1, 0, 0, 0);
          gk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Qs, q0 + 16 * s, 0, lane), df[s], gk0, 0, 0, 0);     // dK^T[d][key] += Q^T dS
          gk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Qs, q0 + 16 * s, 32, lane), df[s], gk1, 0, 0, 0);
        }
        VDK_WAVE_LDS_SYNC();
        if (!(dbg & 2)) {                                             // dQ^T[d][q] partial of this key tile: K^T (A, from the K tile) x dS^T (B, from the hand-off tile)
          s16x8 bt[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            // B fragment: lane (q = l31, hi) <- keys 16s + 4hi + {0..3} and + 8 of column q
            const int sl = lane & 15, chalf = (lane >> 4) & 1;
            const int r1 = 16 * s + 4 * hi + (sl >> 2), r2 = r1 + 8;
            const int byte = 32 * chalf + 8 * (sl & 3);
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Ht + r1 * 64 + (((byte >> 4) ^ ((r1 >> 1) & 3)) << 4) + (byte & 8)));
            s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Ht + r2 * 64 + (((byte >> 4) ^ ((r2 >> 1) & 3)) << 4) + (byte & 8)));
            bt[s] = (s16x8){lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
          }
          unsigned char* const drow = (unsigned char*)dQt + (q0 + l31) * 256;   // dQ f32 [q][64], 16-byte chunk ^ (q & 15)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            f32x16 a = as_zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Wt, 16 * s, 32 * half, lane), bt[s], a, 0, 0, 0);
            if (!(dbg & 1)) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {                           // registers 4g .. 4g+3 = d 8g + 4hi + {0..3} (+ 32 * half): one 16-byte chunk
                f32x4* const pp = (f32x4*)(drow + (((8 * half + 2 * g + hi) ^ (l31 & 15)) << 4));
                f32x4 x = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                if (t != 0) x += *pp;                                 // step 0 stores: every query tile is visited by exactly one wave per step, no zeroing pass
                *pp = x;
              }
            }
          }
        }
        VDK_WAVE_LDS_SYNC();                                          // the hand-off tile may be rewritten in the next step
      }
      __syncthreads();                                                // step boundary: the dQ tiles change hands
    }
    // ---- the next item's operands start travelling now (Q / dO arrays, lse2 and the fragment registers are free) ----------------------------
    const int nxt = item + gridDim.x;
    if (nxt < nitems) request(nxt);
    // ---- outputs: dK, dV from registers and dQ from the d-major LDS tile, all through the wave tile -> 128-byte rows ---------------------------
    if (act && !(dbg & 32)) {
      as_store_tile(Wt, gk0, gk1, scale, dk + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
      as_store_tile(Wt, gv0, gv1, 1.0f, dv + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
    }
    if (!(dbg & 16))
      for (int id = tid; id < N * 8; id += 512) {
        const int row = id >> 3, cp = id & 7;                         // 8 columns: f32 chunks 2cp, 2cp+1
        const f32x4 x0 = *(const f32x4*)((const unsigned char*)dQt + row * 256 + (((2 * cp) ^ (row & 15)) << 4));
        const f32x4 x1 = *(const f32x4*)((const unsigned char*)dQt + row * 256 + (((2 * cp + 1) ^ (row & 15)) << 4));
        *(u32x4*)(dq + (long)b * N * ldd + h * 64 + (long)row * ldd + cp * 8) =
            (u32x4){pack_bf2(x0[0] * scale, x0[1] * scale), pack_bf2(x0[2] * scale, x0[3] * scale), pack_bf2(x1[0] * scale, x1[1] * scale), pack_bf2(x1[2] * scale, x1[3] * scale)};
      }
    if (nxt < nitems) request_ktile(nxt);                             // the wave tile is free again
  }
}

// =====================================================================================  backward (recompute form)
// Same contract as attn_s_bwd_kernel, different trade: 28 MFMAs per (query tile, key tile) pair instead of 20, and in exchange NO shared accumulator, no hand-off
// tile, no read-modify-write and no barrier inside an item's compute.  Q, K, V, dO of the item are all LDS-resident (by LDS-DMA); a wave first owns KEY tile w --
// S = Q K^T and dP = dO V^T per query tile, dV^T += dO^T P, dK^T += Q^T dS in its registers -- and then QUERY tile w -- S^T = K Q^T and dP^T = V dO^T per key tile
// (the transposed products put the query on the lane, which is what the B operand of dQ^T += K^T dS^T needs), dQ^T in its registers.  MFMA time was ~40 us of the fused
// kernel's 340 us; what it spent was the dQ traffic in LDS (68 us) and one workgroup barrier per step.
// LDS (dynamic): Q [R8] | K [R8] | V [R8] | dO [R8] | zero rows up to 32*NKT of the dO array | 8 wave store tiles of 4 KB | lse2 [32*NKT] | D [32*NKT].
template <int NKT>
__global__ __launch_bounds__(512, 2) void attn_s_bwd2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                             const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long ldd, int N, int H, float scale,
                                                             int nitems) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R8 = (N + 7) & ~7;
  constexpr int NP = 32 * NKT;
  constexpr int NPC = (NP * 8 + 511) / 512;                           // 16-byte O pieces per thread for D = rowsum(dO * O)
  unsigned char* const Qs = smem;
  unsigned char* const Ks = smem + R8 * AS_ROW;
  unsigned char* const Vs = smem + 2 * R8 * AS_ROW;
  unsigned char* const Os = smem + 3 * R8 * AS_ROW;                   // dO rows, then zero rows up to NP
  unsigned char* const Wt = smem + (3 * R8 + NP) * AS_ROW + w * 4096;  // this wave's store tile
  float* const lse2 = (float*)(smem + (3 * R8 + NP) * AS_ROW + 8 * 4096);
  float* const Dv = lse2 + NP;
  for (int i = tid * 16; i < (NP - R8) * AS_ROW; i += 512 * 16) *(u32x4*)(Os + R8 * AS_ROW + i) = (u32x4){0u, 0u, 0u, 0u};
  const int nt = (N + 31) >> 5;                                       // query tiles == key tiles
  const bool act = w < nt;
  const float scale2 = scale * VDK_LOG2E;
  const bool ragged = (N & 31) != 0;
  u32x4 opiece[NPC];
  auto request = [&](int it) {
    const int b = it / H, h = it - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    as_dma_rows(Qs, q + off, ld, N, R8, w, 8, lane);
    as_dma_rows(Ks, k + off, ld, N, R8, w, 8, lane);
    as_dma_rows(Vs, v + off, ld, N, R8, w, 8, lane);
    as_dma_rows(Os, dout + offo, ldo, N, R8, w, 8, lane);
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 512 * p, row = id >> 3;
      opiece[p] = (u32x4){0u, 0u, 0u, 0u};
      if (row < N) opiece[p] = *(const u32x4*)(o + offo + (long)row * ldo + (id & 7) * 8);
    }
    for (int i = tid; i < NP; i += 512) lse2[i] = i < N ? lse[((long)b * H + h) * N + i] * VDK_LOG2E : 0.f;
  };
  __syncthreads();                                                    // the zero rows are written
  if ((int)blockIdx.x < nitems) request(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this wave's DMAs (and the previous item's stores) are done
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 512 * p, row = id >> 3, cp = id & 7;
      float d = 0.f;
      if (row < NP) {
        const u32x4 a = *(const u32x4*)(Os + row * AS_ROW + ((cp ^ as_f(row)) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) { d = fmaf(bf_lo(a[e]), bf_lo(opiece[p][e]), d); d = fmaf(bf_hi(a[e]), bf_hi(opiece[p][e]), d); }
      }
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
      if (row < NP && cp == 0) Dv[row] = row < N ? d : 0.f;
    }
    __syncthreads();                                                  // D is complete
    if (act) {
      // ---- phase A: key tile w -------------------------------------------------------------------------------------------------------------
      {
        f32x16 gk0 = as_zero16(), gk1 = as_zero16(), gv0 = as_zero16(), gv1 = as_zero16();
        const int krow = w * 32 + l31;
        s16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = as_row_frag(Ks, krow, ks, hi); vf[ks] = as_row_frag(Vs, krow, ks, hi); }
        for (int qt = 0; qt < nt; ++qt) {
          const int q0 = qt * 32;
          f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag(Qs, q0 + l31, ks, hi), kf[ks], st, 0, 0, 0);   // S[q][key]: lane = key, registers = queries
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag(Os, q0 + l31, ks, hi), vf[ks], dp, 0, 0, 0);   // dP[q][key]
          }
          f32x16 pv, ds;
          const bool edge = ragged && (qt == nt - 1 || w == nt - 1);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 lv = *(const f32x4*)(lse2 + q0 + 8 * g + 4 * hi);
            const f32x4 dd = *(const f32x4*)(Dv + q0 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e;
              float p = fast_exp2(fmaf(st[r], scale2, -lv[e]));
              if (edge && (q0 + 8 * g + 4 * hi + e >= N || krow >= N)) p = 0.f;
              pv[r] = p;
              ds[r] = p * (dp[r] - dd[e]);
            }
          }
          s16x8 pf[2], df[2];
          as_pack_b(pv, pf);
          as_pack_b(ds, df);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            gv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Os, q0 + 16 * s, 0, lane), pf[s], gv0, 0, 0, 0);     // dV^T[d][key] += dO^T P
            gv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Os, q0 + 16 * s, 32, lane), pf[s], gv1, 0, 0, 0);
            gk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Qs, q0 + 16 * s, 0, lane), df[s], gk0, 0, 0, 0);     // dK^T[d][key] += Q^T dS
            gk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Qs, q0 + 16 * s, 32, lane), df[s], gk1, 0, 0, 0);
          }
        }
        // the key tile's gradients leave now (wave-private store tile): their 64 accumulators are free for phase B
        as_store_tile(Wt, gk0, gk1, scale, dk + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
        as_store_tile(Wt, gv0, gv1, 1.0f, dv + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
      }
      // ---- phase B: query tile w (transposed products: lane = query) ---------------------------------------------------------------------------
      {
        f32x16 gq0 = as_zero16(), gq1 = as_zero16();
        const int qrow = w * 32 + l31;
        s16x8 qf[4], gf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = as_row_frag(Qs, qrow, ks, hi); gf[ks] = as_row_frag(Os, qrow, ks, hi); }
        const float lq = lse2[qrow], dq_ = Dv[qrow];
        for (int kt = 0; kt < nt; ++kt) {
          const int k0 = kt * 32;
          f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag(Ks, k0 + l31, ks, hi), qf[ks], st, 0, 0, 0);   // S^T[key][q]: lane = query, registers = keys
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag(Vs, k0 + l31, ks, hi), gf[ks], dp, 0, 0, 0);   // dP^T[key][q]
          }
          f32x16 ds;
          const bool edge = ragged && (kt == nt - 1 || w == nt - 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float p = fast_exp2(fmaf(st[r], scale2, -lq));
            if (edge && (k0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= N || qrow >= N)) p = 0.f;
            ds[r] = p * (dp[r] - dq_);
          }
          s16x8 df[2];
          as_pack_b(ds, df);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            gq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Ks, k0 + 16 * s, 0, lane), df[s], gq0, 0, 0, 0);     // dQ^T[d][q] += K^T dS^T
            gq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag(Ks, k0 + 16 * s, 32, lane), df[s], gq1, 0, 0, 0);
          }
        }
        as_store_tile(Wt, gq0, gq1, scale, dq + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
      }
    }
    __syncthreads();                                                  // every wave is done with the item's arrays
    const int nxt = item + gridDim.x;
    if (nxt < nitems) request(nxt);
  }
}

// =====================================================================================  backward (split recompute form: two kernels, two workgroups per CU)
// The recompute form's two phases as two kernels of 256 threads whose LDS footprint (two operand arrays + 4 store tiles: 72 KB at N = 197) lets TWO workgroups share a
// CU, like the forward: one workgroup's operand loads and gradient stores overlap the other's MFMAs, which the single 139 KB workgroup of the one-kernel forms cannot do
// (its load, compute and store phases are serial: 325-360 us against a 125 us HBM floor).  Q, K, V, dO are read twice (once per kernel).
//   kv kernel: Q, dO LDS-resident; a wave takes key tiles w, w+4, ...: K / V fragments straight from global, dK^T / dV^T in registers.  It also computes
//              D = rowsum(dO * O) (it holds dO) and writes it to `dvec` for the q kernel.
//   q kernel:  K, V LDS-resident; a wave takes query tiles w, w+4, ...: Q / dO fragments straight from global, transposed products (lane = query), dQ^T in registers.
template <int NKT, int OF = 0>
__global__ __launch_bounds__(256, 2) void attn_s_bwd_kv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                               const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                               float* __restrict__ dvec, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long ldd, int N, int H, float scale,
                                                               int nitems) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R8 = (N + 7) & ~7;
  constexpr int NP = 32 * NKT;
  constexpr int NPC = (NP * 8 + 255) / 256;
  unsigned char* const Qs = smem;
  unsigned char* const Os = smem + R8 * AS_ROW;                       // dO rows, then zero rows up to NP
  unsigned char* const Wt = smem + (R8 + NP) * AS_ROW + w * 4096;
  float* const lse2 = (float*)(smem + (R8 + NP) * AS_ROW + 4 * 4096);
  float* const Dv = lse2 + NP;
  for (int i = tid * 16; i < (NP - R8) * AS_ROW; i += 256 * 16) *(u32x4*)(Os + R8 * AS_ROW + i) = (u32x4){0u, 0u, 0u, 0u};
  const int nt = (N + 31) >> 5;
  const float scale2 = scale * VDK_LOG2E;
  const bool ragged = (N & 31) != 0;
  const AsLane al = as_lane(lane);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    __syncthreads();                                                  // the previous item's readers are done (first pass: the zero rows are written)
    as_dma_rows(Qs, q + off, ld, N, R8, w, 4, lane);
    as_dma_rows(Os, dout + offo, ldo, N, R8, w, 4, lane);
    u32x4 opiece[NPC];
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 256 * p, row = id >> 3;
      opiece[p] = (u32x4){0u, 0u, 0u, 0u};
      if (row < N) opiece[p] = *(const u32x4*)(o + offo + (long)row * ldo + (id & 7) * 8);
    }
    for (int i = tid; i < NP; i += 256) lse2[i] = i < N ? lse[((long)b * H + h) * N + i] * VDK_LOG2E : 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 256 * p, row = id >> 3, cp = id & 7;
      float d = 0.f;
      if (row < NP) {
        const u32x4 a = *(const u32x4*)(Os + row * AS_ROW + ((cp ^ as_f(row)) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) { d = fmaf(op_lo<OF>(a[e]), op_lo<OF>(opiece[p][e]), d); d = fmaf(op_hi<OF>(a[e]), op_hi<OF>(opiece[p][e]), d); }
      }
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
      if (row < NP && cp == 0) { Dv[row] = row < N ? d : 0.f; if (row < N) dvec[((long)b * H + h) * N + row] = d; }
    }
    __syncthreads();                                                  // D is complete
    for (int kt = w; kt < nt; kt += 4) {
      f32x16 gk0 = as_zero16(), gk1 = as_zero16(), gv0 = as_zero16(), gv1 = as_zero16();
      const int krow = kt * 32 + l31;
      const long kr = (long)(krow < N ? krow : N - 1) * ld;
      s16x8 kf[4], vf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const s16x8*)(k + off + kr + ks * 16 + hi * 8); vf[ks] = *(const s16x8*)(v + off + kr + ks * 16 + hi * 8); }
#pragma unroll
      for (int qt = 0; qt < NKT; ++qt) {                              // (the launcher instantiates NKT == number of tiles)
        const int q0 = qt * 32;
        f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = vdk_mfma32<OF>(as_row_frag_l(Qs + q0 * AS_ROW, al, ks), kf[ks], st);   // S[q][key]: lane = key, registers = queries
          dp = vdk_mfma32<OF>(as_row_frag_l(Os + q0 * AS_ROW, al, ks), vf[ks], dp);   // dP[q][key]
        }
        f32x16 pv, ds;
        // Masking: only rows of the last query tile beyond N must be silenced (they would add into valid sums).  Lanes of keys beyond N need nothing: a lane is a column
        // of S, dP, dV^T and dK^T, whatever it holds stays in its own column, and those columns are never stored.
        const bool edge = ragged && qt == NKT - 1;
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
        s16x8 pf[2], df[2];
        as_pack_b<OF>(pv, pf);
        as_pack_b<OF>(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gv0 = vdk_mfma32<OF>(as_tr_frag_l(Os + (q0 + 16 * s2) * AS_ROW, al, 0), pf[s2], gv0);     // dV^T[d][key] += dO^T P
          gv1 = vdk_mfma32<OF>(as_tr_frag_l(Os + (q0 + 16 * s2) * AS_ROW, al, 1), pf[s2], gv1);
          gk0 = vdk_mfma32<OF>(as_tr_frag_l(Qs + (q0 + 16 * s2) * AS_ROW, al, 0), df[s2], gk0);     // dK^T[d][key] += Q^T dS
          gk1 = vdk_mfma32<OF>(as_tr_frag_l(Qs + (q0 + 16 * s2) * AS_ROW, al, 1), df[s2], gk1);
        }
      }
      as_store_tile<OF>(Wt, gk0, gk1, scale, dk + (long)b * N * ldd + h * 64, ldd, kt * 32, N, lane);
      as_store_tile<OF>(Wt, gv0, gv1, 1.0f, dv + (long)b * N * ldd + h * 64, ldd, kt * 32, N, lane);
    }
  }
}

template <int NKT, int OF = 0>
__global__ __launch_bounds__(256, 2) void attn_s_bwd_q_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                              const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse, const float* __restrict__ dvec,
                                                              bf16_t* __restrict__ dq, long ldd, int N, int H, float scale, int nitems) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R8 = (N + 7) & ~7;
  constexpr int NP = 32 * NKT;
  unsigned char* const Ks = smem;
  unsigned char* const Vs = smem + R8 * AS_ROW;                       // V rows, then zero rows up to NP
  unsigned char* const Wt = smem + (R8 + NP) * AS_ROW + w * 4096;
  for (int i = tid * 16; i < (NP - R8) * AS_ROW; i += 256 * 16) *(u32x4*)(Vs + R8 * AS_ROW + i) = (u32x4){0u, 0u, 0u, 0u};
  const int nt = (N + 31) >> 5;
  const float scale2 = scale * VDK_LOG2E;
  const bool ragged = (N & 31) != 0;
  const AsLane al = as_lane(lane);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    __syncthreads();
    as_dma_rows(Ks, k + off, ld, N, R8, w, 4, lane);
    as_dma_rows(Vs, v + off, ld, N, R8, w, 4, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int qt = w; qt < nt; qt += 4) {
      f32x16 gq0 = as_zero16(), gq1 = as_zero16();
      const int qrow = qt * 32 + l31;
      const int qr = qrow < N ? qrow : N - 1;
      s16x8 qf[4], gf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { qf[ks] = *(const s16x8*)(q + off + (long)qr * ld + ks * 16 + hi * 8); gf[ks] = *(const s16x8*)(dout + offo + (long)qr * ldo + ks * 16 + hi * 8); }
      const float lq = lse[((long)b * H + h) * N + qr] * VDK_LOG2E, dq_ = dvec[((long)b * H + h) * N + qr];
      // (explicit software pipelining of the fragment reads -- next pair's row fragments and this pair's transposed fragments requested before the exponentials -- was
      //  measured: no gain here, and in the kv kernel it cost 80 spilled registers: 388 vs 309 us.  The per-item fixed costs dominate: operand loads, D, barriers, stores.)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int k0 = kt * 32;
        f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = vdk_mfma32<OF>(as_row_frag_l(Ks + k0 * AS_ROW, al, ks), qf[ks], st);   // S^T[key][q]: lane = query, registers = keys
          dp = vdk_mfma32<OF>(as_row_frag_l(Vs + k0 * AS_ROW, al, ks), gf[ks], dp);   // dP^T[key][q]
        }
        f32x16 ds;
        const bool edge = ragged && kt == NKT - 1;                     // keys beyond N in the last key tile; a lane (= query) beyond N only spoils its own, unstored column
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = fast_exp2(fmaf(st[r], scale2, -lq));
          if (edge && k0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) p = 0.f;
          ds[r] = p * (dp[r] - dq_);
        }
        s16x8 df[2];
        as_pack_b<OF>(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gq0 = vdk_mfma32<OF>(as_tr_frag_l(Ks + (k0 + 16 * s2) * AS_ROW, al, 0), df[s2], gq0);     // dQ^T[d][q] += K^T dS^T
          gq1 = vdk_mfma32<OF>(as_tr_frag_l(Ks + (k0 + 16 * s2) * AS_ROW, al, 1), df[s2], gq1);
        }
      }
      as_store_tile<OF>(Wt, gq0, gq1, scale, dq + (long)b * N * ldd + h * 64, ldd, qt * 32, N, lane);
    }
  }
}

// =====================================================================================  backward (one pass: 5 GEMM-equivalents, no recomputation)  [form 4]
// BUILT AND PARITY-TESTED ON THE EMULATOR, NOT YET MEASURED ON THE MI355X (the round's GPU minutes were spent): opt-in through VDK_ATTN_BWD_FORM=4.
// A workgroup of 8 waves per (batch, head); wave w < nt owns KEY tile w for the whole item (its K / V row fragments stay in registers, its K tile also lies in LDS for the
// transposing reads), wave 7 is the dQ reducer.  The query tiles go by in the OUTER loop (Q_j / dO_j arrive by DMA into a double-buffered 2 x 8 KB), so that
//   * S = Q_j K^T and dP = dO_j V^T are computed once, with the key on the lane, and dV^T / dK^T accumulate in registers straight from the packed P / dS (the kv kernel's
//     arithmetic, bit for bit);
//   * dQ_j -- the one product that contracts over the key, i.e. over lanes and over WAVES -- is formed per wave from its 32 x 32 dS (transposed through 2 KB of LDS) as
//     a partial dQ_j^T tile in the C layout (8 KB fp32), handed to the reducer through the wave's slot and summed there in wave order (deterministic), one rounding at the
//     store.  The reducer works on tile j while the key waves are already in tile j + 1; two barriers per query tile.
// LDS: 16 KB (Q_j, dO_j x 2) + 28 KB (K tiles) + 56 KB (partials) + 14 KB (dS transposes) + 32 KB (store tiles) + lse / D = 148 KB: one workgroup per CU, seven
// independent MFMA streams.
__device__ __forceinline__ s16x8 as_tr_frag64(const unsigned char* tile, int t1, int lane) {     // transposed fragment of a 64-byte-row tile (no swizzle): lane = column l31
  const int s = lane & 15, chalf = (lane >> 4) & 1, hi = lane >> 5;
  const unsigned char* p1 = tile + (t1 + 4 * hi + (s >> 2)) * 64 + 32 * chalf + 8 * (s & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1));
  s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1 + 8 * 64));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return r;
}
template <int NKT>
__global__ __launch_bounds__(512) void attn_s_bwd1p_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                           const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long ldd, int N, int H, float scale,
                                                           int nitems) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NP = 32 * NKT;
  constexpr int NPC = (NP * 8 + 511) / 512;
  unsigned char* const QO = smem;                                     // [buffer 2][Q | dO][32 rows x 128 B]
  unsigned char* const Kt = smem + 16384 + w * 4096;                  // this wave's K tile (waves < nt)
  unsigned char* const Part = smem + 16384 + 7 * 4096;                // [7][8 KB] partial dQ^T tiles, C layout
  unsigned char* const DsT = Part + 7 * 8192 + w * 2048;              // this wave's dS [key][q] bf16, 64-byte rows
  unsigned char* const Wt = Part + 7 * 8192 + 7 * 2048 + w * 4096;    // this wave's store tile
  float* const lse2 = (float*)(Part + 7 * 8192 + 7 * 2048 + 8 * 4096);
  float* const Dv = lse2 + NP;
  const int nt = (N + 31) >> 5;
  const float scale2 = scale * VDK_LOG2E;
  const bool ragged = (N & 31) != 0;
  const AsLane al = as_lane(lane);
  const bool keyw = w < nt;                                           // (wave-uniform)
  // one DMA instruction per wave fills a quarter of one operand's 32-row tile: waves 0..3 Q, 4..7 dO
  auto request_tile = [&](int j, int buf, long off, long offo) {
    const int q0 = j * 32, part = w & 3;
    if (w < 4) as_dma_rows(QO + buf * 8192, q + off + (long)q0 * ld, ld, N - q0, 32, part, 4, lane);
    else as_dma_rows(QO + buf * 8192 + 4096, dout + offo + (long)q0 * ldo, ldo, N - q0, 32, part, 4, lane);
  };
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    __syncthreads();                                                  // the previous item is finished everywhere
    request_tile(0, 0, off, offo);
    s16x8 kf[4], vf[4];
    if (keyw) {
      as_dma_rows(Kt, k + off + (long)(w * 32) * ld, ld, N - w * 32, 32, 0, 1, lane);
      const int krow = w * 32 + l31;
      const long kr = (long)(krow < N ? krow : N - 1) * ld;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const s16x8*)(k + off + kr + ks * 16 + hi * 8); vf[ks] = *(const s16x8*)(v + off + kr + ks * 16 + hi * 8); }
    }
    // D = rowsum(dO * O) on the rounded tensors, straight from global memory: 8 lanes per row
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 512 * p, row = id >> 3, cp = id & 7;
      float d = 0.f;
      if (row < N) {
        const u32x4 a = *(const u32x4*)(dout + offo + (long)row * ldo + cp * 8), c = *(const u32x4*)(o + offo + (long)row * ldo + cp * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { d = fmaf(bf_lo(a[e]), bf_lo(c[e]), d); d = fmaf(bf_hi(a[e]), bf_hi(c[e]), d); }
      }
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
      if (row < NP && cp == 0) Dv[row] = row < N ? d : 0.f;
    }
    for (int i = tid; i < NP; i += 512) lse2[i] = i < N ? lse[((long)b * H + h) * N + i] * VDK_LOG2E : 0.f;
    f32x16 gk0 = as_zero16(), gk1 = as_zero16(), gv0 = as_zero16(), gv1 = as_zero16();
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
    __syncthreads();
    for (int j = 0; j < nt; ++j) {
      const int q0 = j * 32, buf = j & 1;
      const unsigned char* Qs = QO + buf * 8192;
      const unsigned char* Os = Qs + 4096;
      if (j + 1 < nt) request_tile(j + 1, buf ^ 1, off, offo);        // (that buffer's readers passed the barrier that closed tile j - 1)
      f32x16 pq0 = as_zero16(), pq1 = as_zero16();
      if (keyw) {
        f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag_l(Qs, al, ks), kf[ks], st, 0, 0, 0);   // S[q][key]: lane = key, registers = queries
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag_l(Os, al, ks), vf[ks], dp, 0, 0, 0);   // dP[q][key]
        }
        f32x16 pv, ds;
        const bool edge = ragged && j == nt - 1;                      // rows of the last query tile beyond N (the DMA filled them with row N - 1): silenced
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
        s16x8 pf[2], df[2];
        as_pack_b(pv, pf);
        as_pack_b(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Os + 16 * s2 * AS_ROW, al, 0), pf[s2], gv0, 0, 0, 0);     // dV^T[d][key] += dO^T P
          gv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Os + 16 * s2 * AS_ROW, al, 1), pf[s2], gv1, 0, 0, 0);
          gk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 0), df[s2], gk0, 0, 0, 0);     // dK^T[d][key] += Q^T dS
          gk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 1), df[s2], gk1, 0, 0, 0);
        }
        // dS [q][key] (lane = key, registers = queries) -> LDS as [key][q]; read back transposed: lane = query, slots = keys -- the B operand of dQ^T = K^T dS^T.
        // Lanes of keys beyond N hold dS of a duplicated key row: their contribution must not reach dQ.
        const bool kval = w * 32 + l31 < N;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          u32x4 u = *(const u32x4*)&df[s2];
          if (!kval) u = (u32x4){0u, 0u, 0u, 0u};
          *(u32x2*)(DsT + l31 * 64 + (8 * (2 * s2) + 4 * hi) * 2) = (u32x2){u[0], u[1]};
          *(u32x2*)(DsT + l31 * 64 + (8 * (2 * s2 + 1) + 4 * hi) * 2) = (u32x2){u[2], u[3]};
        }
        VDK_WAVE_LDS_SYNC();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const s16x8 dst = as_tr_frag64(DsT, 16 * s2, lane);
          pq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Kt + 16 * s2 * AS_ROW, al, 0), dst, pq0, 0, 0, 0);                 // dQ_j^T[d][q] (this key tile's share)
          pq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Kt + 16 * s2 * AS_ROW, al, 1), dst, pq1, 0, 0, 0);
        }
      }
      __syncthreads();                                                // the reducer has read tile j - 1's partials
      if (keyw) {
        float* slot = (float*)(Part + w * 8192) + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *(f32x4*)(slot + g * 256) = (f32x4){pq0[4 * g], pq0[4 * g + 1], pq0[4 * g + 2], pq0[4 * g + 3]};
          *(f32x4*)(slot + (4 + g) * 256) = (f32x4){pq1[4 * g], pq1[4 * g + 1], pq1[4 * g + 2], pq1[4 * g + 3]};
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);                             // the next tile's DMAs have landed
      __syncthreads();                                                // partials of tile j complete, tile j + 1's operands in place
      if (w == 7) {                                                   // (its pq0 / pq1 are still zero: they become the sum)
        f32x16& gq0 = pq0; f32x16& gq1 = pq1;
        for (int s = 0; s < nt; ++s) {                                // wave order: deterministic
          const float* slot = (const float*)(Part + s * 8192) + lane * 4;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 a = *(const f32x4*)(slot + g * 256), c = *(const f32x4*)(slot + (4 + g) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gq0[4 * g + e] += a[e]; gq1[4 * g + e] += c[e]; }
          }
        }
        as_store_tile(Wt, gq0, gq1, scale, dq + (long)b * N * ldd + h * 64, ldd, q0, N, lane);
      }
    }
    if (keyw) {
      as_store_tile(Wt, gk0, gk1, scale, dk + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
      as_store_tile(Wt, gv0, gv1, 1.0f, dv + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
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
// K^T fragments of the dQ role: re-read from the K tiles (KR = false, the default: 0 bytes of scratch) or held in 28 registers (KR = true: 256 VGPRs + spills, slower).
// LDS: 24 KB (Q_j, dO_j x 3) + 28 KB (K tiles, later the waves' store tiles) + 2 x 15.75 KB (dS) + 31.5 KB (dQ rows) + lse / D = 117 KB; 8 waves x 256 registers.
#define A5_PITCH 72                       /* bytes per key row of the dS exchange: 32 queries x 2 B + 8 (the 8-byte writes of 32 key lanes then hit 32 distinct bank pairs) */
#define A5_QPITCH 144                     /* bytes per query row of the dQ staging: 64 d x 2 B + 16 */
template <int NKT, int OF, bool KR>
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
  unsigned char* const Kall = smem + 3 * 8192;                        // [NKT][4 KB] K tiles; tile w doubles as wave w's store tile at the end of the item
  unsigned char* const Kt = Kall + w * 4096;
  unsigned char* const Ds = Kall + NKT * 4096;                        // [2][NP rows x A5_PITCH] dS [key][query]
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
  // one DMA instruction per wave fills a quarter of one operand's 32-row tile: waves 0..3 Q, 4..7 dO
  auto request_tile = [&](int j, int buf, long off, long offo) {
    const int q0 = j * 32, part = w & 3;
    if (w < 4) as_dma_rows(QO + buf * 8192, q + off + (long)q0 * ld, ld, N - q0, 32, part, 4, lane);
    else as_dma_rows(QO + buf * 8192 + 4096, dout + offo + (long)q0 * ldo, ldo, N - q0, 32, part, 4, lane);
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
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    __syncthreads();                                                  // the previous item is finished everywhere (its store tiles = this item's K tiles)
    if (cspart && pb >= 0) cs_flush();                                // (CSs is written again at the END of this item, several barriers from here)
    request_tile(0, 0, off, offo);
    if (nt > 1) request_tile(1, 1, off, offo);
    s16x8 kf[4], vf[4];
    if (keyw) {
      as_dma_rows(Kt, k + off + (long)(w * 32) * ld, ld, N - w * 32, 32, 0, 1, lane);
      const int krow = w * 32 + l31;
      const long kr = (long)(krow < N ? krow : N - 1) * ld;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const s16x8*)(k + off + kr + ks * 16 + hi * 8); vf[ks] = *(const s16x8*)(v + off + kr + ks * 16 + hi * 8); }
    }
    // D = rowsum(dO * O) on the rounded tensors, straight from global memory: 8 lanes per row (these loads share the prologue's one memory round trip)
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int id = tid + 512 * p, row = id >> 3, cp = id & 7;
      float d = 0.f;
      if (row < N && !(dbg & 4)) {
        const u32x4 a = *(const u32x4*)(dout + offo + (long)row * ldo + cp * 8), c = *(const u32x4*)(o + offo + (long)row * ldo + cp * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { d = fmaf(op_lo<OF>(a[e]), op_lo<OF>(c[e]), d); d = fmaf(op_hi<OF>(a[e]), op_hi<OF>(c[e]), d); }
      }
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
      if (row < NP && cp == 0) Dv[row] = row < N ? d : 0.f;
    }
    for (int i = tid; i < NP; i += 512) lse2[i] = i < N ? lse[((long)b * H + h) * N + i] * VDK_LOG2E : 0.f;
    f32x16 gk0 = as_zero16(), gk1 = as_zero16(), gv0 = as_zero16(), gv1 = as_zero16();
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
    __syncthreads();
    s16x8 ktf[KR ? NKT : 1];
    if (KR) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Kall + t * 4096 + kt_off[0]));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Kall + t * 4096 + kt_off[1]));
        ktf[t] = (s16x8){lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
      }
    }
#pragma unroll 1
    for (int j = 0; j < nt; ++j) {
      const int q0 = j * 32, buf = j % 3, dbuf = j & 1;
      const unsigned char* Qs = QO + buf * 8192;
      const unsigned char* Os = Qs + 4096;
      unsigned char* const Dsj = Ds + dbuf * DSB;
      if (j + 2 < nt) request_tile(j + 2, (j + 2) % 3, off, offo);    // (that buffer held tile j - 1: its readers passed the barrier that closed tile j - 1)
      if (keyw && !(dbg & 1)) {
        f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = vdk_mfma32<OF>(as_row_frag_l(Qs, al, ks), kf[ks], st);   // S[q][key]: lane = key, registers = queries
          dp = vdk_mfma32<OF>(as_row_frag_l(Os, al, ks), vf[ks], dp);   // dP[q][key]
        }
        f32x16 pv, ds;
        const bool edge = ragged && j == nt - 1;                      // rows of the last query tile beyond N (the DMA filled them with row N - 1): silenced
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
        s16x8 pf[2], df[2];
        as_pack_b<OF>(pv, pf);
        as_pack_b<OF>(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gv0 = vdk_mfma32<OF>(as_tr_frag_l(Os + 16 * s2 * AS_ROW, al, 0), pf[s2], gv0);     // dV^T[d][key] += dO^T P
          gv1 = vdk_mfma32<OF>(as_tr_frag_l(Os + 16 * s2 * AS_ROW, al, 1), pf[s2], gv1);
          gk0 = vdk_mfma32<OF>(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 0), df[s2], gk0);     // dK^T[d][key] += Q^T dS
          gk1 = vdk_mfma32<OF>(as_tr_frag_l(Qs + 16 * s2 * AS_ROW, al, 1), df[s2], gk1);
        }
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
      // tile j + 1 (requested one tile ago) must have landed before the barrier publishes it; tile j + 2's request -- this wave's newest, one instruction -- may still travel
      if (j + 2 < nt) __builtin_amdgcn_s_waitcnt(0x0F71); else __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(1) | vmcnt(0)
      __syncthreads();                                                // tile j's dS complete, tile j + 1's operands in place
      // dQ_j^T block (16 d x 16 q) = sum over the key steps of K^T[16 d x 32 keys] . dS^T[32 keys x 16 q]
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (!(dbg & 2))
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        const unsigned char* bp = Dsj + t * 32 * A5_PITCH + ds_off;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(bp));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(bp + 4 * A5_PITCH));
        const s16x8 bfrag = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
        if (KR) {
          acc = vdk_mfma16<OF>(ktf[t], bfrag, acc);
        } else {
          const s16x4 klo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Kall + t * 4096 + kt_off[0]));
          const s16x4 kup = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(Kall + t * 4096 + kt_off[1]));
          acc = vdk_mfma16<OF>((s16x8){klo[0], klo[1], klo[2], klo[3], kup[0], kup[1], kup[2], kup[3]}, bfrag, acc);
        }
      }
      // C layout: lane -> query column l & 15, rows d = 4 (l >> 4) + 0..3: four consecutive d of one query row = 8 bytes of its staged row
      *(u32x2*)(DQs + (q0 + 16 * qb + a16) * A5_QPITCH + 32 * db + 8 * g4) = (u32x2){pack_op2<OF>(acc[0] * scale, acc[1] * scale), pack_op2<OF>(acc[2] * scale, acc[3] * scale)};
    }
    __syncthreads();                                                  // every dQ block is staged; the last tile's dQ phase is done with the K tiles
    if (dbg & 8) continue;
    if (!cspart) {
      if (keyw) {      // (the K tile is free to stage the stores)
        as_store_tile<OF>(Kt, gk0, gk1, scale, dk + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
        as_store_tile<OF>(Kt, gv0, gv1, 1.0f, dv + (long)b * N * ldd + h * 64, ldd, w * 32, N, lane);
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
            *(u32x2*)(Kt + l31 * AS_ROW + (ch << 4) + 8 * hi) = (u32x2){pack_op2<OF>(x0[4 * g] * mul, x0[4 * g + 1] * mul), pack_op2<OF>(x0[4 * g + 2] * mul, x0[4 * g + 3] * mul)};
            *(u32x2*)(Kt + l31 * AS_ROW + (ch1 << 4) + 8 * hi) = (u32x2){pack_op2<OF>(x1[4 * g] * mul, x1[4 * g + 1] * mul), pack_op2<OF>(x1[4 * g + 2] * mul, x1[4 * g + 3] * mul)};
          }
        }
        VDK_WAVE_LDS_SYNC();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = 8 * p + (lane >> 3), cp = lane & 7;
          if (row0 + r < N) {
            const u32x4 v = *(const u32x4*)(Kt + r * AS_ROW + ((cp ^ (r & 7)) << 4));
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
template <int NKT>
static int launch_bwd(const bf16_t* base, long D, long ld, const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, bf16_t* dbase, long ldd, int B, int N, int H,
                      float scale, int grid, hipStream_t s) {
  const int R8 = (N + 7) & ~7, NP = 32 * NKT;
  const size_t lds = (size_t)(R8 + NP) * AS_ROW + 8 * 6144 + (size_t)NP * 256 + (size_t)NP * 8;   // arrays | wave tiles + hand-off tiles | dQ^T f32 | lse2, D
  if (lds > 160 * 1024) return VDK_EUNSUPPORTED;                                                  // N in 209 .. 224: the two-kernel backward of attention.hip
  if (hipFuncSetAttribute((const void*)attn_s_bwd_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
  int dbg = 0;
  if (const char* e = getenv("VDK_ATTN_DBG")) dbg = atoi(e);
  hipLaunchKernelGGL((attn_s_bwd_kernel<NKT>), dim3((unsigned)grid), dim3(512), lds, s, base, base + D, base + 2 * D, ld, o, dout, ldo, lse, dbase, dbase + D, dbase + 2 * D, ldd,
                     N, H, scale, B * H, dbg);
  return VDK_OK;
}
template <int NKT>
static int launch_bwd2(const bf16_t* base, long D, long ld, const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, bf16_t* dbase, long ldd, int B, int N, int H,
                       float scale, int grid, hipStream_t s) {
  const int R8 = (N + 7) & ~7, NP = 32 * NKT;
  const size_t lds = (size_t)(3 * R8 + NP) * AS_ROW + 8 * 4096 + (size_t)NP * 8;   // Q | K | V | dO + zero rows | wave store tiles | lse2, D
  if (lds > 160 * 1024) return VDK_EUNSUPPORTED;
  if (hipFuncSetAttribute((const void*)attn_s_bwd2_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
  hipLaunchKernelGGL((attn_s_bwd2_kernel<NKT>), dim3((unsigned)grid), dim3(512), lds, s, base, base + D, base + 2 * D, ld, o, dout, ldo, lse, dbase, dbase + D, dbase + 2 * D, ldd,
                     N, H, scale, B * H);
  return VDK_OK;
}
static int grid_cap3(int dflt);
template <int NKT, int OF>
static int launch_bwd3(const bf16_t* base, long D, long ld, const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, float* dvec, bf16_t* dbase, long ldd, int B, int N,
                       int H, float scale, hipStream_t s) {
  const int R8 = (N + 7) & ~7, NP = 32 * NKT;
  const size_t lds_kv = (size_t)(R8 + NP) * AS_ROW + 4 * 4096 + (size_t)NP * 8, lds_q = (size_t)(R8 + NP) * AS_ROW + 4 * 4096;
  if (lds_kv > 80 * 1024 || !dvec) return VDK_EUNSUPPORTED;             // two workgroups per CU are the point
  if (hipFuncSetAttribute((const void*)attn_s_bwd_kv_kernel<NKT, OF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv) != hipSuccess ||
      hipFuncSetAttribute((const void*)attn_s_bwd_q_kernel<NKT, OF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
  int grid = B * H;
  const int cap = grid_cap3(512);
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((attn_s_bwd_kv_kernel<NKT, OF>), dim3((unsigned)grid), dim3(256), lds_kv, s, base, base + D, base + 2 * D, ld, o, dout, ldo, lse, dvec, dbase + D, dbase + 2 * D, ldd,
                     N, H, scale, B * H);
  hipLaunchKernelGGL((attn_s_bwd_q_kernel<NKT, OF>), dim3((unsigned)grid), dim3(256), lds_q, s, base, base + D, base + 2 * D, ld, dout, ldo, lse, (const float*)dvec, dbase, ldd, N, H,
                     scale, B * H);
  return VDK_OK;
}
template <int NKT>
static int launch_bwd1p(const bf16_t* base, long D, long ld, const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, bf16_t* dbase, long ldd, int B, int N, int H,
                        float scale, int grid, hipStream_t s) {
  const size_t lds = 16384 + 7 * 4096 + 7 * 8192 + 7 * 2048 + 8 * 4096 + (size_t)NKT * 32 * 8;
  if (hipFuncSetAttribute((const void*)attn_s_bwd1p_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
  hipLaunchKernelGGL((attn_s_bwd1p_kernel<NKT>), dim3((unsigned)grid), dim3(512), lds, s, base, base + D, base + 2 * D, ld, o, dout, ldo, lse, dbase, dbase + D, dbase + 2 * D, ldd,
                     N, H, scale, B * H);
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
  const size_t lds = 3 * 8192 + (size_t)NKT * 4096 + 2 * (size_t)(32 * NKT) * A5_PITCH + (size_t)(32 * NKT) * A5_QPITCH + (size_t)NKT * 32 * 8 + 3 * 8 * 64 * 4;
  float* const cspart = t_cspart;
  if (cspart) t_cs_produced = 1;
  const char* edbg = getenv("VDK_ATTN5_DBG");
  const int dbg = edbg ? atoi(edbg) : 0;
  const char* ekt = getenv("VDK_ATTN5_KT");      // A/B: VDK_ATTN5_KT=regs keeps the K^T fragments of the dQ role in registers (read per launch; default: re-read from LDS)
  const bool kregs = ekt && ekt[0] == 'r';
  if (kregs) {
    if (hipFuncSetAttribute((const void*)attn_s_bwd5_kernel<NKT, OF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
    hipLaunchKernelGGL((attn_s_bwd5_kernel<NKT, OF, true>), dim3((unsigned)grid), dim3(512), lds, s, base, base + D, base + 2 * D, ld, o, dout, ldo, lse, dbase, dbase + D, dbase + 2 * D,
                       ldd, N, H, scale, B * H, dbg, cspart);
  } else {
    if (hipFuncSetAttribute((const void*)attn_s_bwd5_kernel<NKT, OF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
    hipLaunchKernelGGL((attn_s_bwd5_kernel<NKT, OF, false>), dim3((unsigned)grid), dim3(512), lds, s, base, base + D, base + 2 * D, ld, o, dout, ldo, lse, dbase, dbase + D, dbase + 2 * D,
                       ldd, N, H, scale, B * H, dbg, cspart);
  }
  return VDK_OK;
}
static thread_local int g_bwd_form = -1;        // -1: environment VDK_ATTN_BWD_FORM (default 5); 5 = one pass, dS exchanged between the waves, dQ split by output block; 4 = one pass with partial dQ tiles + a reducer wave (slower); 3 = split recompute form (two kernels, 2 workgroups / CU); 2 = one-kernel recompute form; 1 = fused form with the shared dQ tile
int vdk_attention_small_bwd_form(int form) { g_bwd_form = form; return VDK_OK; }

static int grid_cap(int dflt);
static int grid_cap3(int dflt) { return grid_cap(dflt); }
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
  int form = g_bwd_form;
  if (form < 0) { const char* e = getenv("VDK_ATTN_BWD_FORM"); form = e ? atoi(e) : 5; }      // default: the one-pass form with the dS exchange (measured 236-254 us against 272-313 us for the two kernels, profiles/r04_attention_ab.json)
  if (opf && form != 5) form = 3;          // fp16 operands: the two-kernel form or form 5 (the A/B forms 1 / 2 / 4 are bf16 only)
  if (form == 5) {
    switch (nkt) {
#define B5(n) case n: return opf ? launch_bwd5<n, VDK_OPF_F16>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s) \
                                 : launch_bwd5<n, 0>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s);
      B5(1) B5(2) B5(3) B5(4) B5(5) B5(6) B5(7)
#undef B5
    }
  }
  if (form == 4) {
    switch (nkt) {
#define B4(n) case n: return launch_bwd1p<n>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s);
      B4(1) B4(2) B4(3) B4(4) B4(5) B4(6) B4(7)
#undef B4
    }
  }
  if (form == 3) {
    int rc = VDK_EUNSUPPORTED;
    switch (nkt) {
#define B3(n) case n: rc = opf ? launch_bwd3<n, VDK_OPF_F16>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, dvec, (bf16_t*)dqkv, ldd, B, N, H, scale, s) \
                           : launch_bwd3<n, 0>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, dvec, (bf16_t*)dqkv, ldd, B, N, H, scale, s); break;
      B3(1) B3(2) B3(3) B3(4) B3(5) B3(6) B3(7)
#undef B3
    }
    if (rc != VDK_EUNSUPPORTED || opf) return rc;
    form = 2;
  }
#define BW(n) (form == 2 ? launch_bwd2<n>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s) \
                         : launch_bwd<n>(base, D, ld, (const bf16_t*)o, (const bf16_t*)dout, ldo, lse, (bf16_t*)dqkv, ldd, B, N, H, scale, grid, s))
  switch (nkt) {
    case 1: return BW(1);
    case 2: return BW(2);
    case 3: return BW(3);
    case 4: return BW(4);
    case 5: return BW(5);
    case 6: return BW(6);
    default: return BW(7);
  }
#undef BW
}

// attention.hip — K3: multi-head self-attention forward/backward for timm's `Attention`
// (vision_transformer.Attention: qkv Linear -> [B,N,3,H,hd] -> softmax(q k^T / sqrt(hd)) v; built by the
// reference at models/classifier/classify_model.py:49-54 / models/faceX/backbone/timm_wrapper.py:16-21).
// head_dim = 64 (ViT-B/16: 12x64, ViT-L: 16x64).  The N x N score matrix never leaves the CU.
//
// MFMA formulation (v_mfma_f32_32x32x16_bf16, swapped operands so that per-query state is per-LANE):
//   S^T[key][q]  = K . Q^T        A = K rows from LDS (row-major), B = Q fragments in registers
//   O^T[d][q]   += V^T . P^T      A = V^T read from the row-major V tile with ds_read_b64_tr_b16, B = P^T = the S^T accumulator itself:
//                                 C-layout (col = lane&31 = q, row = key(r, lane>>5)) IS a valid B-operand
//                                 layout when the contraction (key) order is permuted consistently on both
//                                 operands: k-slot (hi*8 + j) of step s <-> key 16s + 4hi + (j&3) + 8(j>>2).
// so softmax max/sum/rescale are lane-local (one shfl_xor 32 joins the two half-waves that share a query),
// and O is rescaled by a per-lane factor.  Backward: kernel 1 owns query tiles (dQ, D = rowsum(dO*O)),
// kernel 2 owns key tiles (dK, dV); both recompute P from the saved log-sum-exp.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "vdk_device.h"
#include "vdk_host.h"

#define A_HD 64
#define A_RP 72          // row-major LDS pitch (bf16): 64 + 8 -> 144 B rows, conflict-free ds_read_b128
#define A_FWD_KC 256     // keys staged per chunk (fwd, bwd-dq)
#define A_BWD_QC 256     // queries staged per chunk (bwd-dkv)

// ---- staging helpers (256 threads) ---------------------------------------------------------------
// dst[r][0..63] = src[(row0 + r) * ld + 0..63] for r < ntile, zero rows beyond nvalid
__device__ __forceinline__ void a_stage_rows(bf16_t* dst, const bf16_t* __restrict__ src, long ld, int row0, int nvalid_total,
                                             int ntile) {
  for (int id = threadIdx.x; id < ntile * 8; id += 256) {
    int r = id >> 3, ch = id & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row0 + r < nvalid_total) v = *(const u32x4*)(src + (long)(row0 + r) * ld + ch * 8);
    *(u32x4*)(dst + r * A_RP + ch * 8) = v;
  }
}
// B-operand fragments of a 32-row tile straight from global: lane (row = l&31, hi): src[row][ks*16 + hi*8 ..+7]
__device__ __forceinline__ void a_load_frags(s16x8 (&f)[4], const bf16_t* __restrict__ src, long ld, int row, int nvalid, int hi) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nvalid) v = *(const s16x8*)(src + (long)row * ld + ks * 16 + hi * 8);
    f[ks] = v;
  }
}
// acc[row r of tile][col] = sum_d tile[r][d] * frag[col][d]   (A = 32 LDS rows starting at `rows`, B = frags)
__device__ __forceinline__ f32x16 a_mma_rows(const bf16_t* rows /* &tile[l31][hi*8] */, const s16x8 (&f)[4]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    s16x8 a = *(const s16x8*)(rows + ks * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, f[ks], acc, 0, 0, 0);
  }
  return acc;
}
// pack the 16 accumulator values of a C-layout tile into the two B-operand fragments (permuted k order)
__device__ __forceinline__ void a_pack_b(const float (&p)[16], s16x8 (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 u = {pack_bf2(p[8 * s + 0], p[8 * s + 1]), pack_bf2(p[8 * s + 2], p[8 * s + 3]),
               pack_bf2(p[8 * s + 4], p[8 * s + 5]), pack_bf2(p[8 * s + 6], p[8 * s + 7])};
    f[s] = *(s16x8*)&u;
  }
}
// A-operand X^T[d][idx] read straight from the ROW-major tile X[idx][d] with ds_read_b64_tr_b16: lane (d = d0 + l31, hi),
// step s -> rows idx0 + 16s + 4hi + {0..3} and + 8 (the permuted contraction order of a_pack_b)
__device__ __forceinline__ s16x8 a_load_T(const bf16_t* tile /* row-major, pitch A_RP */, int idx0, int d0, int s, int hi, int lane) {
  const int t1 = idx0 + 16 * s + 4 * hi;
  return tr_frag8(tile, A_RP, t1, t1 + 8, d0, lane);
}

// =====================================================================================  forward
// q/k/v: bf16, element (b, n, h, d) at ptr[(b*N + n) * ld + h*64 + d].  o: same indexing with ldo.
// lse: float [B, H, N].  grid = B*H, block = 256.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, long ld, bf16_t* __restrict__ o, long ldo,
                                                       float* __restrict__ lse, int N, int H, float scale, int nitems) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[A_FWD_KC * A_RP];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[A_FWD_KC * A_RP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int nqt = (N + 31) / 32;
  const int nchunk = (N + A_FWD_KC - 1) / A_FWD_KC;
  const float scale2 = scale * VDK_LOG2E;
  // every wave runs the same number of q-tile rounds so that barriers stay uniform
  const int rounds = (nqt + 3) / 4;
  // One chunk of keys (N <= 256, ViT-B/16 at 224: 197): the workgroup is PERSISTENT over (batch, head) items and the next item's K / V rows are loaded
  // into registers before this item's q-tile rounds and written to LDS after them -- with two workgroups per CU (74 KB of LDS each) a
  // load -> barrier -> compute sequence per item left the HBM round trip exposed.  Longer sequences keep one item per workgroup and stage per chunk.
  const bool persistent = nchunk == 1;
  const int nkp1 = (N + 31) / 32 * 32;
  constexpr int PFN = A_FWD_KC * 8 / 256;                  // 16-byte pieces per thread and matrix
  u32x4 pk[PFN], pv[PFN];
  auto prefetch = [&](int item) {
    const bf16_t* kb2 = k + (long)(item / H) * N * ld + (item % H) * A_HD;
    const bf16_t* vb2 = v + (long)(item / H) * N * ld + (item % H) * A_HD;
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
      const int id = threadIdx.x + i * 256, r = id >> 3, ch = id & 7;
      u32x4 a = {0u, 0u, 0u, 0u}, c2 = {0u, 0u, 0u, 0u};
      if (id < nkp1 * 8 && r < N) { a = *(const u32x4*)(kb2 + (long)r * ld + ch * 8); c2 = *(const u32x4*)(vb2 + (long)r * ld + ch * 8); }
      pk[i] = a; pv[i] = c2;
    }
  };
  const int n_items = nitems;
  if (persistent && (int)blockIdx.x < n_items) prefetch(blockIdx.x);
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
  const int b = item / H, h = item % H;
  const bf16_t* qb = q + (long)b * N * ld + h * A_HD;
  const bf16_t* kb = k + (long)b * N * ld + h * A_HD;
  const bf16_t* vb = v + (long)b * N * ld + h * A_HD;
  if (persistent) {
    __syncthreads();                                        // the previous item's readers of Ks / Vs are done
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
      const int id = threadIdx.x + i * 256, r = id >> 3, ch = id & 7;
      if (id < nkp1 * 8) { *(u32x4*)(Ks + r * A_RP + ch * 8) = pk[i]; *(u32x4*)(Vs + r * A_RP + ch * 8) = pv[i]; }
    }
    __syncthreads();
    if (item + (int)gridDim.x < n_items) prefetch(item + gridDim.x);
  }
  for (int rd = 0; rd < rounds; ++rd) {
    const int qt = rd * 4 + w;
    const bool active = qt < nqt;
    const int qrow = qt * 32 + l31;
    s16x8 qf[4];
    a_load_frags(qf, qb, ld, active ? qrow : N, N, hi);
    float m = -INFINITY, l = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    for (int c = 0; c < nchunk; ++c) {
      const int key0 = c * A_FWD_KC;
      int nk = N - key0; if (nk > A_FWD_KC) nk = A_FWD_KC;
      const int nkp = (nk + 31) / 32 * 32;
      if (!persistent) {
        __syncthreads();  // previous readers of Ks/Vt are done
        a_stage_rows(Ks, kb, ld, key0, N, nkp);
        a_stage_rows(Vs, vb, ld, key0, N, nkp);
        __syncthreads();
      }
      if (active) {
        for (int kt = 0; kt < nkp / 32; ++kt) {
          f32x16 st = a_mma_rows(Ks + (kt * 32 + l31) * A_RP + hi * 8, qf);
          float p[16];
          float mt = -INFINITY;
          if (key0 + kt * 32 + 32 > N) {   // only the ragged last tile needs the key mask (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
              p[r] = (key < N) ? st[r] * scale2 : -INFINITY;   // scores in the log2 domain
              mt = fmaxf(mt, p[r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = st[r] * scale2; mt = fmaxf(mt, p[r]); }
          }
          mt = fmaxf(mt, __shfl_xor(mt, 32));
          // deferred rescale (guide T13): keep the old running max while the tile max exceeds it by < 2^8; P is then
          // bounded by 2^8 instead of 1 (harmless in bf16/fp32) and the 32-register O rescale is skipped almost always.
          if (__any(mt > m + 8.0f)) {
            const float mn = fmaxf(m, mt);
            const float alpha = fast_exp2(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
          }
          float ps = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { p[r] = fast_exp2(p[r] - m); ps += p[r]; }
          l += ps;
          s16x8 pf[2];
          a_pack_b(p, pf);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            s16x8 a0 = a_load_T(Vs, kt * 32, 0, s, hi, lane);
            s16x8 a1 = a_load_T(Vs, kt * 32, 32, s, hi, lane);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pf[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pf[s], o1, 0, 0, 0);
          }
        }
      }
    }
    // finish: both half-waves of a query share (m); their partial sums add up
    const float lt = l + __shfl_xor(l, 32);
    if (active && qrow < N) {
      const float inv = 1.0f / lt;
      bf16_t* orow = o + ((long)b * N + qrow) * ldo + h * A_HD;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * hi;
        *(u32x2*)(orow + d) = (u32x2){pack_bf2(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
        *(u32x2*)(orow + 32 + d) = (u32x2){pack_bf2(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
      }
      if (hi == 0 && lse) lse[((long)b * H + h) * N + qrow] = (m + log2f(lt)) * 0.6931471805599453f;  // natural-log LSE
    }
  }
  }   // items
}

// =====================================================================================  backward 1: dQ (+ D)
// wave owns a query tile, loops over all keys.  dq written with the q indexing (ld_dq), scaled by `scale`.
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v, long ld, const bf16_t* __restrict__ o,
                                                          const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dq, long lddq, float* __restrict__ dvec, int N, int H,
                                                          float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[A_FWD_KC * A_RP];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[A_FWD_KC * A_RP];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const bf16_t* qb = q + (long)b * N * ld + h * A_HD;
  const bf16_t* kb = k + (long)b * N * ld + h * A_HD;
  const bf16_t* vb = v + (long)b * N * ld + h * A_HD;
  const bf16_t* ob = o + (long)b * N * ldo + h * A_HD;
  const bf16_t* dob = dout + (long)b * N * ldo + h * A_HD;
  const int nqt = (N + 31) / 32, nchunk = (N + A_FWD_KC - 1) / A_FWD_KC, rounds = (nqt + 3) / 4;
  for (int rd = 0; rd < rounds; ++rd) {
    const int qt = rd * 4 + w;
    const bool active = qt < nqt;
    const int qrow = qt * 32 + l31;
    const int qsafe = active ? qrow : N;
    s16x8 qf[4], dof[4], of[4];
    a_load_frags(qf, qb, ld, qsafe, N, hi);
    a_load_frags(dof, dob, ldo, qsafe, N, hi);
    a_load_frags(of, ob, ldo, qsafe, N, hi);
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 a = *(u32x4*)&dof[ks], c = *(u32x4*)&of[ks];
#pragma unroll
      for (int e = 0; e < 4; ++e) { dsum = fmaf(bf_lo(a[e]), bf_lo(c[e]), dsum); dsum = fmaf(bf_hi(a[e]), bf_hi(c[e]), dsum); }
    }
    dsum += __shfl_xor(dsum, 32);
    const float mylse2 = ((active && qrow < N) ? lse[((long)b * H + h) * N + qrow] : 0.f) * VDK_LOG2E;
    const float scale2 = scale * VDK_LOG2E;
    if (active && qrow < N && hi == 0) dvec[((long)b * H + h) * N + qrow] = dsum;
    f32x16 g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
    for (int c = 0; c < nchunk; ++c) {
      const int key0 = c * A_FWD_KC;
      int nk = N - key0; if (nk > A_FWD_KC) nk = A_FWD_KC;
      const int nkp = (nk + 31) / 32 * 32;
      if (nchunk > 1 || rd == 0) {
        __syncthreads();
        a_stage_rows(Ks, kb, ld, key0, N, nkp);
        a_stage_rows(Vs, vb, ld, key0, N, nkp);
        __syncthreads();
      }
      if (active) {
        for (int kt = 0; kt < nkp / 32; ++kt) {
          f32x16 st = a_mma_rows(Ks + (kt * 32 + l31) * A_RP + hi * 8, qf);
          f32x16 dp = a_mma_rows(Vs + (kt * 32 + l31) * A_RP + hi * 8, dof);
          float ds[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float p = (key < N && qrow < N) ? fast_exp2(fmaf(st[r], scale2, -mylse2)) : 0.f;
            ds[r] = p * (dp[r] - dsum);
          }
          s16x8 df[2];
          a_pack_b(ds, df);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            s16x8 a0 = a_load_T(Ks, kt * 32, 0, s, hi, lane);
            s16x8 a1 = a_load_T(Ks, kt * 32, 32, s, hi, lane);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, df[s], g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, df[s], g1, 0, 0, 0);
          }
        }
      }
    }
    if (active && qrow < N) {
      bf16_t* drow = dq + ((long)b * N + qrow) * lddq + h * A_HD;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * hi;
        *(u32x2*)(drow + d) = (u32x2){pack_bf2(g0[4 * g] * scale, g0[4 * g + 1] * scale), pack_bf2(g0[4 * g + 2] * scale, g0[4 * g + 3] * scale)};
        *(u32x2*)(drow + 32 + d) = (u32x2){pack_bf2(g1[4 * g] * scale, g1[4 * g + 1] * scale), pack_bf2(g1[4 * g + 2] * scale, g1[4 * g + 3] * scale)};
      }
    }
  }
}

// =====================================================================================  backward 2: dK, dV
// wave owns a key tile, loops over all queries (staged in chunks of A_BWD_QC).
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, long ld, const bf16_t* __restrict__ dout,
                                                           long ldo, const float* __restrict__ lse, const float* __restrict__ dvec,
                                                           bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long lddkv, int N, int H,
                                                           float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[A_BWD_QC * A_RP];
  __shared__ __attribute__((aligned(16))) bf16_t Os[A_BWD_QC * A_RP];
  __shared__ __attribute__((aligned(16))) float lse_s[A_BWD_QC];
  __shared__ __attribute__((aligned(16))) float dv_s[A_BWD_QC];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const bf16_t* qb = q + (long)b * N * ld + h * A_HD;
  const bf16_t* kb = k + (long)b * N * ld + h * A_HD;
  const bf16_t* vb = v + (long)b * N * ld + h * A_HD;
  const bf16_t* dob = dout + (long)b * N * ldo + h * A_HD;
  const float* lse_b = lse + ((long)b * H + h) * N;
  const float* dvec_b = dvec + ((long)b * H + h) * N;
  const int nkt = (N + 31) / 32, nchunk = (N + A_BWD_QC - 1) / A_BWD_QC, rounds = (nkt + 3) / 4;
  for (int rd = 0; rd < rounds; ++rd) {
    const int kt = rd * 4 + w;
    const bool active = kt < nkt;
    const int krow = kt * 32 + l31;
    s16x8 kf[4], vf[4];
    a_load_frags(kf, kb, ld, active ? krow : N, N, hi);
    a_load_frags(vf, vb, ld, active ? krow : N, N, hi);
    f32x16 gk0, gk1, gv0, gv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { gk0[r] = 0.f; gk1[r] = 0.f; gv0[r] = 0.f; gv1[r] = 0.f; }
    for (int c = 0; c < nchunk; ++c) {
      const int q0 = c * A_BWD_QC;
      int nq = N - q0; if (nq > A_BWD_QC) nq = A_BWD_QC;
      const int nqp = (nq + 31) / 32 * 32;
      if (nchunk > 1 || rd == 0) {
        __syncthreads();
        a_stage_rows(Qs, qb, ld, q0, N, nqp);
        a_stage_rows(Os, dob, ldo, q0, N, nqp);
        for (int i = threadIdx.x; i < nqp; i += 256) {
          lse_s[i] = (q0 + i < N) ? lse_b[q0 + i] * VDK_LOG2E : 0.f;
          dv_s[i] = (q0 + i < N) ? dvec_b[q0 + i] : 0.f;
        }
        __syncthreads();
      }
      if (active) {
        for (int qt = 0; qt < nqp / 32; ++qt) {
          f32x16 st = a_mma_rows(Qs + (qt * 32 + l31) * A_RP + hi * 8, kf);    // S[q][key]
          f32x16 dp = a_mma_rows(Os + (qt * 32 + l31) * A_RP + hi * 8, vf);    // dP[q][key]
          float p[16], ds[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 lv = *(const f32x4*)(lse_s + qt * 32 + 8 * g + 4 * hi);
            f32x4 dvv = *(const f32x4*)(dv_s + qt * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e;
              const int qq = q0 + qt * 32 + 8 * g + 4 * hi + e;
              const float pv = (qq < N && krow < N) ? fast_exp2(fmaf(st[r], scale * VDK_LOG2E, -lv[e])) : 0.f;
              p[r] = pv;
              ds[r] = pv * (dp[r] - dvv[e]);
            }
          }
          s16x8 pf[2], df[2];
          a_pack_b(p, pf);
          a_pack_b(ds, df);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            s16x8 ao0 = a_load_T(Os, qt * 32, 0, s, hi, lane);
            s16x8 ao1 = a_load_T(Os, qt * 32, 32, s, hi, lane);
            gv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao0, pf[s], gv0, 0, 0, 0);
            gv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao1, pf[s], gv1, 0, 0, 0);
            s16x8 aq0 = a_load_T(Qs, qt * 32, 0, s, hi, lane);
            s16x8 aq1 = a_load_T(Qs, qt * 32, 32, s, hi, lane);
            gk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq0, df[s], gk0, 0, 0, 0);
            gk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq1, df[s], gk1, 0, 0, 0);
          }
        }
      }
    }
    if (active && krow < N) {
      bf16_t* krow_p = dk + ((long)b * N + krow) * lddkv + h * A_HD;
      bf16_t* vrow_p = dv + ((long)b * N + krow) * lddkv + h * A_HD;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * hi;
        *(u32x2*)(krow_p + d) = (u32x2){pack_bf2(gk0[4 * g] * scale, gk0[4 * g + 1] * scale), pack_bf2(gk0[4 * g + 2] * scale, gk0[4 * g + 3] * scale)};
        *(u32x2*)(krow_p + 32 + d) = (u32x2){pack_bf2(gk1[4 * g] * scale, gk1[4 * g + 1] * scale), pack_bf2(gk1[4 * g + 2] * scale, gk1[4 * g + 3] * scale)};
        *(u32x2*)(vrow_p + d) = (u32x2){pack_bf2(gv0[4 * g], gv0[4 * g + 1]), pack_bf2(gv0[4 * g + 2], gv0[4 * g + 3])};
        *(u32x2*)(vrow_p + 32 + d) = (u32x2){pack_bf2(gv1[4 * g], gv1[4 * g + 1]), pack_bf2(gv1[4 * g + 2], gv1[4 * g + 3])};
      }
    }
  }
}

// short sequences (attention_small.hip): everything of one (batch, head) item in LDS, exact softmax, fused backward
int vdk_attention_small_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H, float scale, int opf, void* stream);
int vdk_attention_small_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv, int64_t ldd, float* dvec, int32_t B, int32_t N,
                            int32_t H, float scale, int opf, void* stream);
// long sequences (attention_long.hip): K / V streamed through a double-buffered LDS chunk by LDS-DMA, online softmax
int vdk_attention_long_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H, float scale, int opf, void* stream);
int vdk_attention_long_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv, int64_t ldd, float* dvec, int32_t B, int32_t N,
                           int32_t H, float scale, int opf, void* stream);
static thread_local int g_attn_legacy = -1;   // -1: env VDK_ATTN_LEGACY decides; 0 / 1: forced by vdk_attention_force_legacy (A/B benchmarking, tests of the long-sequence kernels at small N)
static bool attn_legacy() {
  if (g_attn_legacy >= 0) return g_attn_legacy == 1;
  const char* e = getenv("VDK_ATTN_LEGACY");
  return e && e[0] == '1';
}

void vdk_attention_small_want_colsum(float*);      // attention_small.hip (C++ linkage, like vdk_attention_small_bwd above)
int vdk_attention_small_colsum_produced();
static int attn_long_min() {           // A/B and tests: VDK_ATTN_LONG_MIN=n routes every N >= n to the streaming kernels of attention_long.hip (default: beyond the LDS-resident range)
  const char* e = getenv("VDK_ATTN_LONG_MIN");                          // (read per launch: the tests switch it inside one process)
  return e ? atoi(e) : 1 << 30;
}
static bool attn_long_bwd_off() {      // A/B: VDK_ATTN_LONG_BWD=0 keeps the flash-style backward for N > 224 while the forward is routed
  static const bool off = getenv("VDK_ATTN_LONG_BWD") && getenv("VDK_ATTN_LONG_BWD")[0] == '0';
  return off;
}

extern "C" {

int vdk_attention_fwd_dt(const void*, int64_t, void*, int64_t, float*, int32_t, int32_t, int32_t, int32_t, float, int32_t, void*);
int vdk_attention_bwd_dt(const void*, int64_t, const void*, const void*, int64_t, const float*, void*, int64_t, float*, int32_t, int32_t, int32_t, int32_t, float, int32_t, void*);
int vdk_attention_force_legacy(int32_t on) { g_attn_legacy = on < 0 ? -1 : (on ? 1 : 0); return VDK_OK; }

// qkv: bf16 [B, N, 3, H, 64] (timm's fused qkv Linear output, row stride ld = 3*H*64); o: bf16 [B, N, H*64];
// lse: f32 [B, H, N] (saved for backward; may be NULL for inference).
int vdk_attention_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H,
                      int32_t head_dim, float scale, void* stream) {
  return vdk_attention_fwd_dt(qkv, ld, o, ldo, lse, B, N, H, head_dim, scale, VDK_BF16, stream);
}
// the same with the tensors' 16-bit format as a parameter: dtype VDK_BF16 | VDK_F16 (fp16: q, k, v, o and the probabilities P are IEEE half -- the reference's autocast
// arithmetic, engine/procedure/train.py:118)
int vdk_attention_fwd_dt(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H,
                         int32_t head_dim, float scale, int32_t dtype, void* stream) {
  if (!qkv || !o || B <= 0 || N <= 0 || H <= 0 || (dtype != VDK_BF16 && dtype != VDK_F16)) return vdk_fail(VDK_EINVAL, "vdk_attention_fwd: bad argument");
  const int opf = dtype == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16;
  if (head_dim != A_HD) return vdk_fail(VDK_EUNSUPPORTED, "vdk_attention_fwd: head_dim must be 64");
  if ((ld & 7) || (ldo & 7)) return vdk_fail(VDK_EINVAL, "vdk_attention_fwd: ld % 8");
  if (N <= 256 && N < attn_long_min() && !attn_legacy()) {
    const int rc = vdk_attention_small_fwd(qkv, ld, o, ldo, lse, B, N, H, scale, opf, stream);
    if (rc != VDK_EUNSUPPORTED) return rc ? rc : vdk_check_launch("vdk_attention_fwd");
  }
  if ((N > 256 || N >= attn_long_min()) && !attn_legacy()) {
    const int rc = vdk_attention_long_fwd(qkv, ld, o, ldo, lse, B, N, H, scale, opf, stream);
    return rc ? rc : vdk_check_launch("vdk_attention_fwd");
  }
  if (opf) return vdk_fail(VDK_EUNSUPPORTED, "vdk_attention_fwd: the flash-style kernels of round 1 (vdk_attention_force_legacy) are bf16 only");
  const bf16_t* base = (const bf16_t*)qkv;
  const long D = (long)H * A_HD;
  int grid = B * H;
  int cap = 512;                                            // persistent: two workgroups per CU walk over the (batch, head) items
  if (const char* e = getenv("VDK_ATTN_GRID")) { const int v = atoi(e); if (v > 0) cap = v; }   // tests: force several items per workgroup
  if (N <= A_FWD_KC && grid > cap) grid = cap;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, base, base + D, base + 2 * D,
                     (long)ld, (bf16_t*)o, (long)ldo, lse, (int)N, (int)H, scale, (int)(B * H));
  return vdk_check_launch("vdk_attention_fwd");
}

// dqkv: bf16 [B, N, 3, H, 64] like qkv.  dvec: f32 scratch [B, H, N].
// the same backward with the qkv.bias gradient's partials as a by-product where the chosen kernel can deliver them (the one-pass small-N form): cspart f32 [B][3 * H * 64]
// = per image the column sums of its dq | dk | dv rows as stored; *produced = 1 when written (the caller reduces over B), 0 otherwise (the caller sums dqkv itself)
int vdk_attention_bwd_cs(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv, int64_t lddqkv, float* dvec, int32_t B, int32_t N,
                         int32_t H, int32_t head_dim, float scale, int32_t dtype, float* cspart, int32_t* produced, void* stream_) {
  vdk_attention_small_want_colsum(cspart);
  const int rc = vdk_attention_bwd_dt(qkv, ld, o, dout, ldo, lse, dqkv, lddqkv, dvec, B, N, H, head_dim, scale, dtype, stream_);
  if (produced) *produced = rc == VDK_OK ? vdk_attention_small_colsum_produced() : 0;
  vdk_attention_small_want_colsum(nullptr);
  return rc;
}
int vdk_attention_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv,
                      int64_t lddqkv, float* dvec, int32_t B, int32_t N, int32_t H, int32_t head_dim, float scale, void* stream_) {
  return vdk_attention_bwd_dt(qkv, ld, o, dout, ldo, lse, dqkv, lddqkv, dvec, B, N, H, head_dim, scale, VDK_BF16, stream_);
}
int vdk_attention_bwd_dt(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv,
                         int64_t lddqkv, float* dvec, int32_t B, int32_t N, int32_t H, int32_t head_dim, float scale, int32_t dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !o || !dout || !lse || !dqkv || !dvec || B <= 0 || N <= 0 || H <= 0 || (dtype != VDK_BF16 && dtype != VDK_F16)) return vdk_fail(VDK_EINVAL, "vdk_attention_bwd: bad argument");
  const int opf = dtype == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16;
  if (head_dim != A_HD) return vdk_fail(VDK_EUNSUPPORTED, "vdk_attention_bwd: head_dim must be 64");
  if ((ld & 7) || (ldo & 7) || (lddqkv & 7)) return vdk_fail(VDK_EINVAL, "vdk_attention_bwd: ld % 8");
  if (N <= 224 && N < attn_long_min() && !attn_legacy()) {
    const int rc = vdk_attention_small_bwd(qkv, ld, o, dout, ldo, lse, dqkv, lddqkv, dvec, B, N, H, scale, opf, stream_);
    if (rc != VDK_EUNSUPPORTED) return rc ? rc : vdk_check_launch("vdk_attention_bwd");
  }
  if ((N > 224 || N >= attn_long_min()) && !attn_legacy() && !attn_long_bwd_off()) {
    const int rc = vdk_attention_long_bwd(qkv, ld, o, dout, ldo, lse, dqkv, lddqkv, dvec, B, N, H, scale, opf, stream_);
    return rc ? rc : vdk_check_launch("vdk_attention_bwd");
  }
  if (opf) return vdk_fail(VDK_EUNSUPPORTED, "vdk_attention_bwd: the flash-style kernels of round 1 are bf16 only");
  const bf16_t* base = (const bf16_t*)qkv;
  bf16_t* dbase = (bf16_t*)dqkv;
  const long D = (long)H * A_HD;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(B * H)), dim3(256), 0, stream, base, base + D, base + 2 * D, (long)ld,
                     (const bf16_t*)o, (const bf16_t*)dout, (long)ldo, lse, dbase, (long)lddqkv, dvec, (int)N, (int)H, scale);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(B * H)), dim3(256), 0, stream, base, base + D, base + 2 * D, (long)ld,
                     (const bf16_t*)dout, (long)ldo, lse, (const float*)dvec, dbase + D, dbase + 2 * D, (long)lddqkv, (int)N, (int)H,
                     scale);
  return vdk_check_launch("vdk_attention_bwd");
}

}  // extern "C"

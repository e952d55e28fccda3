// cbir.hip — hot path B: L2-normalise + all-pairs inner product + top-k (faiss IndexFlatIP.search
// as called at /root/reference engine/cbir/evaluation.py:155-168,193 and cbir_eval.py:82-95,116;
// F.normalize at models/faceX/face_model.py:139).
//
// Design (MI355X-first):
//  * scores are EXACT fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain, so the GPU score of
//    every (query, gallery) pair is bit-identical to oracle/cbir_oracle.c's sequential fmaf chain
//    and the top-k indices can be compared bit-exactly.
//  * the Q x N score matrix is never written: each 128-query x 128-row tile is compared in
//    registers against a per-query threshold (k-th best of the rows seen in earlier stages); the
//    rare survivors are appended to an LDS buffer and flushed to per-query candidate lists.
//  * the gallery is scanned in stages of geometrically growing size; after every stage a select
//    kernel sorts each query's candidates by (score desc, index asc), keeps k and raises the
//    threshold.  A stage never holds more rows than (cap - k), so candidate lists cannot
//    overflow by construction (no host sync, no data-dependent retry).
//  * "swapped" operands: A = gallery rows, B = queries, so a lane owns one query column and the
//    threshold compare is per-lane.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "vdk_device.h"
#include "vdk_host.h"

#define CB_BQ 128        // queries per workgroup
#define CB_BG 128        // gallery rows per tile
#define CB_KC 128        // contraction chunk
#define CB_PITCH 68      // floats per (parity, row) line: 64 + 4 pad -> conflict-free ds_read_b128
#define CB_E 1024        // LDS candidate entries
#define CB_EFLUSH 512

// ------------------------------------------------------------------------------------ K12
// out[r][:] = x[r][:] / max(||x[r]||_2, eps)   (F.normalize(p=2, dim=1, eps=1e-12))
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          long n, int d, float eps) {
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= n) return;  // whole wave exits together
  const float* xr = x + row * (long)d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) { float v = xr[c]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  float inv = 1.0f / fmaxf(sqrtf(s), eps);
  float* orow = out + row * (long)d;
  for (int c = lane; c < d; c += 64) orow[c] = xr[c] * inv;
}

// ------------------------------------------------------------------------------------ K13a
struct CbirCand {
  float* score;        // [nq][cap]
  int* idx;            // [nq][cap]
  unsigned* cnt;       // [nq]
  unsigned* overflow;  // [1]
  long cap;
};

__device__ __forceinline__ void cbir_global_append(const CbirCand& c, long q, float s, int gidx) {
  unsigned pos = atomicAdd(&c.cnt[q], 1u);
  if ((long)pos < c.cap) {
    c.score[q * c.cap + pos] = s;
    c.idx[q * c.cap + pos] = gidx;
  } else {
    atomicOr(c.overflow, 1u);
  }
}

// stage a [128 rows][128 floats] chunk of a row-major matrix into the de-interleaved LDS image
//   dst[parity][row][m]  (k = 2m + parity), pitch CB_PITCH floats
__device__ __forceinline__ void cbir_load_chunk(f32x4 (&regs)[8], const float* __restrict__ base, long row0,
                                                long row_end, int D, int kc) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int id = threadIdx.x + 512 * j;
    int r = id >> 5, c4 = id & 31;
    long row = row0 + r;
    int k = kc + c4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < row_end && k < D) v = *(const f32x4*)(base + row * (long)D + k);  // D % 4 == 0
    regs[j] = v;
  }
}
__device__ __forceinline__ void cbir_store_chunk(const f32x4 (&regs)[8], float* lds /*[2][128][PITCH]*/) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int id = threadIdx.x + 512 * j;
    int r = id >> 5, c4 = id & 31;
    f32x2 ev = {regs[j][0], regs[j][2]};
    f32x2 od = {regs[j][1], regs[j][3]};
    *(f32x2*)(lds + (0 * 128 + r) * CB_PITCH + 2 * c4) = ev;
    *(f32x2*)(lds + (1 * 128 + r) * CB_PITCH + 2 * c4) = od;
  }
}

// grid: 1-D, id -> split = id % nsplit (so that a workgroup's XCD, id % 8, only ever streams its own
// gallery slice when nsplit % 8 == 0), qblock = id / nsplit.  block: 512 threads = 8 waves:
//   wave w: query group qg = w & 3 (32 queries), row half rg = w >> 2 (64 of the tile's 128 rows).
__global__ __launch_bounds__(512) void cbir_score_filter_kernel(
    const float* __restrict__ Q, long nq, const float* __restrict__ G, int D, long g_begin, long g_end,
    long rows_per_split, int nsplit, long idx_base, const float* __restrict__ thr, CbirCand cand) {
  __shared__ __attribute__((aligned(16))) float Qs[2 * 128 * CB_PITCH];
  __shared__ __attribute__((aligned(16))) float Gs[2 * 128 * CB_PITCH];
  __shared__ float e_score[CB_E];
  __shared__ int e_idx[CB_E];
  __shared__ unsigned short e_q[CB_E];
  __shared__ unsigned s_cnt;

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int qg = w & 3, rg = w >> 2;
  const int hi = lane >> 5, col = lane & 31;
  const int split = blockIdx.x % nsplit;
  const long qb = blockIdx.x / nsplit;
  const long q0 = qb * CB_BQ;
  const long r_begin = g_begin + (long)split * rows_per_split;
  long r_end = r_begin + rows_per_split;
  if (r_end > g_end) r_end = g_end;
  if (tid == 0) s_cnt = 0;
  if (r_begin >= r_end) return;  // uniform per block

  const int nchunk = (D + CB_KC - 1) / CB_KC;
  const long my_q = q0 + qg * 32 + col;
  const float my_thr = (my_q < nq) ? thr[my_q] : __uint_as_float(0x7f800000u);  // +inf: nothing passes

  f32x4 regs[8];
  if (nchunk == 1) {
    cbir_load_chunk(regs, Q, q0, nq, D, 0);
    cbir_store_chunk(regs, Qs);
  }
  const long ntile = (r_end - r_begin + CB_BG - 1) / CB_BG;
  cbir_load_chunk(regs, G, r_begin, r_end, D, 0);

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

  const float* a0p = Gs + (hi * 128 + rg * 64 + col) * CB_PITCH;
  const float* a1p = a0p + 32 * CB_PITCH;
  const float* bp = Qs + (hi * 128 + qg * 32 + col) * CB_PITCH;

  for (long t = 0; t < ntile; ++t) {
    const long row0 = r_begin + t * CB_BG;
    for (int c = 0; c < nchunk; ++c) {
      cbir_store_chunk(regs, Gs);
      if (nchunk > 1) {
        f32x4 qregs[8];
        cbir_load_chunk(qregs, Q, q0, nq, D, c * CB_KC);
        cbir_store_chunk(qregs, Qs);
      }
      __syncthreads();
      // prefetch the next (tile, chunk) while the MFMAs run
      {
        int cn = c + 1; long tn = t;
        if (cn == nchunk) { cn = 0; tn = t + 1; }
        if (tn < ntile) cbir_load_chunk(regs, G, r_begin + tn * CB_BG, r_end, D, cn * CB_KC);
      }
#pragma unroll 4
      for (int m4 = 0; m4 < CB_KC / 8; ++m4) {  // 16 x (4 k-pairs)
        f32x4 a0 = *(const f32x4*)(a0p + 4 * m4);
        f32x4 a1 = *(const f32x4*)(a1p + 4 * m4);
        f32x4 b = *(const f32x4*)(bp + 4 * m4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[e], acc1, 0, 0, 0);
        }
      }
      if (c == nchunk - 1) {
        // epilogue: per-lane threshold compare (lane owns query `my_q`), survivors -> LDS buffer
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float s = (tt == 0 ? acc0[r] : acc1[r]) + 0.0f;  // canonicalise -0
            long row = row0 + rg * 64 + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < r_end && s > my_thr) {
              int gidx = (int)(idx_base + row);
              unsigned slot = atomicAdd(&s_cnt, 1u);
              if (slot < CB_E) {
                e_score[slot] = s; e_idx[slot] = gidx; e_q[slot] = (unsigned short)(qg * 32 + col);
              } else {
                cbir_global_append(cand, my_q, s, gidx);
              }
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      }
      __syncthreads();
    }
    // flush the LDS candidate buffer when it is half full (uniform decision: s_cnt read after barrier)
    unsigned n = s_cnt;
    if (n >= CB_EFLUSH || t == ntile - 1) {
      if (n > CB_E) n = CB_E;
      for (unsigned i = tid; i < n; i += 512) cbir_global_append(cand, q0 + e_q[i], e_score[i], e_idx[i]);
      __syncthreads();
      if (tid == 0) s_cnt = 0;
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------ K13b
// One workgroup per query: sort the candidate list by (score desc, index asc), keep the best k as
// the carry for the next stage, publish thr = k-th score (or -inf while fewer than k are known) and
// write the (padded) result rows.  Lists longer than SORT_MAX - keep are consumed in rounds.
#define CB_SORT_MAX 4096
__device__ __forceinline__ unsigned long long cbir_key(float s, int idx) {
  return ((unsigned long long)(~f2ord(s)) << 32) | (unsigned)idx;
}
__global__ __launch_bounds__(256) void cbir_select_kernel(CbirCand cand, long nq, int k, float* __restrict__ thr,
                                                          float* __restrict__ out_score, long long* __restrict__ out_idx,
                                                          int write_out, unsigned* __restrict__ carry) {
  __shared__ unsigned long long keys[CB_SORT_MAX];
  const long q = blockIdx.x;
  const int tid = threadIdx.x;
  unsigned n_total = cand.cnt[q];
  if ((long)n_total > cand.cap) n_total = (unsigned)cand.cap;
  int keep = 1; while (keep < k) keep <<= 1;      // pow2 >= k, <= 1024
  const float* cs = cand.score + q * cand.cap;
  const int* ci = cand.idx + q * cand.cap;

  unsigned consumed = 0;
  int have = 0;  // valid carried keys currently in keys[0..have)
  do {
    unsigned room = CB_SORT_MAX - keep;
    unsigned take = n_total - consumed; if (take > room) take = room;
    unsigned n = have + take;
    unsigned sortn = 64; while (sortn < n) sortn <<= 1;
    __syncthreads();
    for (unsigned i = tid; i < sortn; i += 256) {
      if (i >= (unsigned)have) {
        unsigned j = i - have;
        keys[i] = (j < take) ? cbir_key(cs[consumed + j], ci[consumed + j]) : ~0ull;
      }
    }
    __syncthreads();
    for (unsigned size = 2; size <= sortn; size <<= 1) {
      for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
        for (unsigned p = tid; p < (sortn >> 1); p += 256) {
          unsigned lo = ((p / stride) * 2 * stride) + (p % stride);
          unsigned hi2 = lo + stride;
          bool asc = ((lo & size) == 0);
          unsigned long long a = keys[lo], b = keys[hi2];
          if ((a > b) == asc) { keys[lo] = b; keys[hi2] = a; }
        }
        __syncthreads();
      }
    }
    consumed += take;
    have = (int)(n < (unsigned)keep ? n : (unsigned)keep);
    if (have > k) have = k;
  } while (consumed < n_total);

  __syncthreads();
  // carry + threshold + outputs
  for (int i = tid; i < have; i += 256) {
    unsigned long long key = keys[i];
    float s = ord2f(~(unsigned)(key >> 32));
    cand.score[q * cand.cap + i] = s;
    cand.idx[q * cand.cap + i] = (int)(unsigned)(key & 0xffffffffu);
  }
  if (tid == 0) {
    if (carry) carry[q] = (unsigned)have;
    cand.cnt[q] = (unsigned)have;
    // monotone: a bootstrap threshold (a valid lower bound of the k-th best) stays in force until k rows are ranked
    if (have >= k) thr[q] = fmaxf(thr[q], ord2f(~(unsigned)(keys[k - 1] >> 32)));
  }
  if (write_out) {
    for (int i = tid; i < k; i += 256) {
      if (i < have) {
        unsigned long long key = keys[i];
        out_score[q * k + i] = ord2f(~(unsigned)(key >> 32));
        out_idx[q * k + i] = (long long)(int)(unsigned)(key & 0xffffffffu);
      } else {
        out_score[q * k + i] = -3.4028234663852886e38f;  // faiss pads D with -FLT_MAX, I with -1
        out_idx[q * k + i] = -1;
      }
    }
  }
}


// ------------------------------------------------------------------------------------ K13c
// Wave-per-query ranking for k <= 256 (the common case: the reference searches k = 100).  Same contract as
// cbir_select_kernel, optionally fused with the exact re-scoring of the prefilter path (Q != nullptr: entries
// [carry[q], cnt[q]) hold only an index and get the oracle's k-ordered fmaf chain here).  A wave sorts up to CW_KEYS keys
// in its own 8 KB LDS region with wave-synchronous bitonic steps (no workgroup barriers), so 20 waves per CU hide each
// other's global-latency chains; typical steady-state input is k carried keys + ~10 new ones.
// the oracle's score of one pair: k-ordered fmaf chain from +0 (oracle/cbir_oracle.c oracle_ip_pair), loads batched 8 deep
__device__ __forceinline__ float cbir_exact_ip(const float* __restrict__ qrow, const float* __restrict__ g, int D) {
  float acc = 0.f;
  int c = 0;
  for (; c + 32 <= D; c += 32) {
    f32x4 gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gv[j] = *(const f32x4*)(g + c + 4 * j);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 qv = *(const f32x4*)(qrow + c + 4 * j);
      acc = fmaf(qv[0], gv[j][0], acc); acc = fmaf(qv[1], gv[j][1], acc); acc = fmaf(qv[2], gv[j][2], acc); acc = fmaf(qv[3], gv[j][3], acc);
    }
  }
  for (; c < D; c += 4) {
    const f32x4 gv = *(const f32x4*)(g + c);
    const f32x4 qv = *(const f32x4*)(qrow + c);
    acc = fmaf(qv[0], gv[0], acc); acc = fmaf(qv[1], gv[1], acc); acc = fmaf(qv[2], gv[2], acc); acc = fmaf(qv[3], gv[3], acc);
  }
  return acc + 0.0f;
}

// the same chain over a gallery row STORED in fp16 (faiss GpuClonerOptions.useFloat16, engine/cbir/evaluation.py:157-162): every element converts exactly to
// fp32, so the score equals cbir_exact_ip on the fp16-rounded row
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float cbir_exact_ip_h(const float* __restrict__ qrow, const _Float16* __restrict__ g, int D) {
  float acc = 0.f;
  int c = 0;
  for (; c + 64 <= D; c += 64) {          // loads batched 8 deep like the fp32 chain (the chain itself is latency-bound, the loads must not be)
    f16x8_t gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gv[j] = *(const f16x8_t*)(g + c + 8 * j);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 q0 = *(const f32x4*)(qrow + c + 8 * j), q1 = *(const f32x4*)(qrow + c + 8 * j + 4);
      acc = fmaf(q0[0], (float)gv[j][0], acc); acc = fmaf(q0[1], (float)gv[j][1], acc); acc = fmaf(q0[2], (float)gv[j][2], acc); acc = fmaf(q0[3], (float)gv[j][3], acc);
      acc = fmaf(q1[0], (float)gv[j][4], acc); acc = fmaf(q1[1], (float)gv[j][5], acc); acc = fmaf(q1[2], (float)gv[j][6], acc); acc = fmaf(q1[3], (float)gv[j][7], acc);
    }
  }
  for (; c + 8 <= D; c += 8) {
    const f16x8_t gv = *(const f16x8_t*)(g + c);
    const f32x4 q0 = *(const f32x4*)(qrow + c), q1 = *(const f32x4*)(qrow + c + 4);
    acc = fmaf(q0[0], (float)gv[0], acc); acc = fmaf(q0[1], (float)gv[1], acc); acc = fmaf(q0[2], (float)gv[2], acc); acc = fmaf(q0[3], (float)gv[3], acc);
    acc = fmaf(q1[0], (float)gv[4], acc); acc = fmaf(q1[1], (float)gv[5], acc); acc = fmaf(q1[2], (float)gv[6], acc); acc = fmaf(q1[3], (float)gv[7], acc);
  }
  for (; c < D; ++c) acc = fmaf(qrow[c], (float)g[c], acc);
  return acc + 0.0f;
}
// G: fp32 rows, or fp16 rows when g_half
__device__ __forceinline__ float cbir_exact_ip_any(const float* __restrict__ qrow, const float* __restrict__ G, long row, int D, int g_half) {
  return g_half ? cbir_exact_ip_h(qrow, (const _Float16*)G + row * (long)D, D) : cbir_exact_ip(qrow, G + row * (long)D, D);
}

#define CW_KEYS 512
__global__ __launch_bounds__(256) void cbir_rank_wave_kernel(const float* __restrict__ Q, const float* __restrict__ G, int D, long idx_base,
                                                             CbirCand cand, long nq, int k, float* __restrict__ thr, float* __restrict__ out_score,
                                                             long long* __restrict__ out_idx, int write_out, unsigned* __restrict__ carry, int g_half,
                                                             CbirCand dst, int reserve) {
  // reserve == 0: one list per query, [0, carry) = sorted exact entries, [carry, cnt) = new ones, the result goes back to [0, have).
  // reserve == k (pipelined schedule): slots [0, k) of a list are reserved for the carry, the pre-filter appends from slot k on; the result is written to the
  // OTHER list set `dst` (which the next stage's pre-filter is filling concurrently from ITS slot k on), and this list's counter is re-armed to k.
  __shared__ unsigned long long keys_all[4][CW_KEYS];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long q = (long)blockIdx.x * 4 + w;
  if (q >= nq) return;   // wave-uniform
  unsigned long long* K = keys_all[w];
  unsigned n_total = cand.cnt[q];
  if ((long)n_total > cand.cap) n_total = (unsigned)cand.cap;
  const unsigned c0 = Q ? carry[q] : n_total;           // sorted entries with exact scores at the front
  const unsigned new0 = reserve ? (unsigned)reserve : c0;   // where the new (index-only) entries start
  const unsigned n_new = n_total > new0 ? n_total - new0 : 0u;
  const unsigned n_all = c0 + n_new;                     // logical entry e < c0: slot e, else slot new0 + (e - c0)
  const float thr_q = thr[q];
  float* cs = cand.score + q * cand.cap;
  int* ci = cand.idx + q * cand.cap;
  int keep = 64; while (keep < k) keep <<= 1;   // pow2 >= k, <= 256
  const float* qrow = Q ? Q + q * (long)D : nullptr;

  unsigned consumed = 0, have = 0;
  if (Q && n_new <= 64 && c0 <= (unsigned)k) {
    // steady state of the prefilter path: entries [0, c0) are the sorted carry, at most 64 new ones follow.  Rank-merge in
    // registers: every new key is broadcast, each lane counts how many of its carry keys precede it (ballots), no sort.
    unsigned long long nk = ~0ull;
    bool v = false;
    if ((unsigned)lane < n_new) {
      const int gi = ci[new0 + lane];
      const float sc = cbir_exact_ip_any(qrow, G, (long)gi - idx_base, D, g_half);
      if (!(sc < thr_q)) { nk = cbir_key(sc, gi); v = true; }
    }
    const unsigned long long vb = __ballot(v);
    unsigned long long ck[4];
    unsigned sh[4] = {0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned j = lane + 64 * m;
      ck[m] = j < c0 ? cbir_key(cs[j], ci[j]) : ~0ull;
    }
    unsigned myrank = 0;
    for (unsigned long long b = vb; b; b &= b - 1) {
      const int src = __ffsll(b) - 1;
      const unsigned long long x = __shfl(nk, src);
      unsigned less = (unsigned)__popcll(__ballot(v && nk < x));
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        less += (unsigned)__popcll(__ballot(ck[m] < x));
        sh[m] += (unsigned)(x < ck[m]);
      }
      if (lane == src) myrank = less;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned j = lane + 64 * m;
      if (j < c0 && j + sh[m] < (unsigned)k) K[j + sh[m]] = ck[m];
    }
    if (v && myrank < (unsigned)k) K[myrank] = nk;
    have = c0 + (unsigned)__popcll(vb);
    if (have > (unsigned)k) have = (unsigned)k;
    consumed = n_all;
    VDK_WAVE_LDS_SYNC();
  }
  while (consumed < n_all) {
    unsigned take = n_all - consumed;
    if (take > CW_KEYS - (unsigned)keep) take = CW_KEYS - (unsigned)keep;
    const unsigned n = have + take;
    unsigned sortn = 64; while (sortn < n) sortn <<= 1;
    unsigned valid = 0;
    for (unsigned i = lane; i < sortn - have; i += 64) {
      unsigned long long key = ~0ull;
      if (i < take) {
        const unsigned el = consumed + i, e = el < c0 ? el : new0 + (el - c0);
        const int gi = ci[e];
        float sc;
        if (el >= c0) {
          sc = cbir_exact_ip_any(qrow, G, (long)gi - idx_base, D, g_half);
        } else {
          sc = cs[e];
        }
        // a row strictly below the current k-th best can never enter the top-k (ties stay: the index decides)
        if (!(sc < thr_q)) { key = cbir_key(sc, gi); ++valid; }
      }
      K[have + i] = key;
    }
    valid = (unsigned)wave_sum((float)valid);   // exact: counts <= 1024
    VDK_WAVE_LDS_SYNC();
    for (unsigned size = 2; size <= sortn; size <<= 1)
      for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
        for (unsigned p = lane; p < (sortn >> 1); p += 64) {
          const unsigned lo = ((p / stride) * 2 * stride) + (p % stride), hi2 = lo + stride;
          const bool asc = ((lo & size) == 0);
          const unsigned long long a = K[lo], b = K[hi2];
          if ((a > b) == asc) { K[lo] = b; K[hi2] = a; }
        }
        VDK_WAVE_LDS_SYNC();
      }
    consumed += take;
    have += valid;
    if (have > (unsigned)k) have = (unsigned)k;
  }

  float* os = reserve ? dst.score + q * dst.cap : cs;
  int* oi = reserve ? dst.idx + q * dst.cap : ci;
  for (unsigned i = lane; i < have; i += 64) {
    const unsigned long long key = K[i];
    os[i] = ord2f(~(unsigned)(key >> 32));
    oi[i] = (int)(unsigned)(key & 0xffffffffu);
  }
  if (lane == 0) {
    if (carry) carry[q] = have;
    cand.cnt[q] = reserve ? (unsigned)reserve : have;
    if (have >= (unsigned)k) thr[q] = fmaxf(thr_q, ord2f(~(unsigned)(K[k - 1] >> 32)));
  }
  if (write_out) {
    for (int i = lane; i < k; i += 64) {
      if ((unsigned)i < have) {
        const unsigned long long key = K[i];
        out_score[q * k + i] = ord2f(~(unsigned)(key >> 32));
        out_idx[q * k + i] = (long long)(int)(unsigned)(key & 0xffffffffu);
      } else {
        out_score[q * k + i] = -3.4028234663852886e38f;
        out_idx[q * k + i] = -1;
      }
    }
  }
}

// =====================================================================================================================
// K13 fast path (D <= 128): bf16-MFMA candidate pre-filter with a RIGOROUS error bound + exact fp32 re-scoring.
//   * gallery and queries are additionally kept as bf16 rows padded to 128 columns (Gb built once per index).
//   * approximate score s' = sum_k q~_k g~_k with q~ = bf16(q), g~ = bf16(g) (products exact in fp32, fp32 accumulation).
//     Exactly:  s' - s = sum (q~ - q) g~ + sum q (g~ - g),  so by Cauchy-Schwarz
//         |s' - s| <= ||q~ - q|| * max_n ||g~_n||  +  ||q|| * max_n ||g~_n - g_n||  +  acc,
//     where the rounding-error norms are MEASURED per row by the cast kernel (typically 0.0016 ||x||, against the worst case
//     2^-8 ||x||) and acc <= 4e-5 ||q~|| max ||g~|| covers the MFMA's fp32 accumulation (128 terms x 2^-24, 5x margin).
//     All norms are rounded up (x 1.000002).
//   * a row can only belong to the final top-k if its exact score beats the current exact k-th best `thr`; such a row has
//     s' > thr - eps_q, so the filter `s' > thr - eps_q` never drops a true member.  Survivors (a few thousand per query over
//     the whole scan) get their EXACT score from the same k-ordered fmaf chain as the oracle (cbir_rescore_kernel) before
//     the select kernel ranks them -> results are bit-identical to the exact scan, at bf16 MFMA speed.
#define CF_BQ 512
#define CF_BG 128
#define CF_WE 512      // staged survivors per wave

// Qb / Gb rows: bf16 [rows, 128] (zero padded); norms3[row] = (||x||, ||bf16(x)||, ||bf16(x) - x||), each rounded up
__global__ __launch_bounds__(256) void cbir_cast_rows_kernel(const float* __restrict__ x, long n, int D, int DP, bf16_t* __restrict__ xb, float* __restrict__ norms3) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  float sx = 0.f, st = 0.f, se = 0.f;
  for (int c = lane * 2; c < DP; c += 128) {      // DP = D rounded up to 128 columns (zero padded)
    const float a = c < D ? x[row * (long)D + c] : 0.f, b = c + 1 < D ? x[row * (long)D + c + 1] : 0.f;
    const unsigned pk = pack_bf2(a, b);
    *(unsigned*)(xb + row * (long)DP + c) = pk;
    const float at = __uint_as_float(pk << 16), bt = __uint_as_float(pk & 0xffff0000u);
    const float ea = at - a, eb = bt - b;   // exact (Sterbenz) or a correctly rounded difference: error << the 2e-6 margin
    sx += fmaf(b, b, a * a); st += fmaf(bt, bt, at * at); se += fmaf(eb, eb, ea * ea);
  }
  sx = wave_sum(sx); st = wave_sum(st); se = wave_sum(se);
  if (lane == 0) {
    norms3[row * 3 + 0] = sqrtf(sx) * 1.000002f;
    norms3[row * 3 + 1] = sqrtf(st) * 1.000002f;
    norms3[row * 3 + 2] = sqrtf(se) * 1.000002f;
  }
}
// out_bits[j] = bits of max_row norms3[row][j]  (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void cbir_max3_kernel(const float* __restrict__ v, long n, unsigned* __restrict__ out_bits) {
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    m0 = fmaxf(m0, v[i * 3]); m1 = fmaxf(m1, v[i * 3 + 1]); m2 = fmaxf(m2, v[i * 3 + 2]);
  }
  m0 = wave_max(m0); m1 = wave_max(m1); m2 = wave_max(m2);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out_bits, __float_as_uint(m0)); atomicMax(out_bits + 1, __float_as_uint(m1)); atomicMax(out_bits + 2, __float_as_uint(m2));
  }
}
__global__ void cbir_gstat_init_kernel(unsigned* __restrict__ bits, unsigned chunks) { if (threadIdx.x < 4) bits[threadIdx.x] = threadIdx.x == 3 ? chunks : 0u; }
// eps_q of the header comment
// (the accumulation term scales with the contraction length: 4e-5 covers 128 terms with a 5x margin, gstat_bits[3] holds DP / 128)
__device__ __forceinline__ float cbir_eps(const float* __restrict__ qn3, long q, const unsigned* __restrict__ gstat_bits) {
  const float gt = __uint_as_float(gstat_bits[1]), ge = __uint_as_float(gstat_bits[2]);
  const float qn = qn3[q * 3], qt = qn3[q * 3 + 1], qe = qn3[q * 3 + 2];
  const float len = gstat_bits[3] > 1u ? (float)gstat_bits[3] : 1.0f;
  return 1.00001f * (qe * gt + qn * ge) + 4e-5f * len * qt * gt;
}

// grid: id -> split = id % nsplit, qblock = id / nsplit.  512 threads = 8 waves; wave w owns 64 queries (two 32-wide column
// tiles whose bf16 fragments stay in 64 VGPRs for the whole scan) and scores them against all 128 rows of every gallery
// tile, 64 rows at a time.  Gallery tiles arrive by LDS-DMA into a 3-deep ring (two tiles = 64 KB in flight while one is
// consumed): the DMA has ~2 us latency under load, so depth, not bandwidth, is what keeps the MFMAs fed.
//
// BOOT = true is the threshold bootstrap: no candidates are produced, the kernel only reports, per query, the largest
// approximate score of every 128-row tile of a gallery sample (gm[q][tile]).  The k-th largest of these G >= k group maxima,
// minus eps_q, is a lower bound of the exact k-th best score (k distinct rows reach it), i.e. a valid filter threshold before
// any row is ranked: the scan starts with a tight cut instead of a pass-everything ramp (cbir_boot_thr_kernel).
// survivors of one 32 x 32 accumulator block (lane = query QL of the wave, register r = row ROWB + (r & 3) + 8 (r >> 2) + 4 hi) -> the wave's staging arrays.
// `m` = the lane's maximum over the block (the cheap reject that precedes this).  Only lanes that own a survivor build their row mask; slots are handed out lane by
// lane through SGPRs (v_readlane of each owner's count): no atomics, no LDS round trip on the wave's critical path.  A survivor travels with its approximate score
// (the approximate-ranking schedule ranks on it): an owner's only survivor -- the usual case -- IS its maximum `m`; the value of any other one comes from a
// select chain over the 16 registers (a lane-dependent register index has no cheaper form).  LAST: the tile may reach past r_end (rows there are clamped copies).
// (round 6, measured and dropped: one v_cmp per register into scalar masks, OR-ed on the scalar unit, with the emission under per-register scalar branches -- the
//  score then has a static register index -- was 0.5 ms SLOWER per search: the taken branches of the survivor path cost more than the per-lane masks here.)
#define CF_EMIT_OWNED(ACC, PM, POS, QL, ROWB)                                                                                                      \
  do {                                                                                                                                            \
    unsigned mm_ = (PM);                                                                                                                          \
    const bool single_ = (mm_ & (mm_ - 1u)) == 0u;                                                                                                 \
    while (mm_) {                                                                                                                                 \
      const int r = 15 - (__ffs(mm_) - 1);      /* CF_ROWBIT */                                                                                   \
      mm_ &= mm_ - 1;                                                                                                                             \
      float v_ = m;                                                                                                                               \
      if (!single_) { v_ = (ACC)[0]; _Pragma("unroll") for (int j_ = 1; j_ < 16; ++j_) v_ = r == j_ ? (ACC)[j_] : v_; }                             \
      e_idx[w][POS] = (int)(idx_base + (ROWB) + (r & 3) + 8 * (r >> 2) + 4 * hi);                                                                  \
      e_score[w][POS] = v_;                                                                                                                       \
      e_q[w][POS] = (unsigned char)(QL);                                                                                                          \
      ++(POS);                                                                                                                                    \
    }                                                                                                                                             \
  } while (0)
// row mask of a lane: bit 15 - r <-> register r.  cut - acc is negative exactly when acc > cut (no difference of distinct floats rounds to +0), so one subtraction
// and one funnel shift (v_alignbit: pm = pm << 1 | sign) per register, against compare + select + or
#define CF_ROWBIT(r) (15 - (r))
#define CF_STAGE(ACC, CUTV, QL, ROWB, LAST)                                                                                                       \
  do {                                                                                                                                            \
    unsigned pm = 0;                                                                                                                              \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) pm = __builtin_amdgcn_alignbit(pm, __float_as_uint((CUTV) - (ACC)[r]), 31);                     \
    pm &= 0xffffu;                                                                                                                                \
    if (LAST) {                                                                                                                                   \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                                              \
        if ((ROWB) + (r & 3) + 8 * (r >> 2) + 4 * hi >= r_end) pm &= ~(1u << CF_ROWBIT(r));                                                        \
    }                                                                                                                                             \
    const unsigned c = (unsigned)__popc(pm);                                                                                                      \
    /* slots: exclusive prefix sum of the lanes' counts (<= 16: five bit planes, a ballot and a masked popcount each) */                           \
    unsigned pre_ = 0, tot_ = 0;                                                                                                                  \
    _Pragma("unroll") for (int b_ = 0; b_ < 5; ++b_) {                                                                                             \
      const unsigned long long pl_ = __ballot((c >> b_) & 1u);                                                                                    \
      pre_ += (unsigned)__popcll(pl_ & lane_lt) << b_;                                                                                            \
      tot_ += (unsigned)__popcll(pl_) << b_;                                                                                                      \
    }                                                                                                                                             \
    if (wcnt + tot_ > CF_WE) { CF_FLUSH(); }                                                                                                       \
    if (tot_ <= CF_WE) {                                                                                                                          \
      unsigned pos = wcnt + pre_;                                                                                                                 \
      if (c) CF_EMIT_OWNED(ACC, pm, pos, QL, ROWB);                                                                                               \
      wcnt += tot_;                                                                                                                               \
    } else {   /* more survivors in one block than the staging holds (a pass-everything stage): lane by lane, flushing in between */              \
      unsigned long long bl = __ballot(c != 0);                                                                                                   \
      unsigned pos = 0;                                                                                                                           \
      for (unsigned long long b = bl; b; b &= b - 1) {                                                                                            \
        const int L = __ffsll(b) - 1;                                                                                                             \
        const unsigned cL = (unsigned)VDK_READLANE(c, L);                                                                                         \
        if (wcnt + cL > CF_WE) {   /* stage what was granted so far, then flush (uniform branch) */                                               \
          if (c && (bl & ~b & (1ull << lane))) { unsigned pp = pos; CF_EMIT_OWNED(ACC, pm, pp, QL, ROWB); }                                        \
          bl = b;   /* owners before L are done */                                                                                                \
          CF_FLUSH();                                                                                                                             \
        }                                                                                                                                         \
        if (lane == L) pos = wcnt;                                                                                                                \
        wcnt += cL;                                                                                                                               \
      }                                                                                                                                           \
      if (c && (bl & (1ull << lane))) CF_EMIT_OWNED(ACC, pm, pos, QL, ROWB);                                                                       \
    }                                                                                                                                             \
  } while (0)
#define CF_NBUF 3
template <bool BOOT>
__global__ __launch_bounds__(512) void cbir_prefilter_kernel(const bf16_t* __restrict__ Qb, const float* __restrict__ qnorm, long nq,
                                                             const bf16_t* __restrict__ Gb, const unsigned* __restrict__ gmax_bits, long g_begin,
                                                             long g_end, long rows_per_split, int nsplit, long idx_base,
                                                             const float* __restrict__ thr, CbirCand cand, float* __restrict__ gm, long gm_ld, int phased) {
  __shared__ __attribute__((aligned(16))) unsigned char Gs[CF_NBUF * CF_BG * 256];   // 3 x 32 KB
  // survivors are staged per WAVE (no shared counter, no workgroup barrier): slots come from ballots, flushes are wave-local
  __shared__ int e_idx[8][CF_WE];
  __shared__ float e_score[8][CF_WE];          // the approximate score s' of the survivor (the approximate-ranking schedule ranks on it; the exact schedules ignore it)
  __shared__ unsigned short e_rank[8][CF_WE];
  __shared__ unsigned char e_q[8][CF_WE];
  __shared__ unsigned f_cnt[8][64], f_base[8][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const unsigned long long lane_lt = (1ull << lane) - 1ull;
  unsigned wcnt = 0;   // wave-uniform number of staged survivors
  const int split = blockIdx.x % nsplit;
  const long q0 = (long)(blockIdx.x / nsplit) * CF_BQ;
  const long r_begin = g_begin + (long)split * rows_per_split;
  long r_end = r_begin + rows_per_split;
  if (r_end > g_end) r_end = g_end;
  if (r_begin >= r_end) return;

  const long qw0 = q0 + w * 64;   // first query of this wave (CF_FLUSH)
  s16x8 qf[2][8];
  float cut[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const long q = q0 + w * 64 + qt * 32 + l31;
    const bool ok = q < nq;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) v = *(const s16x8*)(Qb + q * 128 + ks * 16 + hi * 8);
      qf[qt][ks] = v;
    }
    // pass <=> s' > thr - eps_q ; +inf for padding queries
    cut[qt] = (ok && !BOOT) ? thr[q] - cbir_eps(qnorm, q, gmax_bits) : __uint_as_float(0x7f800000u);
  }
  float bm[2] = {__uint_as_float(0xff800000u), __uint_as_float(0xff800000u)};

  // wave-local flush of the staged survivors: rank every entry inside its query with LDS atomics, reserve each query's slots
  // with ONE global atomic per lane (lane L <-> the wave's L-th query), then scatter.  Wave-synchronous, no workgroup barrier.
#define CF_FLUSH()                                                                                                        \
  do {                                                                                                                    \
    f_cnt[w][lane] = 0;                                                                                                   \
    VDK_WAVE_LDS_SYNC();                                                                                                  \
    const unsigned ns_ = wcnt < CF_WE ? wcnt : CF_WE;   /* (entries beyond the staging went straight to the lists) */      \
    for (unsigned i_ = lane; i_ < ns_; i_ += 64) e_rank[w][i_] = (unsigned short)atomicAdd(&f_cnt[w][e_q[w][i_]], 1u);       \
    VDK_WAVE_LDS_SYNC();                                                                                                  \
    {                                                                                                                     \
      const unsigned c_ = f_cnt[w][lane];                                                                                 \
      f_base[w][lane] = c_ ? atomicAdd(&cand.cnt[qw0 + lane], c_) : 0u;                                                     \
    }                                                                                                                     \
    VDK_WAVE_LDS_SYNC();                                                                                                  \
    for (unsigned i_ = lane; i_ < ns_; i_ += 64) {                                                                        \
      const unsigned ql_ = e_q[w][i_];                                                                                    \
      const long pos_ = (long)f_base[w][ql_] + e_rank[w][i_];                                                             \
      if (pos_ < cand.cap) { cand.idx[(qw0 + ql_) * cand.cap + pos_] = e_idx[w][i_]; cand.score[(qw0 + ql_) * cand.cap + pos_] = e_score[w][i_]; } \
      else atomicOr(cand.overflow, 1u);                                                                                   \
    }                                                                                                                     \
    VDK_WAVE_LDS_SYNC();                                                                                                  \
    wcnt = 0;                                                                                                             \
  } while (0)
  const long ntile = (r_end - r_begin + CF_BG - 1) / CF_BG;
  // DMA one 128-row tile (4 x 1 KB per wave): LDS slot (row, cp) holds chunk cp ^ (row & 15) of that row
#define CF_ISSUE(buf, t)                                                                                                  \
  do {                                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
      const int L = j * 512 + w * 64 + lane;                                                                              \
      const int row = L >> 4, cp = L & 15, c = cp ^ (row & 15);                                                           \
      long grow = r_begin + (long)(t) * CF_BG + row;                                                                      \
      if (grow > r_end - 1) grow = r_end - 1;                                                                             \
      const bf16_t* src = Gb + grow * 128 + c * 8;                                                                        \
      unsigned char* dst = Gs + (buf) * (CF_BG * 256) + (j * 512 + w * 64) * 16;                                          \
      __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(src), VDK_LDS_PTR(dst), 16, 0, 0);                                   \
    }                                                                                                                     \
  } while (0)
  // A fragments run two k-steps ahead of the MFMAs that consume them (hipcc alone emits read -> wait -> 4 MFMAs per k-step, exposing the LDS latency 16 times per
  // tile); the first two of the second half are fetched during the first half's last k-steps.
#define CF_ALD(hh, ks, rt) (*(const s16x8*)(Gt + ((hh) * 64 + (rt) * 32 + l31) * 256 + ((((ks) * 2 + hi) ^ (l31 & 15)) * 16)))
  // the 32 MFMAs of one 64-row half of the tile at Gt: ACC[row block][query block] = rows x this wave's 64 queries
#define CF_MFMA_HALF(H, ACC)                                                                                              \
  do {                                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                                     \
      const int step = (H) * 8 + ks;           /* 0..15 over the tile */                                                  \
      if (step + 2 < 16) {                                                                                                \
        const int nh = (step + 2) >> 3, nks = (step + 2) & 7;                                                             \
        af[(step + 2) % 3][0] = CF_ALD(nh, nks, 0);                                                                       \
        af[(step + 2) % 3][1] = CF_ALD(nh, nks, 1);                                                                       \
      }                                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);   /* keep the reads up here: the scheduler otherwise sinks them next to their use */ \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                                     \
        _Pragma("unroll") for (int qt = 0; qt < 2; ++qt)   /* (the first k-step starts from the zero block: no 64 v_mov per half) */ \
          (ACC)[rt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step % 3][rt], qf[qt][ks], ks == 0 ? zero16 : (ACC)[rt][qt], 0, 0, 0);   \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
    }                                                                                                                     \
  } while (0)
  // the filter of one half (tile TT, half H): per 32 x 32 block the lane's maximum against its cut -- a cheap reject; the survivors' slow path is rare after the first stages
#define CF_FILTER_HALF(H, ACC, TT, LASTT)                                                                                 \
  do {                                                                                                                    \
    const long row0_ = r_begin + (TT) * CF_BG + (H) * 64;                                                                  \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                                       \
      _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                                                   \
        if (BOOT) {                                                                                                       \
          float m = (ACC)[rt][qt][0];                                                                                     \
          _Pragma("unroll") for (int r = 1; r < 16; ++r) m = fmaxf(m, (ACC)[rt][qt][r]);                                   \
          bm[qt] = fmaxf(bm[qt], m);   /* rows past r_end are copies of row r_end - 1 (clamped DMA): the maximum is unaffected */ \
          if ((H) == 1 && rt == 1) {   /* tile complete: one group maximum per (query, 128-row tile) */                    \
            const float tm = fmaxf(bm[qt], __shfl_xor(bm[qt], 32));                                                       \
            const long q = q0 + w * 64 + qt * 32 + l31;                                                                   \
            if (hi == 0 && q < nq) gm[q * gm_ld + (r_begin - g_begin) / CF_BG + (TT)] = tm;                                \
            bm[qt] = __uint_as_float(0xff800000u);                                                                        \
          }                                                                                                               \
        } else {                                                                                                          \
          float m = (ACC)[rt][qt][0];                                                                                     \
          _Pragma("unroll") for (int r = 1; r < 16; ++r) m = fmaxf(m, (ACC)[rt][qt][r]);                                   \
          if (__any(m > cut[qt])) CF_STAGE((ACC)[rt][qt], cut[qt], qt * 32 + l31, row0_ + rt * 32, (LASTT));               \
        }                                                                                                                 \
      }                                                                                                                   \
  } while (0)
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  CF_ISSUE(0, 0);
  if (ntile > 1) CF_ISSUE(1, 1);
  if (ntile > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 4); else __builtin_amdgcn_s_waitcnt(0x0F70);   // tile 0 landed
  __syncthreads();
  int cur = 0;
  // Two waves share a SIMD (wave w and w + 4) and one matrix pipe.  Left in step by the tile barrier, both multiply (contending for the pipe) and then both filter
  // (the pipe idle).  With `phased` the second group passes the tile barrier BEFORE its second filter instead of after it: the program order is the same
  // (M(h0) F(h0) M(h1) F(h1), the filter works on registers only), but the group then runs half a tile behind in role --
  //     group A:  M(h0)  F(h0)  M(h1)  F(h1) | barrier        group B:  F(prev h1)  M(h0)  F(h0)  M(h1) | barrier
  // so in every slot one wave of a SIMD owns the matrix pipe and the other the vector ALU.
  const bool grp_b = !BOOT && phased && w >= 4;
  f32x16 acc[2][2];
  for (long t = 0; t < ntile; ++t) {
    // ring slot (cur + 2) % 3 was consumed during iteration t - 1 (everybody passed the barrier that ended it)
    int nxt2 = cur + 2; if (nxt2 >= CF_NBUF) nxt2 -= CF_NBUF;
    if (t + 2 < ntile) CF_ISSUE(nxt2, t + 2);
    const unsigned char* Gt = Gs + cur * (CF_BG * 256);
    s16x8 af[3][2];
    af[0][0] = CF_ALD(0, 0, 0); af[0][1] = CF_ALD(0, 0, 1);
    af[1][0] = CF_ALD(0, 1, 0); af[1][1] = CF_ALD(0, 1, 1);
    CF_MFMA_HALF(0, acc);
    CF_FILTER_HALF(0, acc, t, t == ntile - 1);
    CF_MFMA_HALF(1, acc);
    // tile t + 1 must have landed: only tile t + 2's four DMAs may still be in flight
    if (grp_b) {
      if (t + 2 < ntile) __builtin_amdgcn_s_waitcnt(0x0F70 | 4); else __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    CF_FILTER_HALF(1, acc, t, t == ntile - 1);
    if (!grp_b) {
      if (t + 2 < ntile) __builtin_amdgcn_s_waitcnt(0x0F70 | 4); else __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    cur = cur + 1 == CF_NBUF ? 0 : cur + 1;
  }
#undef CF_MFMA_HALF
#undef CF_FILTER_HALF
#undef CF_ALD
#undef CF_ISSUE
  if (!BOOT && wcnt) { CF_FLUSH(); }
}

// The same filter for 128 < D <= 512 (face embeddings: feat_dim 512, models/faceX/backbone/timm_wrapper.py:33-47).  NK = DP / 16 k-steps (16, 24 or 32): a wave
// keeps ONE 32-query tile's fragments (4 NK registers) and a ring slot holds a 32-row gallery tile (32 x DP x 2 B <= 32 KB), so a workgroup covers 256 queries.
// Every wave reads the whole tile for NK MFMAs -- half the operand reuse of the D <= 128 kernel, LDS-bound near 50 % of its MFMA rate, still ~5x the all-pairs
// fp32 scan these dimensions fell back to.  BOOT: gm[q][tile] per 32-row tile.
template <bool BOOT, int NK>
__global__ __launch_bounds__(512) void cbir_prefilter_wide_kernel(const bf16_t* __restrict__ Qb, const float* __restrict__ qnorm, long nq,
                                                                  const bf16_t* __restrict__ Gb, const unsigned* __restrict__ gmax_bits, long g_begin,
                                                                  long g_end, long rows_per_split, int nsplit, long idx_base,
                                                                  const float* __restrict__ thr, CbirCand cand, float* __restrict__ gm, long gm_ld) {
  constexpr int RB = NK * 32;           // bytes per row
  constexpr int CH = RB / 16;           // 16-byte chunks per row
  constexpr int TILEB = 32 * RB;        // one ring slot
  __shared__ __attribute__((aligned(16))) unsigned char Gs[CF_NBUF * TILEB];
  __shared__ int e_idx[8][CF_WE];
  __shared__ float e_score[8][CF_WE];          // the approximate score s' of the survivor (the approximate-ranking schedule ranks on it; the exact schedules ignore it)
  __shared__ unsigned short e_rank[8][CF_WE];
  __shared__ unsigned char e_q[8][CF_WE];
  __shared__ unsigned f_cnt[8][64], f_base[8][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const unsigned long long lane_lt = (1ull << lane) - 1ull;
  unsigned wcnt = 0;
  const int split = blockIdx.x % nsplit;
  const long q0 = (long)(blockIdx.x / nsplit) * 256;
  const long r_begin = g_begin + (long)split * rows_per_split;
  long r_end = r_begin + rows_per_split;
  if (r_end > g_end) r_end = g_end;
  if (r_begin >= r_end) return;
  const long qw0 = q0 + w * 32;
  const long q = qw0 + l31;
  const bool ok = q < nq;
  s16x8 qf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) v = *(const s16x8*)(Qb + q * (long)(RB / 2) + ks * 16 + hi * 8);
    qf[ks] = v;
  }
  const float cut = (ok && !BOOT) ? thr[q] - cbir_eps(qnorm, q, gmax_bits) : __uint_as_float(0x7f800000u);
  const long ntile = (r_end - r_begin + 31) / 32;
  // DMA one 32-row tile: NK / 8 instructions per wave; LDS slot (row, cp) holds chunk cp ^ (row & 15) of that row
#define CW_ISSUE(buf, t)                                                                                                  \
  do {                                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < NK / 8; ++j) {                                                                   \
      const int L = j * 512 + w * 64 + lane;                                                                              \
      const int row = L / CH, cp = L % CH, c = cp ^ (row & 15);                                                           \
      long grow = r_begin + (long)(t) * 32 + row;                                                                         \
      if (grow > r_end - 1) grow = r_end - 1;                                                                             \
      const bf16_t* src = Gb + grow * (long)(RB / 2) + c * 8;                                                             \
      unsigned char* dst = Gs + (buf) * TILEB + (j * 512 + w * 64) * 16;                                                  \
      __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(src), VDK_LDS_PTR(dst), 16, 0, 0);                                   \
    }                                                                                                                     \
  } while (0)
  CW_ISSUE(0, 0);
  if (ntile > 1) CW_ISSUE(1, 1);
  if (ntile > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (NK / 8)); else __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  int cur = 0;
  for (long t = 0; t < ntile; ++t) {
    int nxt2 = cur + 2; if (nxt2 >= CF_NBUF) nxt2 -= CF_NBUF;
    if (t + 2 < ntile) CW_ISSUE(nxt2, t + 2);
    const unsigned char* Gt = Gs + cur * TILEB + l31 * RB;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < NK; ks += 2) {      // two accumulators: consecutive MFMAs do not wait for each other
      const s16x8 a0 = *(const s16x8*)(Gt + (((ks * 2 + hi) ^ (l31 & 15)) << 4));
      const s16x8 a1 = *(const s16x8*)(Gt + ((((ks + 1) * 2 + hi) ^ (l31 & 15)) << 4));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[ks], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[ks + 1], acc1, 0, 0, 0);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc0[r] + acc1[r];
    const long row0 = r_begin + t * 32;
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    if (BOOT) {
      const float tm = fmaxf(m, __shfl_xor(m, 32));
      if (hi == 0 && ok) gm[q * gm_ld + (r_begin - g_begin) / 32 + t] = tm;
    } else if (__any(m > cut)) {
      CF_STAGE(acc, cut, l31, row0, t == ntile - 1);
    }
    if (t + 2 < ntile) __builtin_amdgcn_s_waitcnt(0x0F70 | (NK / 8)); else __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    cur = cur + 1 == CF_NBUF ? 0 : cur + 1;
  }
#undef CW_ISSUE
  if (!BOOT && wcnt) { CF_FLUSH(); }
}
#undef CF_FLUSH
#undef CF_STAGE
#undef CF_EMIT_OWNED
#undef CF_ROWBIT

// ---- approximate ranking between the stages (rank_mode 1 of vdk_cbir_search_fast2: schedule <= -1024) ---------------------------------------------------------------
// The exact schedules re-score every survivor of every stage with the fp32 fmaf chain: ~650 gathered gallery rows per query over a 10^6-row scan, 3.3 of the 4.6 GB a
// search moved (PMC, round 4/5).  Here a stage's survivors are ranked on the approximate score s' the pre-filter already holds (|s' - s| <= eps_q):
//   * k rows with s' >= kth(s') each have s >= kth(s') - eps, so L = kth(s') - eps is a lower bound of the exact k-th best score: thr[q] = L, and the next stage's
//     filter s' > thr - eps is the same test the exact schedules apply;
//   * a row can only belong to the exact top-k if s >= L, i.e. s' >= kth(s') - 2 eps: every such row is KEPT between the stages (k + a band of a few dozen rows on
//     continuous data), nothing else is;
//   * only the rows kept at the END get the exact fmaf chain (~130 per query instead of ~650 over the scan), are sorted on (exact score desc, index asc)
//     and the first k leave: the kept set contains the exact top-k, so the output is bit-identical to the exact schedules'.
// More than CA_KEEP rows inside the band (masses of near-duplicates) raises the overflow flag: the caller repeats with an exact schedule.  k <= 256.
// Between the stages nothing is sorted: a wave holds a query's entries in registers (8 per lane), finds the k-th largest s' by a most-significant-bit-first
// selection on the ordered bit patterns (32 steps of "how many keys >= prefix | bit": compares into scalar masks, popcounts on the scalar unit), keeps what is inside
// the band and writes it back compacted -- no LDS, ~2 us per query, 32 waves per CU.  Only the last stage sorts (exact keys, wave-synchronous bitonic steps in LDS).
#define CA_SLOTS 512     // entries a wave ranks per pass (kept + new; longer lists take more passes)
#define CA_KEEP 448      // kept entries per query between the stages
// the largest v with |{keys >= v}| >= k (keys: 8 per lane, 0 = empty slot); k >= 1
__device__ __forceinline__ unsigned cbir_wave_kth(const unsigned (&key)[CA_SLOTS / 64], unsigned k) {
  unsigned prefix = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned c = prefix | (1u << bit);
    unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < CA_SLOTS / 64; ++j) cnt += (unsigned)__popcll(__ballot(key[j] >= c));
    if (cnt >= k) prefix = c;
  }
  return prefix;
}
template <bool FINAL>
__global__ __launch_bounds__(256) void cbir_rank_approx_kernel(const float* __restrict__ Q, const float* __restrict__ G, int D, long idx_base, CbirCand cand, long nq, int k,
                                                               float* __restrict__ thr, float* __restrict__ out_score, long long* __restrict__ out_idx,
                                                               unsigned* __restrict__ carry, int g_half, const float* __restrict__ qnorm,
                                                               const unsigned* __restrict__ gmax_bits) {
  __shared__ unsigned long long keys_all[FINAL ? 4 : 1][FINAL ? CA_SLOTS : 1];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long q = (long)blockIdx.x * 4 + w;
  if (q >= nq) return;   // wave-uniform
  const unsigned long long lane_lt = (1ull << lane) - 1ull;
  unsigned n_all = cand.cnt[q];
  if ((long)n_all > cand.cap) n_all = (unsigned)cand.cap;      // (the pre-filter has raised the overflow flag)
  float* cs = cand.score + q * cand.cap;
  int* ci = cand.idx + q * cand.cap;
  const float eps = cbir_eps(qnorm, q, gmax_bits);
  float kth = __uint_as_float(0xff800000u);
  unsigned have = 0, consumed = 0;
  bool over = false;
  float sc[CA_SLOTS / 64];
  int gi[CA_SLOTS / 64];
  bool keep[CA_SLOTS / 64];
  // entries [0, carry) are the rows kept by the previous stage, the rest this stage's survivors: all carry s' and are ranked alike.  A pass takes the kept rows
  // (slots [0, have)) and as many unread ones as fit; one pass in practice.
  do {
    unsigned take = n_all - consumed;
    if (take > CA_SLOTS - have) take = CA_SLOTS - have;
    const unsigned n = have + take;
    unsigned key[CA_SLOTS / 64];
#pragma unroll
    for (int j = 0; j < CA_SLOTS / 64; ++j) {
      const unsigned e = lane + 64u * j;                        // logical entry: kept ones first
      key[j] = 0u; sc[j] = 0.f; gi[j] = 0; keep[j] = e < n;
      if (e < n) {
        const unsigned slot = e < have ? e : consumed + (e - have);
        sc[j] = cs[slot]; gi[j] = ci[slot];
        key[j] = f2ord(sc[j]) | 1u;                             // (never 0: an empty slot's key; the last bit of the order is irrelevant inside a 2 eps band)
      }
    }
    consumed += take;
    have = n;
    if (n >= (unsigned)k) {
      kth = fmaxf(kth, ord2f(cbir_wave_kth(key, (unsigned)k) & ~1u));      // (bit 0 cleared: never above the true k-th best)
      const float bound = kth - 2.0f * eps;
      unsigned base = 0;
#pragma unroll
      for (int j = 0; j < CA_SLOTS / 64; ++j) {
        keep[j] = keep[j] && !(sc[j] < bound);
        const unsigned long long mk = __ballot(keep[j]);
        const unsigned pos = base + (unsigned)__popcll(mk & lane_lt);
        if (keep[j] && pos >= CA_KEEP) { keep[j] = false; over = true; }
        if (keep[j] && (!FINAL || consumed < n_all)) { cs[pos] = sc[j]; ci[pos] = gi[j]; }      // compacted to the front (every slot read above is in registers)
        base += (unsigned)__popcll(mk);
      }
      over = __any(over);
      have = base > CA_KEEP ? CA_KEEP : base;
    }
  } while (consumed < n_all);
  if (over && lane == 0) atomicOr(cand.overflow, 1u);
  if (!FINAL) {
    if (lane == 0) {
      carry[q] = have;
      cand.cnt[q] = have;
      if (have >= (unsigned)k) thr[q] = fmaxf(thr[q], kth - eps);
    }
    return;
  }
  // the end of the scan: exact scores of the kept rows (the oracle's k-ordered fmaf chain), exact order, first k out
  unsigned long long* K = keys_all[w];
  const float* qrow = Q + q * (long)D;
  unsigned base = 0;
#pragma unroll
  for (int j = 0; j < CA_SLOTS / 64; ++j) {
    const unsigned long long mk = __ballot(keep[j]);
    if (keep[j]) K[base + (unsigned)__popcll(mk & lane_lt)] = cbir_key(cbir_exact_ip_any(qrow, G, (long)gi[j] - idx_base, D, g_half), gi[j]);
    base += (unsigned)__popcll(mk);
  }
  have = base;
  unsigned sortn = 64; while (sortn < have) sortn <<= 1;
  for (unsigned i = have + lane; i < sortn; i += 64) K[i] = ~0ull;
  VDK_WAVE_LDS_SYNC();
  for (unsigned size = 2; size <= sortn; size <<= 1)
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned p = lane; p < (sortn >> 1); p += 64) {
        const unsigned lo = ((p / stride) * 2 * stride) + (p % stride), hi2 = lo + stride;
        const bool asc = ((lo & size) == 0);
        const unsigned long long a = K[lo], b = K[hi2];
        if ((a > b) == asc) { K[lo] = b; K[hi2] = a; }
      }
      VDK_WAVE_LDS_SYNC();
    }
  for (int i = lane; i < k; i += 64) {
    if ((unsigned)i < have) {
      const unsigned long long key = K[i];
      out_score[q * k + i] = ord2f(~(unsigned)(key >> 32));
      out_idx[q * k + i] = (long long)(int)(unsigned)(key & 0xffffffffu);
    } else {
      out_score[q * k + i] = -3.4028234663852886e38f;
      out_idx[q * k + i] = -1;
    }
  }
}

// thr[q] = (k-th largest of the G group maxima) - eps_q (see BOOT above).  One workgroup per query, bitonic sort in LDS.
#define CF_BOOT_MAXG 4096
__global__ __launch_bounds__(256) void cbir_boot_thr_kernel(const float* __restrict__ gm, int G, int k, const float* __restrict__ qnorm,
                                                            const unsigned* __restrict__ gmax_bits, float* __restrict__ thr) {
  __shared__ unsigned keys[CF_BOOT_MAXG];
  const long q = blockIdx.x;
  const int tid = threadIdx.x;
  unsigned sortn = 64; while (sortn < (unsigned)G) sortn <<= 1;
  for (unsigned i = tid; i < sortn; i += 256) keys[i] = i < (unsigned)G ? ~f2ord(gm[q * G + i]) : 0xffffffffu;   // ascending key = descending score
  __syncthreads();
  for (unsigned size = 2; size <= sortn; size <<= 1)
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned p = tid; p < (sortn >> 1); p += 256) {
        const unsigned lo = ((p / stride) * 2 * stride) + (p % stride), hi2 = lo + stride;
        const bool asc = ((lo & size) == 0);
        const unsigned a = keys[lo], b = keys[hi2];
        if ((a > b) == asc) { keys[lo] = b; keys[hi2] = a; }
      }
      __syncthreads();
    }
  if (tid == 0) thr[q] = ord2f(~keys[k - 1]) - cbir_eps(qnorm, q, gmax_bits);
}

// the same threshold for G <= 1024 group maxima (k <= 256 and the default sample of 4 k tiles): a wave per query, the maxima in registers, the k-th largest by the
// bitwise selection of cbir_rank_approx_kernel -- no sort, no LDS, no workgroup barrier (the sorting kernel above: 67-88 us at 10 k queries)
__global__ __launch_bounds__(256) void cbir_boot_thr_wave_kernel(const float* __restrict__ gm, int G, int k, const float* __restrict__ qnorm,
                                                                 const unsigned* __restrict__ gmax_bits, float* __restrict__ thr, long nq) {
  const int lane = threadIdx.x & 63;
  const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;   // wave-uniform
  unsigned key[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int i = lane + 64 * j; key[j] = i < G ? f2ord(gm[q * G + i]) : 0u; }
  unsigned prefix = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned c = prefix | (1u << bit);
    unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) cnt += (unsigned)__popcll(__ballot(key[j] >= c));
    if (cnt >= (unsigned)k) prefix = c;
  }
  if (lane == 0) thr[q] = ord2f(prefix) - cbir_eps(qnorm, q, gmax_bits);
}

// exact scores of the new candidates of every query: entries [carry[q], cnt[q]) of its list.  One workgroup per query; one
// candidate per thread, k-ordered fmaf chain from +0 (bit-identical to oracle_ip_pair and to the fp32-MFMA scan).
__global__ __launch_bounds__(256) void cbir_rescore_kernel(const float* __restrict__ Q, const float* __restrict__ G, int D, long idx_base, CbirCand cand,
                                                           const unsigned* __restrict__ carry, int g_half) {
  __shared__ __attribute__((aligned(16))) float qs[512];
  const long q = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) qs[c] = Q[q * (long)D + c];
  __syncthreads();
  unsigned n = cand.cnt[q];
  if ((long)n > cand.cap) n = (unsigned)cand.cap;
  for (unsigned i = carry[q] + threadIdx.x; i < n; i += 256) {
    cand.score[q * cand.cap + i] = cbir_exact_ip_any(qs, G, (long)cand.idx[q * cand.cap + i] - idx_base, D, g_half);
  }
}

__global__ void cbir_init_kernel(unsigned* cnt, float* thr, unsigned* overflow, long nq) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) { cnt[i] = 0; thr[i] = __uint_as_float(0xff800000u); }
  if (i == 0) *overflow = 0;
}

// gather S shard-local top-k lists [S][nq][k] into the candidate lists (multi-GPU merge, C5)
__global__ void cbir_fill_from_lists_kernel(CbirCand cand, const float* __restrict__ scores,
                                            const long long* __restrict__ idx, int S, long nq, int k) {
  long q = blockIdx.x;
  unsigned n = 0;
  for (int i = threadIdx.x; i < S * k; i += blockDim.x) {
    int s = i / k, j = i % k;
    long long id = idx[((long)s * nq + q) * k + j];
    if (id >= 0) {
      unsigned pos = atomicAdd(&cand.cnt[q], 1u);
      cand.score[q * cand.cap + pos] = scores[((long)s * nq + q) * k + j];
      cand.idx[q * cand.cap + pos] = (int)id;
    }
  }
  (void)n;
}

// ---- second stream + events of the pipelined schedule (created once per process; timing disabled) ------------------------------------------------------
// (per calling thread AND per device: a stream / event belongs to the device that was current when it was created; one process per GPU is the usual host, a multi-device
//  process gets one set per device)
#include <map>
#include <vector>
struct CbSide { hipStream_t s2 = nullptr; std::vector<hipEvent_t> ev; };
static thread_local std::map<int, CbSide> g_cb_side;
static CbSide& cb_side() { int dev = 0; (void)hipGetDevice(&dev); return g_cb_side[dev]; }
#define g_cb_s2 (cb_side().s2)
#define g_cb_ev (cb_side().ev)
static int g_cb_pipeline = -1;
static bool cb_pipeline_enabled() {
  // default OFF: measured on the MI355X (10 k x 1 M x 128, k = 100) the pipelined schedule is SLOWER, 5.1 ms against 3.6 - 3.85 ms sequential -- the ranking
  // workgroups take CU slots and LDS from the one-workgroup-per-CU scan (its launches average 276 us instead of 171 us) and the one-stage-older thresholds let
  // more rows survive.  Kept behind VDK_CBIR_PIPELINE=1 with its tests (bit-identical results).
  if (g_cb_pipeline < 0) { const char* e = getenv("VDK_CBIR_PIPELINE"); g_cb_pipeline = (e && e[0] == '1') ? 1 : 0; }
  return g_cb_pipeline == 1;
}
static hipStream_t cb_side_stream() {
  if (!g_cb_s2 && hipStreamCreateWithFlags(&g_cb_s2, hipStreamNonBlocking) != hipSuccess) g_cb_s2 = nullptr;
  return g_cb_s2;
}
static int cb_event(size_t i, hipEvent_t* e) {
  while (g_cb_ev.size() <= i) {
    hipEvent_t x;
    if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) return 1;
    g_cb_ev.push_back(x);
  }
  *e = g_cb_ev[i];
  return 0;
}
static int cb_record(size_t slot, hipStream_t on) { hipEvent_t e; if (cb_event(slot, &e)) return 1; return hipEventRecord(e, on) != hipSuccess; }
static int cb_wait(size_t slot, hipStream_t who) { hipEvent_t e; if (cb_event(slot, &e)) return 1; return hipStreamWaitEvent(who, e, 0) != hipSuccess; }
static int cb_order(size_t slot, hipStream_t from, hipStream_t to) { return cb_record(slot, from) || cb_wait(slot, to); }
__global__ void cbir_arm_kernel(unsigned* __restrict__ cnt_a, unsigned* __restrict__ cnt_b, long nq, unsigned reserve) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) { cnt_a[i] = reserve; cnt_b[i] = reserve; }
}

// wave groups of the D <= 128 pre-filter in opposite roles (see the kernel); VDK_CBIR_PHASED=0 restores the in-step form (A/B)
static int cb_phased() { static const int v = [] { const char* e = getenv("VDK_CBIR_PHASED"); return (e && e[0] == '0') ? 0 : 1; }(); return v; }
template <bool BOOT>
static void cb_launch_prefilter(hipStream_t stream, int DP, unsigned grid, const bf16_t* Qb, const float* qnorm, long nq, const bf16_t* Gb, const unsigned* gmax_bits, long begin,
                                long end, long rps, int nsplit, long idx_base, const float* thr, const CbirCand& cand, float* gm, long gm_ld) {
  switch (DP) {
    case 128: hipLaunchKernelGGL(cbir_prefilter_kernel<BOOT>, dim3(grid), dim3(512), 0, stream, Qb, qnorm, nq, Gb, gmax_bits, begin, end, rps, nsplit, idx_base, thr, cand, gm, gm_ld, cb_phased()); break;
    case 256: hipLaunchKernelGGL((cbir_prefilter_wide_kernel<BOOT, 16>), dim3(grid), dim3(512), 0, stream, Qb, qnorm, nq, Gb, gmax_bits, begin, end, rps, nsplit, idx_base, thr, cand, gm, gm_ld); break;
    case 384: hipLaunchKernelGGL((cbir_prefilter_wide_kernel<BOOT, 24>), dim3(grid), dim3(512), 0, stream, Qb, qnorm, nq, Gb, gmax_bits, begin, end, rps, nsplit, idx_base, thr, cand, gm, gm_ld); break;
    default: hipLaunchKernelGGL((cbir_prefilter_wide_kernel<BOOT, 32>), dim3(grid), dim3(512), 0, stream, Qb, qnorm, nq, Gb, gmax_bits, begin, end, rps, nsplit, idx_base, thr, cand, gm, gm_ld); break;
  }
}

// ------------------------------------------------------------------------------------ host side
extern "C" {

int vdk_l2norm_rows(const float* x, float* out, int64_t n, int32_t d, float eps, void* stream) {
  if (!x || !out || n < 0 || d <= 0) return vdk_fail(VDK_EINVAL, "vdk_l2norm_rows: bad argument");
  if (n == 0) return VDK_OK;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long)n, (int)d, eps);
  return vdk_check_launch("l2norm_rows_kernel");
}

// rank every query's candidate list: wave-per-query kernel for k <= 256, workgroup-per-query kernels above.
// Q != nullptr: entries [carry[q], cnt[q]) carry only an index and are re-scored exactly first (prefilter path).
static void cb_rank(hipStream_t stream, const float* Q, const float* G, int D, long idx_base, const CbirCand& cand, long nq, int k, float* thr,
                    float* out_scores, long long* out_idx, int write_out, unsigned* carry, int g_half = 0, const CbirCand* dst = nullptr, int reserve = 0) {
  if (k <= 256) {
    hipLaunchKernelGGL(cbir_rank_wave_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, Q, G, D, idx_base, cand, nq, k, thr, out_scores, out_idx,
                       write_out, carry, g_half, dst ? *dst : cand, reserve);
  } else {
    if (Q) hipLaunchKernelGGL(cbir_rescore_kernel, dim3((unsigned)nq), dim3(256), 0, stream, Q, G, D, idx_base, cand, (const unsigned*)carry, g_half);
    hipLaunchKernelGGL(cbir_select_kernel, dim3((unsigned)nq), dim3(256), 0, stream, cand, nq, k, thr, out_scores, out_idx, write_out, carry);
  }
}

static inline size_t cb_align(size_t x) { return (x + 255) & ~(size_t)255; }

int vdk_cbir_workspace_bytes(int64_t nq, int32_t k, int64_t cap, size_t* bytes) {
  if (!bytes || nq < 0 || k <= 0 || k > 1024 || cap < 2 * (int64_t)k) return vdk_fail(VDK_EINVAL, "vdk_cbir_workspace_bytes: bad argument");
  size_t b = 0;
  b += cb_align((size_t)nq * cap * 4);  // score
  b += cb_align((size_t)nq * cap * 4);  // idx
  b += cb_align((size_t)nq * 4);        // cnt
  b += cb_align((size_t)nq * 4);        // thr
  b += cb_align(4);                     // overflow
  *bytes = b;
  return VDK_OK;
}

static int cb_carve(void* ws, size_t ws_bytes, int64_t nq, int32_t k, int64_t cap, CbirCand* c, float** thr) {
  size_t need = 0;
  int rc = vdk_cbir_workspace_bytes(nq, k, cap, &need);
  if (rc) return rc;
  if (!ws || ws_bytes < need) return vdk_fail(VDK_EWORKSPACE, "vdk_cbir: workspace too small");
  char* p = (char*)ws;
  c->score = (float*)p; p += cb_align((size_t)nq * cap * 4);
  c->idx = (int*)p; p += cb_align((size_t)nq * cap * 4);
  c->cnt = (unsigned*)p; p += cb_align((size_t)nq * 4);
  *thr = (float*)p; p += cb_align((size_t)nq * 4);
  c->overflow = (unsigned*)p;
  c->cap = cap;
  return VDK_OK;
}

// Exact inner-product top-k of Q [nq,D] against gallery rows G [N,D] (both fp32 row-major, device).
// out_scores [nq,k] descending, out_idx [nq,k] = idx_base + row (ties: lower index first), padded with
// (-FLT_MAX, -1) when N < k.  D % 4 == 0.  Stream-ordered, no allocation, no host sync.
int vdk_cbir_search(const float* Q, int64_t nq, const float* G, int64_t N, int32_t D, int32_t k, int64_t idx_base,
                    float* out_scores, int64_t* out_idx, int64_t cap, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!Q || (!G && N > 0) || !out_scores || !out_idx || nq < 0 || N < 0 || D <= 0 || (D & 3) || k <= 0 || k > 1024)
    return vdk_fail(VDK_EINVAL, "vdk_cbir_search: bad argument (need D % 4 == 0, 1 <= k <= 1024)");
  if (idx_base + N > 0x7fffffffLL) return vdk_fail(VDK_EINVAL, "vdk_cbir_search: index range exceeds int32");
  if (nq == 0) return VDK_OK;
  CbirCand cand; float* thr;
  int rc = cb_carve(ws, ws_bytes, nq, k, cap, &cand, &thr);
  if (rc) return rc;
  hipLaunchKernelGGL(cbir_init_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, cand.cnt, thr,
                     cand.overflow, (long)nq);
  const long qblocks = (long)((nq + CB_BQ - 1) / CB_BQ);
  const long max_stage = cap - k;  // no-overflow guarantee: carry (<= k) + every row of the stage
  long begin = 0;
  long stage = 512;
  if (stage > max_stage) stage = max_stage;
  while (begin < N) {
    long end = begin + stage; if (end > N) end = N;
    long rows = end - begin;
    long tiles = (rows + CB_BG - 1) / CB_BG;
    int nsplit = 8;
    while (nsplit > 1 && tiles < nsplit) nsplit >>= 1;
    // more splits when there are few query blocks, to fill 256 CUs
    while ((long)nsplit * qblocks < 512 && (long)nsplit * 2 <= tiles && nsplit < 64) nsplit <<= 1;
    long rps = ((tiles + nsplit - 1) / nsplit) * CB_BG;
    hipLaunchKernelGGL(cbir_score_filter_kernel, dim3((unsigned)(qblocks * nsplit)), dim3(512), 0, stream, Q, (long)nq, G,
                       (int)D, begin, end, rps, nsplit, (long)idx_base, (const float*)thr, cand);
    cb_rank(stream, nullptr, nullptr, 0, 0, cand, (long)nq, (int)k, thr, out_scores, (long long*)out_idx, (int)(end == N), nullptr);
    begin = end;
    stage *= 8;
    if (stage > max_stage) stage = max_stage;
  }
  if (N == 0)
    cb_rank(stream, nullptr, nullptr, 0, 0, cand, (long)nq, (int)k, thr, out_scores, (long long*)out_idx, 1, nullptr);
  return vdk_check_launch("vdk_cbir_search");
}

// ---- fast path host side -----------------------------------------------------------------------------------------------
static inline int cb_dp(int D) { return (D + 127) / 128 * 128; }   // padded width of the bf16 copies
static bool cb_pipeline_enabled();

// Per-index preparation (IndexFlatIP.add time, not per search): Gb = bf16 [N, DP] zero-padded copy of G (DP = D rounded up to 128), gmax_bits[0..2] = bit
// patterns of max_n (||g||, ||bf16(g)||, ||bf16(g) - g||), gmax_bits[3] = DP / 128.  gnorm_ws: f32 [3 N] scratch.  D <= 512.
int vdk_cbir_prepare_gallery(const float* G, int64_t N, int32_t D, void* Gb, float* gnorm_ws, uint32_t* gmax_bits, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((!G && N > 0) || !Gb || !gnorm_ws || !gmax_bits || N < 0 || D <= 0 || D > 512) return vdk_fail(VDK_EINVAL, "vdk_cbir_prepare_gallery: bad argument (D <= 512)");
  const int DP = cb_dp(D);
  hipLaunchKernelGGL(cbir_gstat_init_kernel, dim3(1), dim3(64), 0, stream, (unsigned*)gmax_bits, (unsigned)(DP / 128));
  if (N == 0) return vdk_check_launch("vdk_cbir_prepare_gallery");
  hipLaunchKernelGGL(cbir_cast_rows_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, G, (long)N, (int)D, DP, (bf16_t*)Gb, gnorm_ws);
  hipLaunchKernelGGL(cbir_max3_kernel, dim3(256), dim3(256), 0, stream, (const float*)gnorm_ws, (long)N, (unsigned*)gmax_bits);
  return vdk_check_launch("vdk_cbir_prepare_gallery");
}

static size_t cb_fast_bytes(int64_t nq, int32_t k, int64_t cap, int DP) {
  size_t b = 0;
  vdk_cbir_workspace_bytes(nq, k, cap, &b);
  if (cb_pipeline_enabled()) b += cb_align((size_t)nq * cap * 4) + cb_align((size_t)nq * cap * 4) + cb_align((size_t)nq * 4);   // second list set of the pipelined schedule (score, idx, cnt)
  b += cb_align((size_t)nq * DP * 2);    // Qb
  b += cb_align((size_t)nq * 12);        // query norms (||q||, ||q~||, ||q~ - q||)
  b += cb_align((size_t)nq * 4);         // carry
  b += cb_align((size_t)nq * CF_BOOT_MAXG * 4);   // bootstrap group maxima [nq, G <= 4096]
  return b;
}
int vdk_cbir_fast_workspace_bytes(int64_t nq, int32_t k, int64_t cap, size_t* bytes) {   // D <= 128
  size_t b = 0;
  int rc = vdk_cbir_workspace_bytes(nq, k, cap, &b);
  if (rc) return rc;
  *bytes = cb_fast_bytes(nq, k, cap, 128);
  return VDK_OK;
}
int vdk_cbir_fast2_workspace_bytes(int64_t nq, int32_t D, int32_t k, int64_t cap, size_t* bytes) {
  size_t b = 0;
  int rc = vdk_cbir_workspace_bytes(nq, k, cap, &b);
  if (rc) return rc;
  if (D <= 0 || D > 512) return vdk_fail(VDK_EINVAL, "vdk_cbir_fast2_workspace_bytes: D <= 512");
  *bytes = cb_fast_bytes(nq, k, cap, cb_dp(D));
  return VDK_OK;
}

// Bit-identical results to vdk_cbir_search for D <= 512: bf16 pre-filter with a rigorous error bound + exact re-scoring of the survivors.
//   g_dtype   VDK_F32: G is the fp32 gallery.  VDK_F16: G holds the rows in fp16 (faiss's useFloat16 storage, engine/cbir/evaluation.py:157-162); scores are the
//             k-ordered fp32 fmaf chain over those fp16 values (pass queries that are fp16-representable to reproduce faiss's fp16 x fp16 -> fp32 product).
//   schedule  0: guaranteed -- stages of at most cap - k rows, candidate lists cannot overflow whatever the data (16 prefilter + 16 rank launches at 1 M rows, cap 65 536).
//             1: optimistic -- bootstrap, one stage over the first eighth of the gallery, one over the rest (2 + 4 launches); the lists hold `cap` entries and
//                *overflow_out (device u32, may be NULL) is set when a query produced more survivors than that: the results are then INVALID and the caller
//                repeats the search with schedule 0 (visiondk_amd/cbir.py does).  Typical survivors per query: ~10^3 over a 10^6-row scan.
int vdk_cbir_search_fast2(const float* Q, int64_t nq, const void* G, int32_t g_dtype, const void* Gb, const uint32_t* gmax_bits, int64_t N, int32_t D, int32_t k,
                          int64_t idx_base, float* out_scores, int64_t* out_idx, int64_t cap, int32_t schedule, uint32_t* overflow_out, void* ws, size_t ws_bytes,
                          void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!Q || (!G && N > 0) || (!Gb && N > 0) || !gmax_bits || !out_scores || !out_idx || nq < 0 || N < 0 || D <= 0 || D > 512 || (D & 3) || k <= 0 || k > 1024)
    return vdk_fail(VDK_EINVAL, "vdk_cbir_search_fast2: bad argument (need D % 4 == 0, D <= 512, 1 <= k <= 1024)");
  if (g_dtype != VDK_F32 && g_dtype != VDK_F16) return vdk_fail(VDK_EINVAL, "vdk_cbir_search_fast2: g_dtype must be VDK_F32 or VDK_F16");
  if (g_dtype == VDK_F16 && (D & 7)) return vdk_fail(VDK_EINVAL, "vdk_cbir_search_fast2: fp16 storage needs D % 8 == 0");
  // schedule <= -1024: stages of -schedule rows like schedule >= 1024, RANKED ON THE APPROXIMATE SCORES between the stages (cbir_rank_approx_kernel); k <= 256
  const bool approx = schedule <= -1024;
  if (approx) { if (k > 256) return vdk_fail(VDK_EINVAL, "vdk_cbir_search_fast2: the approximate-ranking schedule serves k <= 256"); schedule = -schedule; }
  if (schedule < 0) return vdk_fail(VDK_EINVAL, "vdk_cbir_search_fast2: schedule must be 0, 1, >= 1024 or <= -1024");
  if (idx_base + N > 0x7fffffffLL) return vdk_fail(VDK_EINVAL, "vdk_cbir_search_fast2: index range exceeds int32");
  if (nq == 0) return VDK_OK;
  const int DP = cb_dp(D), g_half = g_dtype == VDK_F16;
  const int QB = DP == 128 ? CF_BQ : 256, BG = DP == 128 ? CF_BG : 32;      // queries per workgroup, rows per gallery tile
  size_t need = cb_fast_bytes(nq, k, cap, DP);
  if (!ws || ws_bytes < need) return vdk_fail(VDK_EWORKSPACE, "vdk_cbir_search_fast2: workspace too small");
  CbirCand cand; float* thr;
  int rc = cb_carve(ws, ws_bytes, nq, k, cap, &cand, &thr);
  if (rc) return rc;
  size_t base_bytes = 0; vdk_cbir_workspace_bytes(nq, k, cap, &base_bytes);
  char* p = (char*)ws + base_bytes;
  CbirCand cand2 = cand;                                   // second list set (pipelined schedule only)
  if (cb_pipeline_enabled()) {
    cand2.score = (float*)p; p += cb_align((size_t)nq * cap * 4);
    cand2.idx = (int*)p; p += cb_align((size_t)nq * cap * 4);
    cand2.cnt = (unsigned*)p; p += cb_align((size_t)nq * 4);
  }
  bf16_t* Qb = (bf16_t*)p; p += cb_align((size_t)nq * DP * 2);
  float* qnorm = (float*)p; p += cb_align((size_t)nq * 12);
  unsigned* carry = (unsigned*)p; p += cb_align((size_t)nq * 4);
  float* gm = (float*)p;
  const float* Gf = (const float*)G;
  hipLaunchKernelGGL(cbir_init_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, cand.cnt, thr, cand.overflow, (long)nq);
  if (hipMemsetAsync(carry, 0, (size_t)nq * 4, stream) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: memset failed");
  hipLaunchKernelGGL(cbir_cast_rows_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, Q, (long)nq, (int)D, DP, Qb, qnorm);
  const long qblocks = (long)((nq + QB - 1) / QB);
  // schedule >= 1024: staged like the guaranteed schedule, but the stage length is `schedule` rows whatever the list capacity: the lists hold `cap` entries (a stage yields
  // ~10^2 survivors per query behind a bootstrap threshold, not its 10^5 rows), overflow is REPORTED through *overflow_out and the caller repeats with schedule 0.
  const long max_stage = schedule >= 1024 ? (long)schedule : cap - k;
  long begin = 0, stage = 512;
  bool booted = false;
  // threshold bootstrap on a sample of NG full tiles, k <= NG <= min(4k, 4096)
  long NG = N / BG;
  static const long boot_mult = [] { const char* e = getenv("VDK_CBIR_BOOT_MULT"); const long v = e ? atol(e) : 0; return v > 0 ? v : 4L; }();   // sample = boot_mult * k tiles (tuning knob)
  if (NG > boot_mult * k) NG = boot_mult * k;
  if (NG > CF_BOOT_MAXG) NG = CF_BOOT_MAXG;
  if (NG >= k) {
    long ns = 256 / qblocks;
    if (ns > NG / 4) ns = NG / 4;
    if (ns < 1) ns = 1;
    const long rps = ((NG + ns - 1) / ns) * BG;
    cb_launch_prefilter<true>(stream, DP, (unsigned)(qblocks * ns), Qb, qnorm, (long)nq, (const bf16_t*)Gb, (const unsigned*)gmax_bits, 0L, NG * BG, rps, (int)ns, (long)idx_base,
                              thr, cand, gm, NG);
    if (NG <= 1024) hipLaunchKernelGGL(cbir_boot_thr_wave_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, (const float*)gm, (int)NG, (int)k, (const float*)qnorm,
                                       (const unsigned*)gmax_bits, thr, (long)nq);
    else hipLaunchKernelGGL(cbir_boot_thr_kernel, dim3((unsigned)nq), dim3(256), 0, stream, (const float*)gm, (int)NG, (int)k, (const float*)qnorm,
                            (const unsigned*)gmax_bits, thr);
    stage = max_stage;   // the cut is already tight: no ramp
    booted = true;
  }
  if (stage > max_stage) stage = max_stage;
  const bool optimistic = schedule == 1 && booted;      // without a bootstrap threshold the first stage passes everything: keep the guaranteed ramp
  if (optimistic) { stage = (N / 8 + BG - 1) / BG * BG; if (stage < BG) stage = BG; }
  // Pipelined stages (k <= 256, more than two stages ahead): the ranking of stage i runs on a second stream while the pre-filter of stage i + 1 scans, so only
  // the scan is on the critical path.  Two list sets alternate; a list's slots [0, k) are reserved for the carried top-k, which the ranking writes into the
  // OTHER set; stage i + 1 filters with the threshold of stage i - 1 (any earlier threshold is a valid lower bound: the results stay bit-identical,
  // a few more rows survive).  Dependencies: P(i) after R(i-2) [threshold, list reuse]; R(i) after P(i) and R(i-1) [same stream].
  const bool pipelined = !approx && !optimistic && k <= 256 && cb_pipeline_enabled() && (N - begin) > 2 * stage;
  hipStream_t s2 = stream;
  if (pipelined) {
    s2 = cb_side_stream();
    if (!s2) return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: side stream");
    hipLaunchKernelGGL(cbir_arm_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, cand.cnt, cand2.cnt, (long)nq, (unsigned)k);
    if (cb_order(0, stream, s2)) return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: event");     // s2 starts after the setup on `stream`
  }
  long si = 0;
  while (begin < N) {
    long end = begin + stage; if (end > N) end = N;
    const long rows = end - begin, tiles = (rows + BG - 1) / BG;
    // one round of workgroups over the 256 CUs: qblocks * nsplit <= 256, every split at least 4 tiles long
    long nsplit_l = 256 / qblocks;
    if (nsplit_l > tiles / 4) nsplit_l = tiles / 4;
    if (nsplit_l < 1) nsplit_l = 1;
    const int nsplit = (int)nsplit_l;
    const long rps = ((tiles + nsplit - 1) / nsplit) * BG;
    const CbirCand& cur = (pipelined && (si & 1)) ? cand2 : cand;
    const CbirCand& oth = (pipelined && (si & 1)) ? cand : cand2;
    if (pipelined && si >= 2 && cb_wait(2 + (si - 2) % 4, stream)) return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: event");   // P(i) after R(i-2)
    cb_launch_prefilter<false>(stream, DP, (unsigned)(qblocks * nsplit), Qb, qnorm, (long)nq, (const bf16_t*)Gb, (const unsigned*)gmax_bits, begin, end, rps, nsplit, (long)idx_base,
                               thr, cur, nullptr, 0L);
    if (pipelined) {
      if (cb_order(1, stream, s2)) return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: event");                                   // R(i) after P(i)
      cb_rank(s2, Q, Gf, (int)D, (long)idx_base, cur, (long)nq, (int)k, thr, out_scores, (long long*)out_idx, (int)(end == N), carry, g_half, &oth, (int)k);
      if (cb_record(2 + si % 4, s2)) return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: event");
    } else if (approx) {
      if (end == N) hipLaunchKernelGGL(cbir_rank_approx_kernel<true>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, Q, Gf, (int)D, (long)idx_base, cand, (long)nq, (int)k,
                                       thr, out_scores, (long long*)out_idx, carry, g_half, (const float*)qnorm, (const unsigned*)gmax_bits);
      else hipLaunchKernelGGL(cbir_rank_approx_kernel<false>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, Q, Gf, (int)D, (long)idx_base, cand, (long)nq, (int)k,
                              thr, out_scores, (long long*)out_idx, carry, g_half, (const float*)qnorm, (const unsigned*)gmax_bits);
    } else {
      cb_rank(stream, Q, Gf, (int)D, (long)idx_base, cand, (long)nq, (int)k, thr, out_scores, (long long*)out_idx, (int)(end == N), carry, g_half);
    }
    begin = end;
    ++si;
    if (optimistic) stage = N;   // the rest in one stage
    else { stage *= 8; if (stage > max_stage) stage = max_stage; }
  }
  if (pipelined && si > 0 && cb_wait(2 + (si - 1) % 4, stream)) return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: event");      // join: results are ordered before what follows on `stream`
  if (N == 0)
    cb_rank(stream, Q, Gf, (int)D, (long)idx_base, cand, (long)nq, (int)k, thr, out_scores, (long long*)out_idx, 1, carry, g_half);      // (empty lists: pads)
  if (overflow_out && hipMemcpyAsync(overflow_out, cand.overflow, 4, hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_cbir_search_fast2: memcpy failed");
  return vdk_check_launch("vdk_cbir_search_fast2");
}

// the original entry point: fp32 gallery, D <= 128, guaranteed schedule
int vdk_cbir_search_fast(const float* Q, int64_t nq, const float* G, const void* Gb, const uint32_t* gmax_bits, int64_t N, int32_t D, int32_t k,
                         int64_t idx_base, float* out_scores, int64_t* out_idx, int64_t cap, void* ws, size_t ws_bytes, void* stream_) {
  if (D > 128) return vdk_fail(VDK_EINVAL, "vdk_cbir_search_fast: D <= 128 (vdk_cbir_search_fast2 takes D <= 512)");
  return vdk_cbir_search_fast2(Q, nq, G, VDK_F32, Gb, gmax_bits, N, D, k, idx_base, out_scores, out_idx, cap, 0, nullptr, ws, ws_bytes, stream_);
}

// Merge S per-shard results (scores [S,nq,k], idx [S,nq,k], -1 = empty) into the global top-k with the
// same (score desc, index asc) rule -> bit-identical to a single-shard search.  (new, C5)
int vdk_cbir_merge_topk(const float* scores, const int64_t* idx, int32_t S, int64_t nq, int32_t k, float* out_scores,
                        int64_t* out_idx, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!scores || !idx || S <= 0 || nq < 0 || k <= 0 || k > 1024) return vdk_fail(VDK_EINVAL, "vdk_cbir_merge_topk: bad argument");
  if (nq == 0) return VDK_OK;
  int64_t cap = (int64_t)S * k; if (cap < 2 * (int64_t)k) cap = 2 * (int64_t)k;
  CbirCand cand; float* thr;
  int rc = cb_carve(ws, ws_bytes, nq, k, cap, &cand, &thr);
  if (rc) return rc;
  hipLaunchKernelGGL(cbir_init_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, cand.cnt, thr,
                     cand.overflow, (long)nq);
  hipLaunchKernelGGL(cbir_fill_from_lists_kernel, dim3((unsigned)nq), dim3(256), 0, stream, cand, scores,
                     (const long long*)idx, (int)S, (long)nq, (int)k);
  cb_rank(stream, nullptr, nullptr, 0, 0, cand, (long)nq, (int)k, thr, out_scores, (long long*)out_idx, 1, nullptr);
  return vdk_check_launch("vdk_cbir_merge_topk");
}

}  // extern "C"

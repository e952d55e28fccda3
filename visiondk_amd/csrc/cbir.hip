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
                                                          int write_out) {
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
    cand.cnt[q] = (unsigned)have;
    thr[q] = (have >= k) ? ord2f(~(unsigned)(keys[k - 1] >> 32)) : __uint_as_float(0xff800000u);  // -inf
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

// ------------------------------------------------------------------------------------ host side
extern "C" {

int vdk_l2norm_rows(const float* x, float* out, int64_t n, int32_t d, float eps, void* stream) {
  if (!x || !out || n < 0 || d <= 0) return vdk_fail(VDK_EINVAL, "vdk_l2norm_rows: bad argument");
  if (n == 0) return VDK_OK;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long)n, (int)d, eps);
  return vdk_check_launch("l2norm_rows_kernel");
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
    hipLaunchKernelGGL(cbir_select_kernel, dim3((unsigned)nq), dim3(256), 0, stream, cand, (long)nq, (int)k, thr,
                       out_scores, (long long*)out_idx, (int)(end == N));
    begin = end;
    stage *= 8;
    if (stage > max_stage) stage = max_stage;
  }
  if (N == 0)
    hipLaunchKernelGGL(cbir_select_kernel, dim3((unsigned)nq), dim3(256), 0, stream, cand, (long)nq, (int)k, thr, out_scores,
                       (long long*)out_idx, 1);
  return vdk_check_launch("vdk_cbir_search");
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
  hipLaunchKernelGGL(cbir_select_kernel, dim3((unsigned)nq), dim3(256), 0, stream, cand, (long)nq, (int)k, thr, out_scores,
                     (long long*)out_idx, 1);
  return vdk_check_launch("vdk_cbir_merge_topk");
}

}  // extern "C"

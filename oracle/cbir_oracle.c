/* cbir_oracle.c — TEST INFRASTRUCTURE: CPU restatement of hot path B.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may call this; the product never does.
 *
 * Restates, for the path  /root/reference engine/cbir/evaluation.py:155-168 (index: faiss
 * IndexFlatIP, METRIC_INNER_PRODUCT, train no-op, add float32 [N,d]) and :171-200 (search: per
 * query batch `faiss_index.search(q.astype(float32), k)` -> (scores float32 [b,k] descending,
 * indices int64 [b,k], -1 padded)), plus F.normalize(p=2, dim=1, eps=1e-12) at
 * models/faceX/face_model.py:139.
 *
 * faiss 1.8.0 (README.md:34) is an un-vendored dependency that is absent from /root/reference and
 * not installable here, so its published algorithm is restated: IndexFlatIP.search = exhaustive
 * inner product S[q][n] = sum_k Q[q][k] * G[n][k] followed by the k largest per query, result
 * sorted by decreasing score, missing results padded with index -1 and score -FLT_MAX
 * (faiss HeapArray<CMin<float,idx_t>>::heapify fills with CMin::neutral() = -FLT_MAX, ids -1).
 * faiss leaves two things unspecified that a bit-exact comparison needs, and this oracle DEFINES them:
 *   (1) summation order: one fmaf chain per pair, k ascending, starting from +0.0f
 *       (acc = fmaf(q[k], g[k], acc)), with -0.0 canonicalised to +0.0 at the end;
 *   (2) tie order: equal scores are ordered by ascending gallery index.
 * PARITY PINNING: the reference has no tests/golden vectors for this path; the
 * oracle is pinned in tests/test_golden.py (test_cbir_oracle_vs_float64_numpy) against numpy's float64 `Q @ G.T` + stable argsort
 * (scores within 2e-6, indices equal wherever the float64 gap exceeds that) and against the committed
 * fixture tests/golden/cbir_small.npz produced by tests/golden/make_golden.py.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PANEL 16

#ifdef _OPENMP
#include <omp.h>
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
void oracle_set_threads(int n) { (void)n; }
#endif

/* out[r] = x[r] / max(||x[r]||_2, eps); norm accumulated in float64 (torch's vectorised float sum
 * is not order-defined; consumers compare with a 1e-6 relative tolerance). */
void oracle_l2norm_rows(const float* x, int64_t n, int32_t d, float eps, float* out) {
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int c = 0; c < d; ++c) s += (double)x[r * d + c] * (double)x[r * d + c];
    float nrm = (float)sqrt(s);
    float inv = 1.0f / (nrm > eps ? nrm : eps);
    for (int c = 0; c < d; ++c) out[r * d + c] = x[r * d + c] * inv;
  }
}

/* exact score of one pair: k-ascending fmaf chain */
float oracle_ip_pair(const float* q, const float* g, int32_t d) {
  float acc = 0.0f;
  for (int k = 0; k < d; ++k) acc = fmaf(q[k], g[k], acc);
  return acc + 0.0f;
}

static inline int better(float s, int64_t i, float s2, int64_t i2) { return s > s2 || (s == s2 && i < i2); }

/* insert (s, idx) into a list sorted by (score desc, idx asc) holding `*cnt` <= k entries */
static inline void insert_sorted(float* sc, int64_t* id, int* cnt, int k, float s, int64_t idx) {
  int n = *cnt;
  if (n == k) {
    if (!better(s, idx, sc[k - 1], id[k - 1])) return;
    n = k - 1;
  }
  int p = n;
  while (p > 0 && better(s, idx, sc[p - 1], id[p - 1])) { sc[p] = sc[p - 1]; id[p] = id[p - 1]; --p; }
  sc[p] = s; id[p] = idx;
  *cnt = n + 1;
}

/* Q [nq,d], G [N,d] row-major float32.  out_scores [nq,k], out_idx [nq,k] (global index = idx_base+row). */
void oracle_flat_ip_search(const float* Q, int64_t nq, const float* G, int64_t N, int32_t d, int32_t k,
                           int64_t idx_base, float* out_scores, int64_t* out_idx) {
  /* transpose the gallery into [N/PANEL][d][PANEL] panels so that PANEL independent fmaf chains
   * vectorise; every chain is still strictly k-ordered. */
  int64_t npan = (N + PANEL - 1) / PANEL;
  float* P = (float*)calloc((size_t)(npan ? npan : 1) * d * PANEL, sizeof(float));
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < npan; ++p)
    for (int j = 0; j < PANEL; ++j) {
      int64_t row = p * PANEL + j;
      if (row >= N) break;
      for (int kk = 0; kk < d; ++kk) P[((size_t)p * d + kk) * PANEL + j] = G[row * d + kk];
    }
  /* blocks of QB queries share each 8 KB panel while it is L1-resident */
  enum { QB = 16 };
  int64_t nqb = (nq + QB - 1) / QB;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t qb = 0; qb < nqb; ++qb) {
    int64_t qlo = qb * QB, qhi = qlo + QB < nq ? qlo + QB : nq;
    int cnt[QB];
    float thr[QB];
    for (int i = 0; i < QB; ++i) { cnt[i] = 0; thr[i] = -INFINITY; }
    for (int64_t p = 0; p < npan; ++p) {
      const float* pp = P + (size_t)p * d * PANEL;
      for (int64_t q = qlo; q < qhi; ++q) {
        const float* qv = Q + q * d;
        float acc[PANEL];
        for (int j = 0; j < PANEL; ++j) acc[j] = 0.0f;
        for (int kk = 0; kk < d; ++kk) {
          float qk = qv[kk];
          for (int j = 0; j < PANEL; ++j) acc[j] = fmaf(qk, pp[kk * PANEL + j], acc[j]);
        }
        float* sc = out_scores + q * k;
        int64_t* id = out_idx + q * k;
        int* c = &cnt[q - qlo];
        for (int j = 0; j < PANEL; ++j) {
          int64_t row = p * PANEL + j;
          if (row >= N) break;
          float s = acc[j] + 0.0f;
          if (*c < k || s > thr[q - qlo]) { /* equal-to-thr with a larger index can never win */
            insert_sorted(sc, id, c, k, s, idx_base + row);
            if (*c == k) thr[q - qlo] = sc[k - 1];
          }
        }
      }
    }
    for (int64_t q = qlo; q < qhi; ++q)
      for (int i = cnt[q - qlo]; i < k; ++i) { out_scores[q * k + i] = -FLT_MAX; out_idx[q * k + i] = -1; }
  }
  free(P);
}

/* k-way merge of S shard results [S][nq][k] with the same total order (multi-GPU C5 reference) */
void oracle_merge_topk(const float* scores, const int64_t* idx, int32_t S, int64_t nq, int32_t k, float* out_scores,
                       int64_t* out_idx) {
  for (int64_t q = 0; q < nq; ++q) {
    float* sc = out_scores + q * k;
    int64_t* id = out_idx + q * k;
    int cnt = 0;
    for (int s = 0; s < S; ++s)
      for (int j = 0; j < k; ++j) {
        int64_t ii = idx[((int64_t)s * nq + q) * k + j];
        if (ii < 0) continue;
        insert_sorted(sc, id, &cnt, k, scores[((int64_t)s * nq + q) * k + j], ii);
      }
    for (int i = cnt; i < k; ++i) { sc[i] = -FLT_MAX; id[i] = -1; }
  }
}

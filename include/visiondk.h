/* visiondk.h — C ABI of libvisiondk_hip.so: the MI355X (gfx950) hot path of wuji3/visiondk.
 *
 * The reference is pure Python; it has no FFI of its own.  Its seams for this path are Python duck
 * types; the functions below are what a ctypes binding behind those seams calls.
 * Each entry cites the reference interface (file:line under /root/reference) whose arithmetic it
 * replaces.  INTEGRATION.md shows the reference-side ctypes stubs.
 *
 * Conventions (every function):
 *   - returns 0 on success or a negative VDK_E* code; vdk_last_error() gives a thread-local message;
 *   - all pointers are DEVICE pointers owned by the caller (torch's caching allocator in practice),
 *     borrowed for the duration of the stream-ordered work; nothing is allocated or freed inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); no entry point synchronises;
 *   - scratch memory is passed explicitly (`ws`, `ws_bytes`); *_workspace_bytes() tells how much;
 *   - matrices are row-major; bf16 = raw uint16 bfloat16 bits; leading dimensions are in ELEMENTS.
 */
#ifndef VISIONDK_H
#define VISIONDK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDK_OK 0
#define VDK_EINVAL (-1)
#define VDK_EWORKSPACE (-2)
#define VDK_ELAUNCH (-3)
#define VDK_EUNSUPPORTED (-4)

/* dtypes */
#define VDK_BF16 0
#define VDK_F32 1
/* GEMM epilogue activations */
#define VDK_ACT_NONE 0
#define VDK_ACT_GELU 1   /* exact-erf GELU (timm Mlp act_layer=nn.GELU); optional pre-activation saved to aux */
#define VDK_ACT_DGELU 2  /* multiply by GELU'(aux)  (backward of the above) */

const char* vdk_last_error(void);
int vdk_is_device_build(void);   /* 1 = compiled by hipcc for gfx950 */
int vdk_abi_version(void);

/* ------------------------------------------------------------------ hot path B: retrieval ----- */

/* F.normalize(x, p=2, dim=1, eps) — models/faceX/face_model.py:139 (extract_cbir), :111 (extract_face).
 * x, out: float32 [n, d]. */
int vdk_l2norm_rows(const float* x, float* out, int64_t n, int32_t d, float eps, void* stream);

/* faiss IndexFlatIP(METRIC_INNER_PRODUCT).search — engine/cbir/evaluation.py:155,168,193 and
 * cbir_eval.py:82,95,116.  Q float32 [nq, D], G float32 [N, D], D % 4 == 0, 1 <= k <= 1024.
 * out_scores float32 [nq, k] descending; out_idx int64 [nq, k] = idx_base + gallery row; equal scores
 * are ordered by ascending index; when N < k the tail is (-FLT_MAX, -1) as faiss pads.  Scores are
 * exact fp32 (k-ordered fmaf chain per pair).  `cap` = per-query candidate capacity (>= 2k) that sizes
 * the workspace: vdk_cbir_workspace_bytes(nq, k, cap). */
int vdk_cbir_workspace_bytes(int64_t nq, int32_t k, int64_t cap, size_t* bytes);
int vdk_cbir_search(const float* Q, int64_t nq, const float* G, int64_t N, int32_t D, int32_t k, int64_t idx_base,
                    float* out_scores, int64_t* out_idx, int64_t cap, void* ws, size_t ws_bytes, void* stream);
/* merge S per-shard results [S, nq, k] (idx < 0 = empty) — the gallery-sharded multi-GPU search
 * (north_star; the reference replicates instead, cbir/evaluation.py:157-162).
 * workspace: vdk_cbir_workspace_bytes(nq, k, max(S*k, 2k)). */
int vdk_cbir_merge_topk(const float* scores, const int64_t* idx, int32_t S, int64_t nq, int32_t k, float* out_scores,
                        int64_t* out_idx, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ hot path A: dense ops ----- */

/* F.linear and its two gradients (timm Attention.qkv/proj, Mlp.fc1/fc2, PatchEmbed.proj, head; built by
 * models/classifier/classify_model.py:49-54, models/faceX/backbone/timm_wrapper.py:16-47):
 *   C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T), bf16 operands, fp32 accumulation on MFMA.
 *   fwd: A = x, B = W[out,in];  dgrad: A = dy, B = W^T[in,out];  wgrad: A = dy^T, B = x^T (split-K).
 * epilogue order: *alpha, +bias[N] (f32), act, +residual[M,N] (f32), store as c_dtype.
 * K, N, lda, ldb, ldc, ldaux % 8 == 0.  splitk > 1 needs ws of vdk_gemm_splitk_workspace_bytes(). */
typedef struct VdkGemmDesc {
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  int32_t M, N, K;
  int32_t c_dtype;         /* VDK_BF16 | VDK_F32 */
  const float* bias;       /* [N] or NULL */
  const float* residual;   /* f32 [M, ldr] or NULL */
  int64_t ldr;
  int32_t act;             /* VDK_ACT_* */
  void* aux;               /* bf16 [M, ldaux]: GELU writes the pre-activation, DGELU reads it; may be NULL for GELU */
  int64_t ldaux;
  float alpha;
  int32_t splitk;          /* <= 1: none */
} VdkGemmDesc;
int vdk_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t splitk, size_t* bytes);
int vdk_gemm_bf16_nt(const VdkGemmDesc* d, void* ws, size_t ws_bytes, void* stream);

/* out[c][r] = in[r][c] (bf16), rows R..Rpad-1 of the new contraction dim zero-filled; feeds wgrad. */
int vdk_transpose_bf16(const void* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, int32_t Rpad,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VISIONDK_H */

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
#define VDK_F16 2   /* IEEE half: gallery storage of the retrieval index (faiss useFloat16) */
/* GEMM epilogue activations */
#define VDK_ACT_NONE 0
#define VDK_ACT_GELU 1   /* exact-erf GELU (timm Mlp act_layer=nn.GELU); optional pre-activation saved to aux */
#define VDK_ACT_DGELU 2  /* multiply by GELU'(aux)  (backward of the above) */
/* the same pair with the derivative evaluated ONCE, in the forward epilogue that already holds erf and exp of the pre-activation: aux (same shape and pitch, 2-byte
 * elements) receives GELU'(pre-activation) as IEEE fp16 instead of the bf16 pre-activation -- the derivative lies in [-0.13, 1.13], where fp16 keeps 11 significand
 * bits -- and the backward epilogue is one multiplication (vdk_gemm_bf16_nt only; aux is required).  NOT what the engines use: the reference's autocast evaluates
 * GELU' from the bf16-ROUNDED pre-activation, and a derivative taken from the unrounded one moves the ViT gradients to 1.54x the oracle's own fp32-vs-fp64
 * accumulation floor (tests/test_parity_bf16.py allows 1.5x) for 0.7 ms of a 38 ms step (measured negatives). */
#define VDK_ACT_GELU_SAVE_GRAD 3
#define VDK_ACT_MUL_AUX 4

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
/* Fast path of the same search for D <= 128 (bit-identical results): a bf16-MFMA pre-filter with the rigorous bound
 * |s' - s| <= ||q~-q|| max||g~|| + ||q|| max||g~-g|| (+ accumulation), all norms measured, keeps every row that could still enter the
 * top-k; only those survivors (about a thousand per query) get the exact fp32 fmaf-chain score before ranking.
 * vdk_cbir_prepare_gallery runs once per index (add time): Gb = bf16 [N, 128] zero-padded copy, gmax_bits = uint32[4] receiving the
 * bits of max_n (||g||, ||g~||, ||g~-g||), gnorm_ws = f32 [3N] scratch. */
int vdk_cbir_prepare_gallery(const float* G, int64_t N, int32_t D, void* Gb, float* gnorm_ws, uint32_t* gmax_bits, void* stream);
int vdk_cbir_fast_workspace_bytes(int64_t nq, int32_t k, int64_t cap, size_t* bytes);
int vdk_cbir_search_fast(const float* Q, int64_t nq, const float* G, const void* Gb, const uint32_t* gmax_bits, int64_t N, int32_t D, int32_t k,
                         int64_t idx_base, float* out_scores, int64_t* out_idx, int64_t cap, void* ws, size_t ws_bytes, void* stream);
/* vdk_cbir_search_fast generalised (csrc/cbir.hip): D <= 512 (face embeddings, timm_wrapper.py:33-47), fp16 gallery STORAGE (g_dtype VDK_F16: G holds half rows; faiss
 * GpuClonerOptions.useFloat16 at engine/cbir/evaluation.py:157-162), and schedule 1 = bootstrap + two stages with candidate lists of `cap` entries whose overflow is
 * REPORTED in *overflow_out (device u32) instead of being impossible by construction -- the caller repeats with schedule 0 if it is set.  Gb / gmax_bits from
 * vdk_cbir_prepare_gallery (Gb = bf16 [N, DP], DP = D rounded up to 128).  Workspace: vdk_cbir_fast2_workspace_bytes.
 * schedule >= 1024: stages of `schedule` rows with lists of `cap` entries (overflow reported).  schedule <= -1024: the same stages, ranked between the stages on the
 * pre-filter's approximate scores (every row within the error band of the k-th best is kept; only the rows kept at the end are re-scored exactly: same results, a
 * fifth of the gallery rows gathered); k <= 256; a band wider than the kernel's slots is reported through *overflow_out as well. */
int vdk_cbir_fast2_workspace_bytes(int64_t nq, int32_t D, int32_t k, int64_t cap, size_t* bytes);
int vdk_cbir_search_fast2(const float* Q, int64_t nq, const void* G, int32_t g_dtype, const void* Gb, const uint32_t* gmax_bits, int64_t N, int32_t D, int32_t k,
                          int64_t idx_base, float* out_scores, int64_t* out_idx, int64_t cap, int32_t schedule, uint32_t* overflow_out, void* ws, size_t ws_bytes,
                          void* stream);

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
 * K, N, lda, ldb, ldc, ldaux % 8 == 0.  splitk > 1 needs ws of vdk_gemm_splitk_workspace_bytes().
 * splitk == -1: stream-K allowed.  ws is then a PERSISTENT workspace of vdk_gemm_streamk_workspace_bytes() whose first 64 KB (tile counters) the caller zeroed
 * once and every launch leaves zero; the launcher uses it when whole-tile rounds would leave the last round mostly empty (one persistent workgroup per CU, each
 * owning a contiguous range of (tile, k-tile) units; partial tiles are combined by the last arriver in K order: bit-reproducible, nobody spins).  One workspace
 * serves one stream at a time. */
/* Implicit-GEMM convolution (im2col-free): when VdkGemmDesc.conv is set, A is an NHWC bf16 tensor [B, H, W, Cin] gathered on the fly.  GEMM row
 * m = (b, oy, ox) over the OH x OW row grid, GEMM column k = (ky*KW + kx)*Cin + c, so B must hold the weight as [N][KH*KW*Cin] in that order
 * (vdk_conv_weight_prep).  transposed = 0: forward conv (source pixel oy*stride + ky - pad); transposed = 1: the input gradient of that conv
 * (rows = input pixels, A = dY [B, H, W, Cin=Cout], source pixel (oy + pad - ky)/stride when it divides).  Cin % 8 == 0; K == KH*KW*Cin. */
typedef struct VdkConvGeom {
  int32_t Cin, H, W, OH, OW, KH, KW, stride, pad, transposed;
  int32_t rows;   /* VdkGemmDesc.trans = 1 only -- the WEIGHT GRADIENT of the convolution without an im2col matrix: A = dY [rows, Cout] (row m = (b, oy, ox), lda = Cout), B = the
                   * NHWC input [batch, H, W, Cin] gathered on the fly as the k-major im2col operand [rows, KH*KW*Cin] (ldb ignored), C = dW' f32 [Cout, KH*KW*Cin] (column
                   * order of vdk_conv_weight_prep; vdk_conv_wgrad_unpermute brings it back to [Cout][Cin][KH][KW]).  rows = batch * OH * OW (< 2^24) is the number of VALID
                   * contraction rows; VdkGemmDesc.K is that number rounded up to a multiple of 128 (rows beyond `rows` read as zeros).  transposed = 0, N = KH*KW*Cin. */
} VdkConvGeom;
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
  int32_t row_group;       /* > 0: output row m -> m + m/row_group + 1, residual row -> m % row_group + 1
                              (PatchEmbed rows written straight into the [B, 1+np, D] token buffer + pos_embed);
                              < 0: output rows stay, residual row -> m % |row_group| (the same without a class token: [B, np, D] + pos_embed) */
  int32_t trans;           /* 0: C = A[M,K] . B[N,K]^T.  1: TN, A is [K, M] and B is [K, N] row-major, C = A^T . B (wgrad straight from
                              dY[t][out], X[t][in]); needs K and the split size % 64 == 0, M % 8 == 0, lda/ldb % 8 == 0 */
  int32_t a_row_group;     /* trans=1 only, > 0: A's k-row t lives at physical row t + t/a_row_group + 1 (token buffer minus cls rows) */
  const VdkConvGeom* conv; /* NULL: dense A.  else: implicit-GEMM convolution operand (lda ignored) */
  float* a_colsum;         /* NULL, or f32 [vdk_gemm_a_colsum_rows(M,N,K)][K]: by-product of the NT 256x256 kernel, partial column sums of A, one row per 256-row tile
                              (sum the rows -> colsum(A) = bias gradient of the Linear whose dY is this dgrad GEMM's A); error if that kernel does not serve the problem */
  float* c_colsum;         /* optional by-product of the 256x256 NT kernel with a plain or dGELU bf16 epilogue: f32 [vdk_gemm_c_colsum_rows(M, N, K)][N] partial column sums of the
                              STORED (bf16-rounded) output (sum the rows -> colsum(C) = bias gradient of the Linear whose dY this GEMM's output is) */
  int32_t ab_dtype;        /* format of A, B, aux and a 16-bit C: VDK_BF16 (0, the default of a zeroed descriptor) or VDK_F16 -- IEEE half operands on v_mfma_f32_32x32x16_f16,
                              what the reference's `torch.autocast(device_type=...)` gives its matmuls on a GPU (engine/procedure/train.py:118: no dtype => float16).  With
                              VDK_F16 a 16-bit output is requested as c_dtype = VDK_F16.  fp16 operands exclude conv, a_colsum and stream-K. */
  const float* col_scale;  /* NULL, or f32 [N]: the accumulator of column n is multiplied by col_scale[n] BEFORE bias / residual -- C = residual + (acc * col_scale + bias).
                              ConvNeXt's layer scale (x + gamma * fc2(g), timm ConvNeXtBlock behind models/faceX/backbone/timm_wrapper.py:16-21) in the fc2 epilogue when the
                              operands are fp16: gamma (1e-6 at timm's init) folded into an fp16 weight would underflow.  fp32 outputs with a bias only (act NONE, no split-K). */
  const float* row_scale;  /* NULL, or f32 [ceil(M / rows_per_scale)]: C = residual + row_scale[m / rows_per_scale] * (acc + bias) -- stochastic depth (timm DropPath: the branch of
                              sample b is multiplied by mask_b / keep_prob before the shortcut is added; timm's Swin builds with drop_path_rate = 0.1 behind
                              models/classifier/classify_model.py:49-54) in the proj / fc2 epilogues, one factor per sample.  Same restrictions as col_scale, and a residual. */
  int32_t rows_per_scale;  /* rows of C that share one row_scale entry (tokens per image) */
} VdkGemmDesc;
int vdk_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t splitk, size_t* bytes);
int vdk_gemm_streamk_workspace_bytes(size_t* bytes);
int vdk_gemm_bf16_nt(const VdkGemmDesc* d, void* ws, size_t ws_bytes, void* stream);
/* OCP fp8 operands on the block-scaled MFMA (csrc/gemm_fp8.hip): the Linear GEMMs of BASELINE.json configs[4] ("SigLIP ViT-L/14 336 ... fp8 MFMA"; the reference has
 * no fp8 code -- this is timm's Linear, models/classifier/classify_model.py:49-54, under per-tensor delayed scaling).  fmt 0 = e4m3 (activations, weights), 1 = e5m2 (gradients).
 *   vdk_quant_fp8         x (bf16 | f32, n % 16 == 0) -> fp8(clamp(x * scale[0])); amax[0] = max(amax[0], max |x|) (device scalars; scale NULL = 1, amax NULL = skip)
 *   vdk_fp8_scale_update  n tensors: scale = fmt_max / (margin * amax), scale_inv = 1 / scale (kept while amax == 0), amax := 0
 *   vdk_gemm_fp8_nt       C = epilogue(a_scale_inv[0] * b_scale_inv[0] * A . B^T), A [M, K] (a_fmt), B [N, K] (e4m3), fp32 accumulation; epilogue fields of the
 *                         descriptor as for vdk_gemm_bf16_nt; needs M, N >= 256, K % 128 == 0, lda / ldb % 16 == 0 (elements = bytes); no split-K / TN / conv. */
int vdk_quant_fp8(const void* x, int32_t x_dtype, int64_t n, const float* scale, void* out_fp8, int32_t fmt, float* amax, void* stream);
int vdk_fp8_scale_update(float* amax, float* scale, float* scale_inv, int32_t n, int32_t fmt, float margin, void* stream);
int vdk_gemm_fp8_nt(const VdkGemmDesc* d, int32_t a_fmt, int32_t b_fmt, const float* a_scale_inv, const float* b_scale_inv, void* stream);
/* the same GEMM, and the epilogue that stores the bf16 C also writes what vdk_quant_fp8(C, out_scale, out_fmt) would (bit-identical) and accumulates max |C| into out_amax:
 * the output of fc1 + GELU / of the dGELU input-gradient GEMM is the A operand of the next fp8 GEMM (timm Mlp, models/classifier/classify_model.py:49-54), so the separate
 * quantisation pass over it (2 B read + 1 B written per element) disappears.  GELU (+ bias + aux) and DGELU epilogues only, bf16 C, N % 64 == 0, ldo8 % 8 == 0. */
int vdk_gemm_fp8_nt_q8(const VdkGemmDesc* d, int32_t a_fmt, int32_t b_fmt, const float* a_scale_inv, const float* b_scale_inv, void* out8, int64_t ldo8, int32_t out_fmt,
                       const float* out_scale, float* out_amax, void* stream);
int vdk_gemm_c_colsum_rows(int32_t M, int32_t N, int32_t K);   /* rows of VdkGemmDesc.c_colsum, 0 = by-product not available for this problem */
int vdk_gemm_a_colsum_rows(int32_t M, int32_t N, int32_t K);   /* rows of VdkGemmDesc.a_colsum, 0 = by-product not available for this problem */
/* leave n CUs (0..128, rounded so that the walk stays a multiple of 8) out of the persistent GEMM grids of this process: a data-parallel host sets it to the number of
 * channels its collectives run on (visiondk_amd/comm.py: 32), so that an all-reduce in flight on another stream and a persistent GEMM fit on the chip together instead of
 * the GEMM's static tile walk waiting for the collective to end.  Results do not depend on it.  Environment VDK_GEMM_RESERVE_CUS overrides. */
int vdk_gemm_reserve_cus(int32_t n);
int vdk_gemm_reserved_cus(void);   /* the value in force (so that a caller can scope a reserve to a window and restore what it found) */

/* live GEMM timing for bench.py's `roofline` (HIP events on the launch stream around every GEMM kernel):
 * begin(max_launches) pre-creates the events; end() synchronises and returns the totals since begin(). */
int vdk_prof_begin(int32_t max_launches);
int vdk_prof_pause(int32_t paused);   /* suspend (1) / resume (0) recording between vdk_prof_begin and vdk_prof_end */
int vdk_prof_end(double* total_ms, int64_t* launches, double* total_flops);
int vdk_prof_bytes(double* total_bytes);   /* algorithmic bytes of those launches (every operand / output / epilogue tensor counted once) */

/* out[c][r] = in[r][c] (bf16), rows R..Rpad-1 of the new contraction dim zero-filled; feeds wgrad.
 * in_row_group > 0: logical row r lives at physical row r + r/in_row_group + 1 (token buffer without cls rows).
 * colsum_partial (optional): f32 [ceil(Rpad/64)][C] per-row-tile column sums (the Linear bias gradient rides along). */
int vdk_transpose_bf16(const void* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, int32_t Rpad,
                       int32_t in_row_group, float* colsum_partial, void* stream);


/* N <= 256 keys: forward with the whole (batch, head) item in LDS and an exact (non-online) softmax, N <= 224: recompute-form backward in two kernels
 * (csrc/attention_small.hip); longer sequences: forward and recompute-form backward with the other operand streamed through a double-buffered LDS chunk
 * (csrc/attention_long.hip).  on = 1 forces the round-1 flash-style kernels of csrc/attention.hip at every N (A/B timing, tests), 0 the default
 * routing, -1 hands the choice back to the VDK_ATTN_LEGACY environment variable. */

/* timm Attention core: softmax(q k^T * scale) v per head, flash-style (the N x N matrix is never
 * written).  qkv: bf16 [B, N, 3, H, 64] = the fused qkv Linear output (row stride ld elements);
 * o: bf16 [B, N, H*64] (row stride ldo); lse: f32 [B, H, N] log-sum-exp saved for backward (NULL ok).
 * head_dim must be 64 (ViT-B/16: 12 heads, ViT-L: 16 heads). */
int vdk_attention_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H,
                      int32_t head_dim, float scale, void* stream);
/* backward: dqkv bf16 [B, N, 3, H, 64] (row stride lddqkv); dvec: f32 scratch [B, H, N]. */
int vdk_attention_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv,
                      int64_t lddqkv, float* dvec, int32_t B, int32_t N, int32_t H, int32_t head_dim, float scale, void* stream);
/* vdk_attention_fwd / vdk_attention_bwd with the 16-bit format of q, k, v, o, dO, dqkv as a parameter: dtype = VDK_BF16 | VDK_F16.  VDK_F16 is the arithmetic of the
 * reference's GPU path: `torch.autocast(device_type=...)` (engine/procedure/train.py:118) without a dtype is float16, so timm's Attention (softmax(q k^T / sqrt(hd)) v
 * behind models/classifier/classify_model.py:49-54) reads fp16 operands there; P and dS are rounded to fp16 where the bf16 kernels round to bf16.  The short- and the
 * long-sequence kernels serve both formats; the flash-style kernels of round 1 (vdk_attention_force_legacy) are bf16 only -> VDK_EUNSUPPORTED. */
int vdk_attention_fwd_dt(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H, int32_t head_dim, float scale, int32_t dtype,
                         void* stream);
int vdk_attention_bwd_dt(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv, int64_t lddqkv, float* dvec, int32_t B,
                         int32_t N, int32_t H, int32_t head_dim, float scale, int32_t dtype, void* stream);

/* F.layer_norm over the last dim (timm blocks' norm1/norm2 and the final norm, eps 1e-6).  x: f32 rows at
 * stride ldx (so the final norm can run on the cls rows only: ldx = N*C); y: bf16 or f32; mean/rstd: f32 [T]
 * saved for backward (NULL ok).  C % 4 == 0. */
int vdk_layernorm_fwd(const float* x, int64_t ldx, int32_t T, int32_t C, const float* gamma, const float* beta, float eps,
                      void* y, int64_t ldy, int32_t y_dtype, float* mean, float* rstd, void* stream);
/* vdk_layernorm_fwd with bf16 y, and the fp8 quantisation of y written by the same kernel (out8 [T, ldo8] bytes = what vdk_quant_fp8(y, out_scale, out_fmt) gives, amax
 * accumulated into out_amax): the engine's fp8 mode feeds norm1 / norm2 outputs to fp8 GEMMs.  128 < C <= 1024, ldo8 % 4 == 0. */
int vdk_layernorm_fwd_q8(const float* x, int64_t ldx, int32_t T, int32_t C, const float* gamma, const float* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd,
                         void* out8, int64_t ldo8, int32_t out_fmt, const float* out_scale, float* out_amax, void* stream);
/* backward: dx = LN'(dy) [+ dres] as f32 (dx) and/or bf16 (dxb); dgamma, dbeta f32 [C] overwritten. */
int vdk_layernorm_bwd_workspace_bytes(int32_t T, int32_t C, size_t* bytes);
int vdk_layernorm_bwd(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* mean,
                      const float* rstd, const float* gamma, const float* dres, int64_t lddres, int32_t T, int32_t C, float* dx,
                      int64_t lddx, void* dxb, int64_t lddxb, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                      void* stream);

/* nn.BatchNorm1d on [B, F] — last layer of the TimmWrapper neck (models/faceX/backbone/timm_wrapper.py:37,46).  training != 0:
 * batch statistics (saved for backward) + running-stat update (unbiased var, momentum); else running statistics. */
int vdk_batchnorm1d_fwd(const float* x, int64_t ldx, int32_t B, int32_t F, const float* gamma, const float* beta, float eps, float momentum,
                        int32_t training, float* running_mean, float* running_var, float* y, int64_t ldy, float* save_mean, float* save_invstd,
                        void* stream);
int vdk_batchnorm1d_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int32_t B, int32_t F, const float* gamma, const float* save_mean,
                        const float* save_invstd, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* stream);

/* out[c] = scale * sum_{s<S} in[s*ld + c]  (deterministic; pos_embed/cls gradients, partial combines) */
int vdk_reduce_rows_f32(const float* in, int64_t ld, int32_t S, int64_t n, float* out, float scale, void* stream);
/* tests: the same reduction through the in-library batch path the engines use (up to 8 reductions per launch); `buf` is SCRATCH: from 1024 partial rows on they are folded onto
 * the first 64 in place before the final sum (fixed order: bit-reproducible) */
/* out[c] = sum_r in[r][c], in bf16 [T, N] — bias gradient of a Linear */
int vdk_colsum_bf16_workspace_bytes(int32_t T, int32_t N, size_t* bytes);
int vdk_colsum_bf16(const void* in, int64_t ld, int32_t T, int32_t N, float* out, void* ws, size_t ws_bytes, void* stream);

/* nn.CrossEntropyLoss(label_smoothing) fwd+bwd in one pass — models/losses/loss.py:71-73; with yb != NULL it is
 * mixup_criterion (engine/procedure/train.py:34-35): lam*CE(ya) + (1-lam)*CE(yb).
 * loss_rows f32 [B] (caller averages); dlogits = grad_scale * dLoss_row/dlogits as bf16 [B, lddl] (columns
 * C..lddl-1 zeroed: they pad the contraction dim of the head GEMMs) and/or f32 [B, lddf]. */
int vdk_softmax_ce(const float* logits, int64_t ldl, int32_t B, int32_t C, const int64_t* ya, const int64_t* yb, float lam,
                   float label_smoothing, float grad_scale, float* loss_rows, void* dlogits_bf16, int64_t lddl,
                   float* dlogits_f32, int64_t lddf, void* stream);
/* nn.BCEWithLogitsLoss — models/losses/loss.py:68-70; focal_gamma > 0: FocalLoss(BCEWithLogits, gamma, alpha), loss.py:27-54,74-76.
 * loss_rows[b] = sum_c loss(b,c) (caller divides by B*C). */
int vdk_bce_logits(const float* logits, int64_t ldl, const float* targets, int64_t ldt, int32_t B, int32_t C, float grad_scale,
                   float focal_gamma, float focal_alpha, float* loss_rows, void* dlogits_bf16, int64_t lddl, float* dlogits_f32, int64_t lddf,
                   void* stream);
/* vdk_softmax_ce / vdk_bce_logits under GradScaler (engine/procedure/train.py:205 `scaler.scale(loss).backward()`): the gradient written to dlogits16 (dl_dtype =
 * VDK_BF16 | VDK_F16) and dlogits_f32 is multiplied by grad_scale AND by the device scalar loss_scale[0] (NULL: 1) -- the loss scale lives on the device so that a skipped
 * step or a grown scale costs no host round trip.  loss_rows stay unscaled. */
int vdk_softmax_ce_amp(const float* logits, int64_t ldl, int32_t B, int32_t C, const int64_t* ya, const int64_t* yb, float lam, float label_smoothing, float grad_scale,
                       const float* loss_scale, float* loss_rows, void* dlogits16, int64_t lddl, int32_t dl_dtype, float* dlogits_f32, int64_t lddf, void* stream);
int vdk_bce_logits_amp(const float* logits, int64_t ldl, const float* targets, int64_t ldt, int32_t B, int32_t C, float grad_scale, const float* loss_scale,
                       float focal_gamma, float focal_alpha, float* loss_rows, void* dlogits16, int64_t lddl, int32_t dl_dtype, float* dlogits_f32, int64_t lddf,
                       void* stream);

/* PatchEmbed.proj (Conv2d(3, D, p, stride p)) as an im2col-free GEMM operand: out bf16 [B*gh*gw, Kp],
 * column k = c*p*p + ky*p + kx, zero-padded to Kp (Kp % 8 == 0). */
int vdk_patchify_bf16(const float* x, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* out, int32_t Kp,
                      void* stream);
/* tok[b, 0, :] = cls_token + pos_embed[0]  (timm _pos_embed: cat cls, then add pos) */
int vdk_cls_rows(float* tok, int64_t batch_stride, int32_t B, int32_t D, const float* cls, const float* pos0, void* stream);
int vdk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream);
int vdk_cast_f32_f16(const float* in, void* out, int64_t n, void* stream);   /* the same cast to IEEE half (fp16 operand mode, VdkVitConfig.operand) */
/* out[c][r] (bf16) = in[r][c] (f32): the [in,out] copy of a Linear weight for dgrad */
int vdk_transpose_cast_f32_bf16(const float* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, int32_t Rpad,
                                void* stream);

/* ---- fused multi-tensor step over FLAT buffers -------------------------------------------------
 * Trainer.update (engine/procedure/train.py:203-215): [scaler.unscale_] -> clip_grad_norm_(max_norm=10) ->
 * optimizer.step (torch.optim.SGD momentum/weight_decay, engine/optimizer.py:119-121) -> ema.update
 * (models/ema.py:28-37), plus the bf16 refresh of the weights the GEMMs read. */
int vdk_sumsq_workspace_bytes(size_t* bytes);
int vdk_sumsq_f32(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream);
int vdk_sgd_step(float* params, const float* grads, float* momentum_buf, float* ema, void* params_bf16, int64_t n, float lr,
                 float momentum, float weight_decay, float grad_scale, const float* normsq, float max_norm, float ema_decay,
                 int32_t first_step, void* stream);
/* The same pass for a train step captured in a hipGraph (torch.cuda.CUDAGraph): by-value kernel arguments freeze at capture, but the scheduler changes lr
 * (engine/scheduler.py) and ModelEMA's decay warms up every update (models/ema.py:24), so lr, momentum, weight_decay, ema_decay and first_step (!= 0)
 * are read from DEVICE memory `hyper` f32 [5] when the kernel runs. */
int vdk_sgd_step_graph(float* params, const float* grads, float* momentum_buf, float* ema, void* params_bf16, int64_t n, const float* hyper, float grad_scale,
                       const float* normsq, float max_norm, void* stream);
/* Trainer.update under GradScaler (engine/procedure/train.py:203-215; GradScaler built at engine/vision_engine.py:232): `scaler.unscale_` + `clip_grad_norm_` +
 * `scaler.step(optimizer)` + ModelEMA.update in the one pass of vdk_sgd_step.  The gradients (and normsq = their sum of squares, vdk_sumsq_f32) carry the loss scale
 * loss_state[0] (device; NULL: none): g / scale is what is clipped and applied; when normsq is not finite (an inf / NaN anywhere in the gradient) parameters and momentum are
 * left untouched -- GradScaler.step skips optimizer.step() -- while the EMA still moves (the reference calls ema.update(model) regardless).  params16 (may be NULL) is
 * refreshed in p16_dtype = VDK_BF16 | VDK_F16: the engine's operand copy of the weights.  Call vdk_loss_scale_update afterwards (`scaler.update()`, train.py:211):
 * loss_state f32 [3] = {scale, growth tracker, skipped-step count}; non-finite normsq: scale *= backoff_factor, tracker = 0; otherwise tracker += 1 and, at
 * growth_interval, scale *= growth_factor, tracker = 0 (torch.cuda.amp.GradScaler defaults: init 65536, growth 2, backoff 0.5, interval 2000). */
int vdk_sgd_step_amp(float* params, const float* grads, float* momentum_buf, float* ema, void* params16, int32_t p16_dtype, int64_t n, float lr, float momentum,
                     float weight_decay, float grad_scale, const float* loss_state, const float* normsq, float max_norm, float ema_decay, int32_t first_step, void* stream);
int vdk_loss_scale_update(float* loss_state, const float* normsq, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream);
/* SAM.first_step (engine/optimizer.py:43-55,77-87): normsq_out = sum((|p| or 1) * g)^2; old_params = params;
 * params += (p^2 or 1) * g * rho / (sqrt(normsq) + 1e-12).  second_step = copy old_params back + vdk_sgd_step on the new grads.
 * ws: vdk_sumsq_workspace_bytes(). */
int vdk_sam_first_step(float* params, const float* grads, float* old_params, int64_t n, float rho, int32_t adaptive, float* normsq_out, void* ws,
                       size_t ws_bytes, void* stream);
/* OHEMImageSampler.sample (structure/sampler.py:11-31): mask u8 [B]; prob_ws f32 [B] scratch */
int vdk_ohem_mask(const float* logits, int64_t ldl, int32_t B, int32_t C, const int64_t* labels, int32_t min_kept, float thresh, int64_t ignore_index,
                  float* prob_ws, uint8_t* mask, void* stream);
/* top-k per logits row, descending, ties -> lower index (engine/procedure/evaluation.py:106 `argsort(1, descending=True)[:, :k]`) */
int vdk_topk_rows(const float* x, int64_t ld, int32_t B, int32_t C, int32_t k, int64_t* idx, float* values, void* stream);
/* mixup_data (engine/procedure/train.py:24-32): out[b] = lam*x[b] + (1-lam)*x[perm[b]] */
int vdk_mixup(const float* x, const int64_t* perm, float lam, int32_t B, int64_t per_sample, float* out, void* stream);


/* ---- PRECISE inference path: fp32-in / fp32-out contractions on the fp32 MFMA (v_mfma_f32_32x32x2_f32) -------------------------
 * For evaluation / embedding extraction within 1e-3 (measured ~1e-6) of the reference's PyTorch-CPU fp32 path (north_star's tolerance); the
 * training path keeps bf16 MFMA operands.  C[z] = epilogue(alpha * A[z] B[z]^T): A [M, K] rows, B [N, K] rows or (b_kmajor) [K, N];
 * K, lda, ldb % 4 == 0; z = (z1, z2) with z1 < batch1, z2 < batch2 and element strides s?1 / s?2 (0 batches = 1). */
typedef struct VdkGemmF32Desc {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int32_t M, N, K;
  const float* bias;          /* [N] or NULL */
  const float* residual;      /* f32 [M, ldr] or NULL, added after the activation */
  int64_t ldr;
  int32_t act;                /* VDK_ACT_NONE | VDK_ACT_GELU (exact erf) */
  float alpha;                /* 0 means 1 */
  const float* col_scale;     /* [N] or NULL: multiplies the activated value before the residual (ConvNeXt layer scale) */
  int32_t b_kmajor;
  int32_t batch1, batch2;
  int64_t sa1, sa2, sb1, sb2, sc1, sc2;
  int32_t a_kmajor;           /* 1: A is [K, M] row-major (M % 4 == 0): C = A^T B -- with b_kmajor the weight-gradient form dW[out, in] = dY[t, out]^T X[t, in] */
  int64_t k_total;            /* > 0: the contraction has k_total rows and is SPLIT over batch1: batch z multiplies rows [z K, min((z + 1) K, k_total)) into its own C slab
                                 (sa1 = K lda, sb1 = K ldb for k-major operands, sc1 = slab pitch); the caller sums the slabs (vdk_reduce_rows_f32) */
} VdkGemmF32Desc;
int vdk_gemm_f32_nt(const VdkGemmF32Desc* d, void* stream);
/* ---- Swin Transformer (timm swin_*_patch4_window7_224: the default backbone of both shipped configs, configs/classification/pet.yaml:25, configs/faceX/cbir.yaml:26) ----
 * The attention step of timm's WindowAttention per (window, head): S = (q hd^-0.5) k^T + bias[head] (+ mask[window mod nW]); P = softmax(S); o = P v, with 49 tokens (7 x 7)
 * and head dim 32.  qkv bf16 [windows * 49, ld]: q | k | v thirds of 3 * H * 32 columns, rows in (window, token) order; bias f32 [H, 49, 49] (the relative-position table
 * gathered by the caller); mask f32 [nW, 49, 49] (0 / -100 of the shifted windows) or NULL; lse f32 [windows, H, 49] saved for the backward.  backward: dqkv bf16 (dq | dk | dv),
 * dbias f32 [H, 49, 49] summed over every window in a fixed order (no atomics).  Both calls take a workspace (the bias + mask re-laid in the MFMA register order; the
 * backward also the per-wave d(bias) partials).  rowidx int32 [windows * 49] or NULL: token j of window w is tensor row rowidx[w * 49 + j] of qkv / o / dout / dqkv (timm's
 * roll + window_partition / window_reverse + roll around the attention in SwinTransformerBlock, as an index: no gather copies). */
int vdk_window_attention_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, const float* bias, const float* mask, int32_t nW, int64_t windows, int32_t H, int32_t N,
                             int32_t hd, float scale, const int32_t* rowidx, void* ws, size_t ws_bytes, void* stream);
int vdk_window_attention_fwd_workspace_bytes(int32_t nW, int32_t H, size_t* bytes);   /* nW = 0 without a mask */
int vdk_window_attention_bwd_workspace_bytes(int64_t windows, int32_t nW, int32_t H, size_t* bytes);
int vdk_window_attention_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, const float* bias, const float* mask, int32_t nW,
                             int64_t windows, int32_t H, int32_t N, int32_t hd, float scale, const int32_t* rowidx, void* dqkv, int64_t ldd, float* dbias, void* ws,
                             size_t ws_bytes, void* stream);
/* backward of timm's bias gather relative_position_bias_table[relative_position_index] -> [H, N, N]: dtable f32 [R, H] = sum over the positions (uses int32 [R, U], -1
 * padded, list order) of dbias f32 [H, NN]: a deterministic gather per table entry instead of an index_add */
int vdk_relpos_bias_table_grad(const float* dbias, const int32_t* uses, int32_t R, int32_t U, int32_t H, int32_t NN, float* dtable, void* stream);
/* elementwise / reduction pieces of the fp32-class TRAINING path of the face / CBIR task (the reference runs that loop without autocast, engine/procedure/train.py:217-227):
 * exact-erf GELU and its derivative with the library erff / expf, a row scale (ConvNeXt layer scale folded into fc2), deterministic column sums of an f32 tensor
 * (bias gradients) and the inverse of vdk_space_to_depth2_f32 */
int vdk_gelu_f32(const float* u, float* g, int64_t n, void* stream);
int vdk_dgelu_f32(float* d_inplace, const float* u, int64_t n, void* stream);
int vdk_rowscale_f32(const float* in, const float* scale, float* out, int64_t rows, int64_t cols, void* stream);
int vdk_colsum_f32_workspace_bytes(int64_t T, int32_t N, size_t* bytes);
int vdk_colsum_f32(const float* x, int64_t ld, int64_t T, int32_t N, float* out, void* ws, size_t ws_bytes, void* stream);
int vdk_depth_to_space2_f32(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* in-place softmax(scale * x) over the first `cols` columns of every row; columns [cols, ld) are zeroed */
int vdk_softmax_rows_f32(float* x, int64_t ld, int64_t rows, int32_t cols, float scale, void* stream);
/* fp32 operands of the k = stride convolutions for the precise path: PatchEmbed / ConvNeXt stem (NCHW input, k = c*p*p + ky*p + kx) and the
 * 2x2 stride-2 downsample on NHWC rows (k = c*4 + 2*ky + kx, the weight's own flattening) */
int vdk_patchify_f32(const float* x, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, float* out, void* stream);
int vdk_space_to_depth2_f32(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* ---- CNN backbone pieces (timm ConvNeXt behind TimmWrapper, models/faceX/backbone/timm_wrapper.py:16-37) --------------------
 * All activations are NHWC f32 / bf16 rows, so pointwise / 4x4-stem / 2x2-downsample convolutions are vdk_gemm_bf16_nt and
 * LayerNorm2d is vdk_layernorm_*.  ConvNeXtBlock.conv_dw (Conv2d(C, C, 7, padding 3, groups=C)):
 * out = bias + res + dwconv(in, wt) with wt tap-major f32 [49][C] (vdk_dwconv7_weight_prep); flip = 1 turns the same kernel into
 * the input gradient (taps reversed; res = shortcut gradient; out_bf16 = copy for the next GEMM).  bias / res / out / out_bf16 may be NULL. */
int vdk_dwconv7_fwd(const float* in, const float* wt, const float* bias, const float* res, float* out, void* out_bf16, int32_t B, int32_t H, int32_t W,
                    int32_t C, int32_t flip, void* stream);
int vdk_dwconv7_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, size_t* bytes);
/* dw f32 [C][49] (= conv_dw.weight [C,1,7,7]) and db f32 [C] from the block input `in` and the gradient `dy` of the conv output */
int vdk_dwconv7_wgrad(const float* in, const float* dy, float* dw, float* db, int32_t B, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes,
                      void* stream);
int vdk_dwconv7_weight_prep(const float* w, float* wt, int32_t C, void* stream);
/* stride-2 2x2 convolution (ConvNeXt downsample) as a GEMM: space-to-depth operand bf16 [B*H/2*W/2][4C] with k = (2*(y&1) + (x&1))*C + c
 * (inverse = 1: depth-to-space, for the input gradient); weight [Co][Ci][2][2] -> bf16 [Co][4Ci] in the same k order + its transpose;
 * gradient back to the timm layout. */
int vdk_space_to_depth2_bf16(const void* in, void* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t inverse, void* stream);
int vdk_conv2x2_weight_prep(const float* w, void* wb, void* wtb, int32_t Co, int32_t Ci, void* stream);
int vdk_conv2x2_wgrad_unpermute(const float* dwp, float* dw, int32_t Co, int32_t Ci, void* stream);
/* ConvNeXtBlock layer scale (x * gamma after mlp.fc2) folded into fc2: W2p = gamma (.) W2 (bf16 [C][M]), W2pt = W2p^T, b2p = gamma (.) b2;
 * vdk_layerscale_grad applies the chain rule to the folded gradients: dW2 = gamma (.) dW2p, db2 = gamma (.) db2p,
 * dgamma[c] = sum_k dW2p[c][k] W2[c][k] + db2p[c] b2[c]. */
int vdk_layerscale_weight_prep(const float* w2, const float* b2, const float* gamma, void* w2p, void* w2pt, float* b2p, int32_t C, int32_t M, void* stream);
int vdk_layerscale_grad(const float* dw2p, const float* db2p, const float* w2, const float* b2, const float* gamma, float* dw2, float* db2, float* dgamma,
                        int32_t C, int32_t M, void* stream);

/* ---- native ViT engine: timm VisionTransformer forward/backward over flat buffers ---------------------
 * Replaces `self.model(images)` + `loss.backward()` of Trainer.compute_loss / Trainer.update
 * (engine/procedure/train.py:177-215) for the model `timm.create_model('vit_*', num_classes=C)` built at
 * models/classifier/classify_model.py:49-54.  Semantics restated from timm 0.9.16 (requirements.txt:10; not
 * vendored): patch_embed conv -> cat cls_token -> + pos_embed -> depth x pre-norm blocks (LayerNorm eps,
 * fused qkv, softmax attention, proj; LayerNorm, fc1, exact GELU, fc2) -> final LayerNorm -> token 0 -> head. */
typedef struct VdkVitConfig {
  int32_t batch, img_size, patch_size, in_chans;
  int32_t dim, depth, heads, mlp_dim, num_classes;
  float ln_eps;
  int32_t no_class_token;  /* 1: timm class_token=False (SigLIP): N = num_patches tokens, no cls_token parameter; feature mode only (num_classes == 0: pooling + head on top,
                              visiondk_amd/vit.py AttentionPoolLatent) */
  int32_t fp8;             /* BASELINE.json configs[4] "fp8 MFMA": 0 = bf16 operands; 1 = the four Linears of every block run their forward and input-gradient GEMMs on OCP fp8
                              operands (e4m3 activations / weights, e5m2 gradients, per-tensor DELAYED scaling: this step's scale comes from the last step's amax);
                              2 = the same with CURRENT scaling (an amax pass in front of every quantisation: calibration steps).  Weight gradients stay bf16 TN GEMMs. */
  void* fp8_w;             /* fp8 != 0: e4m3 operand copies, n_floats bytes in the parameter layout followed by n_transposed bytes in the wt16 layout (vdk_vit_refresh_weights) */
  float* fp8_state;        /* fp8 != 0: f32 [3][12 * depth]: amax | scale | 1 / scale; slot 12 l + k, k = 0..3 h1, attn out, h2, gelu out (e4m3), 4..7 qkv / proj / fc1 / fc2
                              weights (e4m3), 8..11 dL/d(fc2 out), dL/du, dL/d(proj out), dL/dqkv (e5m2).  Initialise scale = 1/scale = 1, amax = 0; call vdk_vit_fp8_update
                              after every backward. */
  int32_t operand;         /* VDK_BF16 (0): bf16 GEMM operands / saved activations (BASELINE.json configs[1] "bf16").  VDK_F16: IEEE fp16 -- the reference's own GPU arithmetic
                              (engine/procedure/train.py:118 `torch.autocast(device_type=...)` without a dtype = float16, with GradScaler train.py:205-211): wb16 / wt16, every
                              saved activation and every gradient tensor of the engine are fp16; the caller scales dlogits by the loss scale (vdk_softmax_ce_amp) and the
                              optimizer un-scales (vdk_sgd_step_amp).  Same speed (same MFMA rate), 8x smaller operand rounding.  Excludes fp8. */
  int32_t pre_norm;        /* 1: timm pre_norm=True (the CLIP ViTs, `vit_*_clip_*`: models/classifier/classify_model.py:49-54 builds them by id): a LayerNorm `norm_pre` between the
                              embedding (patches + pos_embed, class token) and the first block, and a patch embedding WITHOUT bias (timm: bias = not pre_norm).  The flat layout
                              keeps the bias slot (it stays zero and is not reported by vdk_vit_param_info) and gains norm_pre.weight / norm_pre.bias behind it. */
} VdkVitConfig;
typedef void (*vdk_grad_ready_fn)(void* user, int64_t offset, int64_t numel);

/* flat parameter layout: n_floats fp32 elements (params, grads, momentum, ema, and the bf16 copy `wb16` all use
 * it); tensors are reported in timm state_dict order with timm names; n_transposed = bf16 elements of `wt16`. */
int vdk_vit_param_count(const VdkVitConfig* cfg, int64_t* n_floats, int32_t* n_tensors, int64_t* n_transposed);
int vdk_vit_param_info(const VdkVitConfig* cfg, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel,
                       int64_t* shape4, int32_t* ndim);
int vdk_vit_workspace_bytes(const VdkVitConfig* cfg, size_t* bytes);
/* delayed scaling bookkeeping of the fp8 mode: scale = fmt_max / amax for every slot with amax > 0, amax := 0 */
int vdk_vit_fp8_update(const VdkVitConfig* cfg, void* stream);
int vdk_vit_refresh_weights(const VdkVitConfig* cfg, const float* params, void* wb16, void* wt16, int32_t skip_wb16, void* stream);
/* x f32 [B, in_chans, img, img] -> logits f32 [B, Cp], Cp = num_classes rounded up to 8 */
int vdk_vit_forward(const VdkVitConfig* cfg, const float* x, const float* params, const void* wb16, void* ws, size_t ws_bytes,
                    float* logits, void* stream);
/* dlogits bf16 [B, Cp] -> grads (flat fp32, overwritten).  on_ready: see csrc/vit_engine.hip (DDP bucket hook).
 * ALL FOUR ENGINES (vit / swin / convnext / resnet): `grads` has the layout of `params` (tensors at multiples of 64 floats) and must be ZERO-INITIALISED ONCE by its owner --
 * a backward overwrites every tensor's gradient but never the alignment gaps between tensors (a 1000-wide bias leaves 24 floats), and the passes over the whole flat buffer
 * (vdk_sumsq_f32 for the clip norm, vdk_allreduce_bucket, the fp16 overflow check of vdk_sgd_step_amp) read them. */
int vdk_vit_backward(const VdkVitConfig* cfg, const void* dlogits, const float* params, const void* wb16, const void* wt16, void* ws,
                     size_t ws_bytes, float* grads, vdk_grad_ready_fn on_ready, void* user, void* stream, void* side_stream);
/* PRECISE forward (evaluation / embedding extraction): same network, fp32 activations, every contraction on the fp32 MFMA, reads the fp32 master
 * weights; agrees with the reference's PyTorch-CPU fp32 path to ~1e-6 (north_star asks for 1e-3).  Nothing is kept for a backward. */
int vdk_vit_workspace_f32_bytes(const VdkVitConfig* cfg, size_t* bytes);
int vdk_vit_forward_f32(const VdkVitConfig* cfg, const float* x, const float* params, void* ws, size_t ws_bytes, float* logits, void* stream);
/* side_stream (may be NULL = same as stream): a second caller-owned hipStream_t that receives the weight-gradient GEMMs and
 * bias column sums; ordered against `stream` with events, joined before the call returns control of `stream`. */


/* ---- margin-softmax heads of the faceX / CBIR training path (fused with the cross-entropy) ------------------------------
 * ArcFace models/faceX/head/arcface.py:20-36, CircleLoss circleloss.py:21-43, MV_Softmax mv_softmax.py:25-44, followed by
 * nn.CrossEntropyLoss (engine/procedure/train.py:196).  weight W is [feat_dim, num_class] row-major like the reference's
 * Parameter.  Pipeline (visiondk_amd/heads.py): vdk_colnorm_fwd + vdk_rownorm_fwd -> cos = f^ W^ (vdk_gemm_bf16_nt, trans=1)
 * -> vdk_margin_ce (logits / loss / d cos) -> dW^ = f^T dcos (trans=1), df^ = dcos W^T (NT, split-K) -> vdk_colnorm_bwd,
 * vdk_rownorm_bwd.  MagFace (magface.py) returns a tuple the reference Trainer cannot consume: not built. */
#define VDK_HEAD_ARCFACE 0
#define VDK_HEAD_CIRCLE 1
#define VDK_HEAD_MV_AM 2
#define VDK_HEAD_MV_ARC 3
typedef struct VdkMarginHead {
  int32_t mode;        /* VDK_HEAD_* */
  float scale;         /* arcface / mv: scale;  circle: gamma */
  float margin;        /* arcface: margin_arc;  circle: margin;  mv: margin */
  float margin_am;     /* arcface only */
  float mv_weight;     /* mv only */
  const float* row_margin;   /* arcface only, optional (NULL): f32 [B] per-row margin_arc -- MagFace's magnitude-aware margin, models/faceX/head/magface.py:26-31 */
} VdkMarginHead;
/* ---- BatchNorm-based CNN pieces (timm ResNet BasicBlock family, `timm-resnet18` = BASELINE.json configs[0]) -------------------------
 * Convolutions run as implicit GEMMs (VdkGemmDesc.conv); these are the layouts and the non-GEMM layers around them.
 * vdk_conv_weight_prep: w f32 [Co][Ci][KH][KW] -> wf bf16 [Co][KH*KW*Cip] (forward operand, ci padded to Cip % 8 == 0 with zeros) and, if wd != NULL,
 * wd bf16 [Cip][KH*KW*Co] (input-gradient operand); vdk_conv_wgrad_unpermute: dWp f32 [Co][KH*KW*Cip] -> dW f32 [Co][Ci][KH][KW]. */
int vdk_conv_weight_prep(const float* w, void* wf, void* wd, int32_t Co, int32_t Ci, int32_t Cip, int32_t KH, int32_t KW, void* stream);
int vdk_conv_wgrad_unpermute(const float* dwp, float* dw, int32_t Co, int32_t Ci, int32_t Cip, int32_t KH, int32_t KW, void* stream);
/* image f32 NCHW -> bf16 NHWC with the channels zero-padded to Cp (the 7x7 stem becomes an ordinary implicit conv with Cin = 8) */
int vdk_nchw_to_nhwc_bf16(const float* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp, void* stream);
/* explicit im2col (bf16 NHWC -> [B*OH*OW][KH*KW*C]); used only for the weight gradient (TN GEMM dY^T . col) */
int vdk_im2col_bf16(const void* in, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                    void* stream);
/* nn.BatchNorm2d on NHWC rows x f32 [R, C] fused with the BasicBlock tail: out = [relu](bn(x) [+ res]) as bf16 and / or f32; res f32 or bf16.
 * Backward (training mode): dout f32 = gradient of out, out_bf16 = out (ReLU mask, NULL = no ReLU) -> dy_bf16 (gradient of x, the conv GEMMs' operand),
 * dres f32 (gradient of res = masked dout, optional), dgamma, dbeta.  sync != NULL turns both into SyncBatchNorm: forward all-reduces (sum x, sum x^2, count),
 * backward (sum g, sum g x^, count); dgamma / dbeta stay local sums like torch.nn.SyncBatchNorm (the gradient all-reduce follows). */
/* SyncBatchNorm hook (the reference's opt-in, engine/vision_engine.py:224-225): called on the host with a device vector of n floats whose producer kernel is
 * already enqueued on `stream`; the callee enqueues a SUM all-reduce of it over the data-parallel ranks, ordered before later work on `stream`. */
typedef void (*vdk_stat_sync_fn)(void* user, float* stats, int64_t n);
int vdk_bn_rows_workspace_bytes(int64_t R, int32_t C, size_t* bytes);
int vdk_bn_act_fwd(const float* x, int64_t R, int32_t C, const float* gamma, const float* beta, float eps, float momentum, int32_t training, float* running_mean,
                   float* running_var, const float* res_f32, const void* res_bf16, int32_t relu, void* out_bf16, float* out_f32, float* save_mean, float* save_invstd,
                   void* ws, size_t ws_bytes, vdk_stat_sync_fn sync, void* user, void* stream);
int vdk_bn_act_bwd(const float* x, const float* dout, const void* out_bf16, int64_t R, int32_t C, const float* gamma, const float* save_mean, const float* save_invstd,
                   void* dy_bf16, float* dres, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, vdk_stat_sync_fn sync, void* user, void* stream);
/* vdk_bn_act_bwd without an activation mask and with the input gradient in fp32: the BatchNorm2d / BatchNorm1d of the embedding neck
 * (models/faceX/backbone/timm_wrapper.py:30-38) as SyncBatchNorm (sync != NULL); its forward is vdk_bn_act_fwd(relu = 0, out_f32). */
int vdk_bn_rows_bwd(const float* x, const float* dout, int64_t R, int32_t C, const float* gamma, const float* save_mean, const float* save_invstd, float* dx,
                    float* dgamma, float* dbeta, void* ws, size_t ws_bytes, vdk_stat_sync_fn sync, void* user, void* stream);
/* nn.MaxPool2d(3, 2, 1) on bf16 NHWC (backward routes to the FIRST maximum in (ky, kx) order, like torch) and global average pooling.
 * argmax (optional, uint8 [B, OH, OW, C]): the winning window position 0..8 written by the forward; the backward uses it when given (in may then be NULL),
 * else it re-derives the winners from `in`. */
int vdk_maxpool3s2_fwd(const void* in, void* out, uint8_t* argmax, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int vdk_maxpool3s2_bwd(const void* in, const uint8_t* argmax, const float* dout, float* din, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int vdk_avgpool_fwd(const void* in, void* out, int32_t B, int32_t Bp, int32_t HW, int32_t C, void* stream);
int vdk_avgpool_bwd(const void* dfeat, int64_t ld, float* dout, int32_t B, int32_t HW, int32_t C, void* stream);

/* ---- the step in front of the path: validation-time input pipeline ------------------------------------------------
 * Replaces, for a whole batch, the per-image CPU chain the val / CBIR-gallery dataloaders run (configs/classification/pet.yaml:94-101,
 * configs/faceX/cbir.yaml:92-99):  ResizeAndPadding2Square(size, training=False) (dataset/transforms.py:325-362: long side to `size` with PIL BILINEAR,
 * `int(side * (size / max_side))`, centred on a zero canvas)  ->  to_tensor (:466-468: uint8 HWC -> float32 CHW / 255)  ->  normalize (:474-477).
 * Bit-exact with Pillow's 8-bit two-pass resampling and torchvision's float32 expressions (oracle/preprocess_ref.py).
 *   pixels   device, uint8, 16-byte aligned: decoded RGB images back to back, each HWC with row stride 3*w; image b starts at byte offsets[b]; the
 *            allocation must extend to a multiple of 16 bytes (rows are staged with aligned 16-byte loads)
 *   offsets  device int64 [B];   wh  device int32 [B][2] = (width, height);   max_side  >= every width and height (host-known; sizes the coefficient tables)
 *   out      device float32 [B][3][S][S];   S <= 1024, sides <= 8192
 *   status   device int32 [B] or NULL: 0 ok; 1 = an output side truncates to 0 (PIL raises ValueError("height and width must be > 0")), the image's
 *            output is then the normalised zero canvas; 2 = non-positive side or side > 8192 */
/* global average pool over the HW rows of every image of an f32 NHWC map, and its backward (f32 and / or bf16 map gradient): timm's classifier heads */
int vdk_avgpool_rows_f32_fwd(const float* in, float* out, int32_t B, int32_t HW, int32_t C, void* stream);
int vdk_avgpool_rows_f32_bwd(const float* dpool, float* dmap, void* dmap_bf16, int32_t B, int32_t HW, int32_t C, void* stream);
/* x[i] *= scale[0] (reciprocal != 0: x[i] /= scale[0]) with the factor read from DEVICE memory when the kernel runs: GradScaler's loss scale entering a gradient tensor at the
 * boundary of an fp16 graph (`scaler.scale(loss).backward()`, engine/procedure/train.py:205; FaceTrainStep multiplies d(loss)/d(embedding)), and a gradient region kept at a
 * power-of-two scale returning to its size (the ConvNeXt engine's fc1 gradients under fp16 operands).  x 16-byte aligned. */
int vdk_scale_dev_f32(float* x, int64_t n, const float* scale, int32_t reciprocal, void* stream);
int vdk_preprocess_workspace_bytes(int32_t B, int32_t S, int32_t max_side, size_t* bytes);
int vdk_preprocess_resize_pad_normalize(const uint8_t* pixels, const int64_t* offsets, const int32_t* wh, int32_t B, int32_t S, int32_t max_side, float mean0,
                                        float mean1, float mean2, float std0, float std1, float std2, float* out, int32_t* status, void* ws, size_t ws_bytes,
                                        void* stream);

/* ---- native ConvNeXt engine: timm ConvNeXt in feature mode (num_classes=0, global_pool='') over flat buffers --------------------
 * Replaces `self.model(x)` of TimmWrapper.forward (models/faceX/backbone/timm_wrapper.py:16-21,51) and its backward for the CNN
 * backbones of the face / CBIR path (`convnext_base`, configs/faceX/cbir.yaml:4-8).  Semantics restated from timm 0.9.16 (not vendored;
 * oracle/convnext_ref.py): stem Conv4x4/4 + LayerNorm2d, stages of [LayerNorm2d + Conv2x2/2] + ConvNeXtBlocks (dwconv7x7, LayerNorm, fc1, GELU,
 * fc2, layer scale, shortcut), head.norm.  Input NCHW f32 as the dataloader provides it; output the head-normed map as NHWC rows. */
typedef struct VdkConvNextConfig {
  int32_t batch, img_size, in_chans;
  int32_t depths[4];
  int32_t dims[4];
  float ln_eps;
  int32_t num_classes;   /* 0: feature mode (num_classes=0, global_pool=''), what TimmWrapper builds; > 0: timm's classifier head -- global average pool ->
                          * head.norm -> head.fc -- what VisionWrapper.create_model builds (models/classifier/classify_model.py:49-54, `timm-convnext_*`) */
  int32_t operand;       /* VDK_BF16 (0) | VDK_F16: format of the GEMM operands, the saved 16-bit activations and the 16-bit gradient tensors (wb16 / wx copies included).
                          * VDK_F16 is what the face / CBIR step uses to stay within 1e-3 of the reference's fp32 loop (engine/procedure/train.py:217-227, no autocast there) and
                          * what the classifier loop's autocast computes in (train.py:118); the caller multiplies the loss scale into dout (GradScaler, train.py:205) and
                          * un-scales in the optimizer (vdk_sgd_step_amp).  The layer scale is then applied in the fc2 epilogue (VdkGemmDesc.col_scale), see csrc/convnext_engine.hip */
} VdkConvNextConfig;
/* flat parameter layout (timm state_dict order and names) + size of `wx`, the derived operand copies kept next to `wb16` */
int vdk_convnext_param_count(const VdkConvNextConfig* cfg, int64_t* n_floats, int32_t* n_tensors, size_t* wx_bytes);
int vdk_convnext_param_info(const VdkConvNextConfig* cfg, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int64_t* shape4,
                            int32_t* ndim);
int vdk_convnext_workspace_bytes(const VdkConvNextConfig* cfg, size_t* bytes);
int vdk_convnext_refresh_weights(const VdkConvNextConfig* cfg, const float* params, void* wb16, void* wx, int32_t skip_wb16, void* stream);
/* x f32 [B, in_chans, img, img] -> out f32 [B*(img/32)^2, dims[3]] (row (b, y, x), channel-contiguous); classifier mode: logits f32 [B, up(num_classes, 8)] */
int vdk_convnext_forward(const VdkConvNextConfig* cfg, const float* x, const float* params, const void* wb16, const void* wx, void* ws, size_t ws_bytes,
                         float* out, void* stream);
/* PRECISE forward, as vdk_vit_forward_f32 (wx: only the tap-major depthwise weights are read from it) */
int vdk_convnext_workspace_f32_bytes(const VdkConvNextConfig* cfg, size_t* bytes);
int vdk_convnext_forward_f32(const VdkConvNextConfig* cfg, const float* x, const float* params, const void* wx, void* ws, size_t ws_bytes, float* out,
                             void* stream);
/* The same engine TRAINING in fp32-class arithmetic (feature mode, num_classes = 0): fp32 activations, every contraction on the fp32 MFMA, library erff / expf -- the arithmetic of
 * the reference's face / CBIR loop, which runs without autocast (engine/procedure/train.py:217-227).  forward keeps in `ws` what backward needs; grads: flat f32, param layout,
 * fully overwritten; on_ready as in vdk_vit_backward.  `wx` only supplies the tap-major depthwise weights (vdk_convnext_refresh_weights). */
int vdk_convnext_train_f32_workspace_bytes(const VdkConvNextConfig* cfg, size_t* bytes);
int vdk_convnext_forward_train_f32(const VdkConvNextConfig* cfg, const float* x, const float* params, const void* wx, void* ws, size_t ws_bytes, float* out, void* stream);
int vdk_convnext_backward_train_f32(const VdkConvNextConfig* cfg, const float* dout, const float* params, const void* wx, void* ws, size_t ws_bytes, float* grads,
                                    vdk_grad_ready_fn on_ready, void* user, void* stream);
/* dout f32 (same shape as out) -> grads (flat fp32, overwritten); on_ready as in vdk_vit_backward.  Classifier mode: dout = dlogits bf16
 * [up(B, 64), up(num_classes, 8)], padding rows / columns zero (as vdk_resnet_backward takes it) */
int vdk_convnext_backward(const VdkConvNextConfig* cfg, const void* dout, const float* params, const void* wb16, const void* wx, void* ws, size_t ws_bytes,
                          float* grads, vdk_grad_ready_fn on_ready, void* user, void* stream);

/* ---- native Swin engine: timm `swin_{tiny,small,base,large}_patch4_window7_224` over flat buffers (csrc/swin_engine.hip) -------------------------------------------------
 * Replaces `self.model(images)` + `loss.backward()` for the backbone BOTH shipped configs of the reference select by default (`timm-swin_base_patch4_window7_224`:
 * configs/classification/pet.yaml:25, configs/faceX/cbir.yaml:26; built by timm.create_model in models/classifier/classify_model.py:49-54 and
 * models/faceX/backbone/timm_wrapper.py:16-21).  Semantics restated from timm 0.9.16 (oracle/swin_ref.py, pinned against transformers.SwinModel).  Same protocol as the ViT
 * engine: flat fp32 params / grads in timm state_dict order with timm names (vdk_swin_param_info), wb16 = the bf16 copy in the same layout, wt16 = [in, out] bf16 copies of
 * every Linear for the input-gradient GEMMs (vdk_swin_refresh_weights), one call per forward and one per backward on one stream, activations in the caller's workspace.
 * num_classes = 0 is timm's forward_features: the final-normed NHWC map as rows f32 [B * 49 * (img / 224)^2, 8 * embed_dim] (what TimmWrapper's neck consumes). */
typedef struct VdkSwinConfig {
  int32_t batch, img_size, in_chans, embed_dim;
  int32_t depths[4], heads[4];
  int32_t num_classes;
  float ln_eps;
  int32_t operand;        /* VDK_BF16 | VDK_F16: format of the GEMM / attention operands and of the saved 16-bit activations (fp16 = the reference's autocast dtype, train.py:118) */
  const float* drop_path; /* NULL (evaluation, or drop_path_rate = 0), or DEVICE f32 [2 * sum(depths)][batch]: stochastic depth.  timm builds swin_* with drop_path_rate = 0.1 when the
                           * reference calls timm.create_model(name, ...) (models/classifier/classify_model.py:49-54; rates linspace(0, 0.1, sum(depths)) over the blocks): in
                           * training every block's two branches are multiplied per SAMPLE by mask / keep_prob before the shortcut is added.  Row 2 * k is block k's attention
                           * branch, row 2 * k + 1 its MLP branch; entry [b] = 0 or 1 / keep_prob, drawn by the caller per step; forward and the backward of that forward get the
                           * same buffer.  The proj / fc2 epilogues apply it (VdkGemmDesc.row_scale), the backward scales the 16-bit gradient copies the branch GEMMs read. */
} VdkSwinConfig;
int vdk_swin_param_count(const VdkSwinConfig* cfg, int64_t* n_floats, int32_t* n_tensors, int64_t* n_transposed);
int vdk_swin_param_info(const VdkSwinConfig* cfg, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int64_t* shape4, int32_t* ndim);
int vdk_swin_workspace_bytes(const VdkSwinConfig* cfg, size_t* bytes);
int vdk_swin_refresh_weights(const VdkSwinConfig* cfg, const float* params, void* wb16, void* wt16, int32_t skip_wb16, void* stream);
/* x f32 [B, in_chans, img, img] -> out f32: logits [B, up(num_classes, 8)] or the feature rows (see above); activations stay in ws */
int vdk_swin_forward(const VdkSwinConfig* cfg, const float* x, const float* params, const void* wb16, void* ws, size_t ws_bytes, float* out, void* stream);
/* dout: dlogits bf16 [B, up(num_classes, 8)] (padding columns zero), or f32 feature-row gradients -> grads (flat fp32, overwritten); on_ready as in vdk_vit_backward.
 * `grads` must be ZERO-INITIALISED ONCE by its owner: every tensor's gradient is overwritten by each call, but the alignment gaps of the flat layout (tensors start at multiples
 * of 64 floats: the 96 / 192-wide vectors and the 169 x heads tables leave some) are never written, and vdk_sumsq_f32 (the clip norm), vdk_allreduce_bucket and the fp16
 * overflow check all run over the whole n_floats. */
int vdk_swin_backward(const VdkSwinConfig* cfg, const void* dout, const float* params, const void* wb16, const void* wt16, void* ws, size_t ws_bytes, float* grads,
                      vdk_grad_ready_fn on_ready, void* user, void* stream);

/* ---- collectives of the multi-GPU paths for a host without a process group of its own: RCCL over xGMI, one process per GPU ------------------------------
 * Replaces, for a C / C++ host, what the reference gets from torch.distributed: the process group of main.py:39-40 and the gradient all-reduce of the DistributedDataParallel
 * wrap (engine/vision_engine.py:313,510), plus the query all-gather of the gallery-sharded search.  (The Python host of this repository drives the same exchange through c10d,
 * visiondk_amd/comm.py.)  librccl is dlopen()ed at first use -- environment VDK_RCCL_LIB names it explicitly -- so single-GPU hosts never load it.
 *   vdk_comm_unique_id : 128 bytes (ncclUniqueId) created on one rank; the host distributes them to every rank (its own rendezvous).
 *   vdk_comm_init      : collective over all ranks, on the caller's current device: communicator + a dedicated stream for the collectives.
 *   vdk_allreduce_bucket : SUM all-reduce of grads[offset, offset + numel) in place, ordered after everything enqueued on launch_stream so far, on the communicator's stream:
 *                          called from the vdk_grad_ready_fn callback of vdk_vit_backward / vdk_convnext_backward / vdk_resnet_backward it overlaps the rest of the backward.
 *                          The 1/world factor belongs to the optimizer (vdk_sgd_step's grad_scale).
 *   vdk_comm_finish    : launch_stream waits on the device for every collective issued since the last finish (call before vdk_sumsq_f32 / vdk_sgd_step).
 *   vdk_allgather      : recv[r * bytes_per_rank, ...) = rank r's `send` (the sharded search gathers every rank's query embeddings); ordered with launch_stream both ways. */
typedef struct VdkComm VdkComm;
int vdk_comm_unique_id(void* id128);
int vdk_comm_init(const void* id128, int32_t rank, int32_t world, VdkComm** comm);
int vdk_comm_destroy(VdkComm* comm);
int vdk_comm_rank(const VdkComm* comm);
int vdk_comm_world(const VdkComm* comm);
/* Timing trace of the collectives against the launch stream (diagnostic; the one-GPU evidence that a bucket's all-reduce runs while the backward is still executing --
 * torch DDP's overlap at engine/vision_engine.py:313,510).  vdk_comm_trace(c, 1) clears and arms it: every vdk_allreduce_bucket then records a start event on the collectives'
 * stream, vdk_comm_trace_close_last its end event (after anything the caller put behind it on vdk_comm_stream(c): tests enqueue vdk_debug_occupy_cus there as a stand-in for a
 * multi-GPU collective's kernel), vdk_comm_mark a time stamp on the launch stream.  vdk_comm_trace_read SYNCHRONISES and returns milliseconds after mark 0:
 * ar_ms [2 * n_ar] = (start, end) pairs, numel [n_ar], marks_ms [n_marks]. */
int vdk_comm_trace(VdkComm* c, int32_t enable);
int vdk_comm_trace_close_last(VdkComm* c);
int vdk_comm_mark(VdkComm* c, void* launch_stream);
int vdk_comm_trace_read(VdkComm* c, float* ar_ms, int64_t* numel, int32_t ar_cap, int32_t* n_ar, float* marks_ms, int32_t marks_cap, int32_t* n_marks);
void* vdk_comm_stream(VdkComm* c);
int vdk_allreduce_bucket(VdkComm* comm, float* grads, int64_t offset, int64_t numel, void* launch_stream);
int vdk_comm_finish(VdkComm* comm, void* launch_stream);
int vdk_allgather(VdkComm* comm, const void* send, void* recv, int64_t bytes_per_rank, void* launch_stream);

/* ---- native ResNet engine: timm BasicBlock ResNets (resnet18 / resnet34) over flat buffers ---------------------------------------------
 * Replaces `self.model(images)` + `loss.backward()` for the classifier built by timm.create_model('resnet18', num_classes=C)
 * (models/classifier/classify_model.py:49-54; `timm-resnet18` is the reference's CPU plumbing config).  Semantics restated from timm 0.9.16
 * (oracle/resnet_ref.py).  Every convolution is an implicit GEMM (VdkGemmDesc.conv) in forward and input gradient. */
typedef struct VdkResNetConfig {
  int32_t batch, img_size, in_chans;
  int32_t widths[4];
  int32_t depths[4];
  int32_t num_classes;
  float bn_eps, bn_momentum;
  int32_t mid[4];        /* all 0: BasicBlock network (resnet18 / 34), `widths` = block channels.  > 0: Bottleneck network (resnet50 / 101 / 152,
                          * wide_resnet*_2): inner width of the 1x1 -> 3x3 -> 1x1 blocks, `widths` = block OUTPUT channels (4 x planes) */
  int32_t stem_width;    /* 0: widths[0] (basic) or 64 (bottleneck) */
  int32_t operand_dtype; /* VDK_BF16 (0, the default of a zeroed config) or VDK_F16: the 16-bit format of every convolution / fc operand, saved activation and gradient operand --
                          * IEEE half is what the reference's `torch.autocast(device_type=...)` computes in on a GPU (engine/procedure/train.py:118), with GradScaler's loss scale
                          * carrying the gradients (dlogits arrives scaled, the optimizer un-scales).  8x smaller operand rounding at the same MFMA rate. */
} VdkResNetConfig;
/* the calling thread's 16-bit format for the NHWC / BatchNorm / pooling functions of csrc/resnet_ops.hip (vdk_nchw_to_nhwc_bf16, vdk_bn_act_*, vdk_maxpool3s2_*, vdk_avgpool_*,
 * vdk_conv_weight_prep): VDK_BF16 (default) | VDK_F16.  The engine entry points below set it from VdkResNetConfig.operand_dtype and restore bf16 on return. */
int vdk_resnet_ops_format(int32_t dtype);
/* trainable parameters live in one flat f32 buffer (params / grads), BatchNorm running statistics in another (buffers); `wx` = derived operand copies */
int vdk_resnet_param_count(const VdkResNetConfig* cfg, int64_t* n_floats, int32_t* n_tensors, int64_t* n_buffer_floats, int32_t* n_buffers, size_t* wx_bytes);
/* which = 0: parameters, 1: buffers (running_mean / running_var); timm state_dict names */
int vdk_resnet_param_info(const VdkResNetConfig* cfg, int32_t which, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int64_t* shape4,
                          int32_t* ndim);
int vdk_resnet_workspace_bytes(const VdkResNetConfig* cfg, size_t* bytes);
int vdk_resnet_refresh_weights(const VdkResNetConfig* cfg, const float* params, void* wb16, void* wx, int32_t skip_wb16, void* stream);
/* x f32 [B, in_chans, img, img] -> logits f32 [B, Cp] (Cp = num_classes rounded up to 8); training != 0: batch statistics + running update */
int vdk_resnet_forward(const VdkResNetConfig* cfg, const float* x, const float* params, float* buffers, const void* wb16, const void* wx, int32_t training, void* ws,
                       size_t ws_bytes, float* logits, vdk_stat_sync_fn bn_sync, void* bn_user, void* stream);
/* dlogits bf16 [B, Cp] (padding columns zero) -> grads (flat f32, overwritten); on_ready as in vdk_vit_backward */
int vdk_resnet_backward(const VdkResNetConfig* cfg, const void* dlogits, const float* params, const void* wb16, const void* wx, void* ws, size_t ws_bytes, float* grads,
                        vdk_grad_ready_fn on_ready, void* user, vdk_stat_sync_fn bn_sync, void* bn_user, void* stream);
/* bn_sync != NULL: every BatchNorm of the network runs as SyncBatchNorm (pass the same hook to forward and backward) */

/* timm AttentionPoolLatent (global_pool='map', SigLIP): ONE projected latent query q (f32 [H*64]) against the N tokens of every image; kv = bf16 [B*N, ldkv]
 * output of the kv Linear (k | v halves of H*64 each).  fwd: out f32 [B, ldo] = softmax_n(scale <q_h, k_n>) . v; probs f32 [B, H, N] (optional, for the backward).
 * bwd: dout f32 [B, lddo] -> dkv bf16 [B*N, lddkv], dq_part f32 [B, H*64] (sum the rows -> dL/dq).  head_dim 64, N <= 4096. */
int vdk_attn_pool_fwd(const float* q, const void* kv, int64_t ldkv, int32_t B, int32_t N, int32_t H, float scale, float* out, int64_t ldo, float* probs, void* stream);
int vdk_attn_pool_bwd(const float* q, const void* kv, int64_t ldkv, const float* probs, const float* dout, int64_t lddo, int32_t B, int32_t N, int32_t H, float scale,
                      void* dkv, int64_t lddkv, float* dq_part, void* stream);
/* the same with the 16-bit format of kv / dkv as a parameter (VDK_BF16 | VDK_F16: the trunk's operand format; fp16 = the reference's autocast arithmetic, train.py:118) */
int vdk_attn_pool_fwd_dt(const float* q, const void* kv, int64_t ldkv, int32_t B, int32_t N, int32_t H, float scale, float* out, int64_t ldo, float* probs, int32_t dtype,
                         void* stream);
int vdk_attn_pool_bwd_dt(const float* q, const void* kv, int64_t ldkv, const float* probs, const float* dout, int64_t lddo, int32_t B, int32_t N, int32_t H, float scale,
                         void* dkv, int64_t lddkv, float* dq_part, int32_t dtype, void* stream);

/* F.normalize(W, dim=0): inv[c] = 1/max(||W[:,c]||, eps); Wb = bf16 [3D, ldb]: the normalised weight as split planes
 * (hi, hi, lo) stacked along the contraction dim (rows [0,D) alone are the plain bf16 copy); columns C..Cp-1 zero */
int vdk_colnorm_fwd(const float* W, int64_t ldw, int32_t D, int32_t C, int32_t Cp, float eps, float* inv, void* Wb, int64_t ldb, int32_t planes, void* stream);
int vdk_colnorm_bwd(const float* W, int64_t ldw, const float* inv, const float* dWh, int64_t ldg, int32_t D, int32_t C, float* dW, int64_t ldo,
                    void* stream);
/* vdk_colnorm_fwd / vdk_rownorm_fwd with the format of the planes as a parameter: dtype = VDK_BF16 | VDK_F16.  fp16 planes (hi = fp16(v), lo = fp16(v - hi)) are the
 * operands of a head that runs under GradScaler with fp16 gradients (FaceTrainStep over an fp16 backbone): the cosines keep their fp32-class accuracy (|v| <= 1, the low
 * plane's subnormals resolve 6e-8), and the two backward products read the hi plane with 8x less operand rounding than bf16's. */
int vdk_colnorm_fwd_dt(const float* W, int64_t ldw, int32_t D, int32_t C, int32_t Cp, float eps, float* inv, void* Wb, int64_t ldb, int32_t planes, int32_t dtype, void* stream);
int vdk_rownorm_fwd_dt(const float* f, int32_t B, int32_t Bp, int32_t D, float eps, float* fh, void* fb, void* fbt, float* inv, int32_t planes, int32_t dtype, void* stream);
/* F.normalize(feats): fh f32 [B, D]; fb bf16 [Bp, D]; fbt bf16 [3D, Bp] = transposed split planes (hi, lo, hi), so that the
 * K = 3D GEMM fbt^T . Wb accumulates hi*hi + lo*hi + hi*lo (fp32-class cos); rows/cols B..Bp-1 zero; inv f32 [B] */
int vdk_rownorm_fwd(const float* f, int32_t B, int32_t Bp, int32_t D, float eps, float* fh, void* fb, void* fbt, float* inv, int32_t planes, void* stream);
int vdk_rownorm_bwd(const float* fh, const float* inv, const float* dfh, int64_t lddfh, int32_t B, int32_t D, float* df, void* stream);
/* The fused form of the wide heads (the B x C cosines never reach memory as fp32): the cos GEMM f^ W^ (TN: fbt bf16 [K, Bp], wb bf16 [K, Cp]; K = D or 3 D split
 * planes) runs twice with the head applied to the tile in registers.  pass 1 -> stats f32 [B][ceil(Cp / 64)][4] (per 64-column slice: max logit, sum exp(logit - max), sum
 * logit) and tlogit [B]; vdk_margin_rowstat -> rowstat f32 [B][2] = (row max, 1 / sum exp) and loss_rows; pass 2 -> dcos bf16 [Bp, lddc] = grad_scale * dLoss/dcos (rows >= B,
 * columns >= C zero).  gt (MV-Softmax only, else NULL): the target cosines, vdk_margin_target_cos_direct.  VDK_EUNSUPPORTED when the 256x256 TN kernel cannot serve the shape. */
int vdk_margin_cos_pass(const VdkMarginHead* h, int32_t pass, const void* fbt, int64_t ld_f, const void* wb, int64_t ld_w, int32_t B, int32_t Bp, int32_t C, int32_t Cp, int32_t K,
                        const int64_t* labels, const float* gt, float* stats, float* tlogit, const float* rowstat, float label_smoothing, float grad_scale, void* dcos,
                        int64_t lddc, void* stream);
int vdk_margin_rowstat(const float* stats, int64_t nslice, const float* tlogit, int32_t B, int32_t C, float label_smoothing, float* rowstat, float* loss_rows, void* stream);
int vdk_margin_target_cos_direct(const void* fbt, int64_t ld_f, const void* wb, int64_t ld_w, int32_t K, int32_t B, const int64_t* labels, float* gt, void* stream);   /* gt[b] = sum_k fbt[k][b] wb[k][y_b] */
/* cos f32 [B, ldc] -> any of: logits f32 [B, ldl] (what the reference head returns), loss_rows f32 [B] (CE with optional label
 * smoothing), dcos bf16 [B, lddc] = grad_scale * dLoss/dcos (padding columns zeroed) */
int vdk_margin_ce(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, float label_smoothing,
                  float grad_scale, float* logits, int64_t ldl, float* loss_rows, void* dcos_bf16, int64_t lddc, void* stream);
/* vdk_margin_ce under GradScaler (`scaler.scale(loss).backward()`, engine/procedure/train.py:205): dcos (dc_dtype = VDK_BF16 | VDK_F16) = loss_scale[0] * grad_scale *
 * dLoss/dcos with the scale read from DEVICE memory (NULL: 1); loss_rows and logits are never scaled */
int vdk_margin_ce_amp(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, float label_smoothing, float grad_scale,
                      const float* loss_scale, float* logits, int64_t ldl, float* loss_rows, void* dcos16, int64_t lddc, int32_t dc_dtype, void* stream);
/* the same with d(loss)/d(cos) left in fp32 [B, lddc] (padding columns zeroed): head of the fp32-class training mode (FaceTrainStep(precision="fp32")) */
int vdk_margin_ce_f32(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, float label_smoothing, float grad_scale,
                      float* loss_rows, float* dcos_f32, int64_t lddc, void* stream);
/* backward of the logits-returning form: dcos bf16 = dlogits * d(logit)/d(cos) */
/* Class-sharded margin head for data-parallel training with a large identity count (instead of all-reducing the [D, C] head gradient --
 * 2 GB at C = 10^6 -- every rank keeps a column shard of the head, the features are all-gathered and only per-row scalars and the [B, D] feature gradient
 * cross the links).  The fused vdk_margin_ce splits into three local passes around the collectives: target cosine (SUM), per-row statistics
 * (max, sum exp(. - max), sum logit, target logit: MAX / rescaled SUM / SUM / SUM), gradient with the global max and sum.  cosv is the shard's [B, Cloc] block,
 * global column = c_base + local column; same margins as arcface.py / circleloss.py / mv_softmax.py (see vdk_margin_ce). */
int vdk_margin_target_cos(const float* cosv, int64_t ldc, int32_t B, int32_t Cloc, int64_t c_base, const int64_t* labels, float* gt, void* stream);
int vdk_margin_stats(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t Cloc, int64_t c_base, const int64_t* labels, const float* gt,
                     float* stats /* [B][4] */, void* stream);
int vdk_margin_grad(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t Cloc, int64_t c_base, int64_t C_total, const int64_t* labels,
                    const float* gt, const float* gmax, const float* gsum, float label_smoothing, float grad_scale, void* dcos_bf16, int64_t lddc, void* stream);
int vdk_margin_bwd(const VdkMarginHead* h, const float* cosv, int64_t ldc, int32_t B, int32_t C, const int64_t* labels, const float* dlogits,
                   int64_t lddl, void* dcos_bf16, int64_t lddc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VISIONDK_H */

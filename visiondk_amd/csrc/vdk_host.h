// vdk_host.h — host-side plumbing shared by the C-ABI entry points: error codes, thread-local
// last-error string, launch checking.  Every entry point returns 0 or a negative code; it never
// allocates, never synchronises and borrows all pointers (SURVEY.md §8(b) contract).
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "visiondk.h"

int vdk_fail(int code, const char* msg);
int vdk_check_launch(const char* what);

// In-library helper (C++ linkage, not part of the C ABI): many transpose + cast jobs (out[c][r] bf16 = in[r][c] f32, rows [R, Rpad) zero-filled) in ONE launch.
// A weight refresh of a 12-layer ViT is 49 such jobs of ~10 us each; as separate launches they sit at the launch-latency floor.
struct VdkTcItem { const float* in; void* out; int ldi, R, C, ldo, Rpad; };
int vdk_transpose_cast_batch(const VdkTcItem* items, int n, void* stream, int opf = 0 /* output format: VDK_OPF_BF16 | VDK_OPF_F16 (vdk_device.h) */);
// out bf16 [R, ldo] = in f32 [R, C] (row stride ldi) with the columns [C, ldo) zero-filled (operand copies of weights whose row length is not a multiple of 8)
int vdk_cast_pad_rows(const float* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, void* stream, int opf = 0);
// vdk_patchify_bf16 / vdk_cast_f32_bf16 with the 16-bit output format as a parameter
int vdk_patchify_16(const float* x, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* out, int32_t Kp, int opf, void* stream);
int vdk_cast_f32_16(const float* in, void* out, int64_t n, int opf, void* stream);

// fp8 copy of a LayerNorm kernel's bf16 output (fp8 mode of the ViT engine): out [rows, ld] bytes = fp8(clamp(value * scale[0])), amax[0] = max(amax[0], max |value|)
struct LnQ8 { unsigned char* out; long ld; const float* scale; float* amax; int fmt; };
// out[c] = scale * sum_{s < S} in[s * ld + c], c < n: one of up to 8 row reductions that vdk_reduce_rows_batch runs in a single launch
// `in` is SCRATCH of the caller: a tall job (S >= 1024 partial rows) is folded IN PLACE onto its first 64 rows before the final reduction (reduce_rows_fold_kernel), so
// the partial buffer must not be read again afterwards and two jobs of a batch must not share rows of it.  Every in-library producer (LayerNorm / column-sum / depthwise
// weight-gradient partials) hands its own workspace slice; that is why the pointer is not const.
// one block's additive attention tile for vdk_wa_prep_table_batch (csrc/window_attention.hip): bm <- table (+ mask of nW windows when mask != null)
struct VdkWaPrepJob { const float* table; const float* mask; float* bm; int nW, H; };
int vdk_wa_prep_table_batch(const VdkWaPrepJob* jobs, int n, void* stream);
struct VdkReduceJob { float* in; long ld; int S; long n; float* out; float scale; };
int vdk_reduce_rows_batch(const VdkReduceJob* jobs, int n, void* stream);

int vdk_colsum_bf16_deferred(const void* in, int64_t ld, int32_t T, int32_t N, float* out, void* ws, size_t ws_bytes, void* stream, VdkReduceJob* job,
                             const LnQ8* q8 = nullptr /* the pass also writes the fp8 copy of `in` (rows of ld bytes) and its amax */,
                             int opf = 0 /* format of `in`: VDK_OPF_BF16 | VDK_OPF_F16 (vdk_device.h) */);
int vdk_colsum_16(const void* in, int64_t ld, int32_t T, int32_t N, float* out, void* ws, size_t ws_bytes, int opf, void* stream);
// vdk_dwconv7_wgrad whose reduction over the per-slice partials [S][49 C | C] is left to the caller (needs db == dw + 49 C: the flat gradient buffer's layout)
int vdk_dwconv7_wgrad_deferred(const float* in, const float* dy, float* dw, float* db, int32_t B, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, void* stream,
                               VdkReduceJob* job);
int vdk_layernorm_bwd_deferred(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                               const float* dres, int64_t lddres, int32_t T, int32_t C, float* dx, int64_t lddx, void* dxb, int64_t lddxb, float* dgamma, float* dbeta,
                               void* ws, size_t ws_bytes, void* stream, VdkReduceJob* job,
                               float* dxb_colsum = nullptr /* [C]: column sums of the bf16 output dxb (a Linear's bias gradient), reduction described by *job2 */, VdkReduceJob* job2 = nullptr,
                               const LnQ8* dxb_q8 = nullptr /* with dxb_colsum, C <= 1024, bf16 dy: dxb's fp8 copy rides along */,
                               int opf = 0 /* format of a 16-bit dy and of dxb (dy_dtype VDK_F16 implies fp16; an fp32 dy takes it from here) */,
                               const float* dy_scale = nullptr /* device scalar multiplied into dy as it is loaded (job == nullptr: the reductions run inside) */,
                               const float* dxb_rs = nullptr /* per-sample factor of the 16-bit copy: dxb = 16bit(dx * dxb_rs[row / dxb_rps]) (stochastic depth) */, int dxb_rps = 1,
                               const void* dres16 = nullptr /* the residual gradient as 16-bit rows (fp16, pitch lddres) instead of fp32 `dres`: the ViT engine's fp16 mode keeps the
                               residual-gradient stream in 16 bits (its copy for the next GEMM IS the stream) */);
// x16[r, :] *= rs[r / rps] in place (16-bit operand format opf): the same factor for a 16-bit gradient copy that no LayerNorm backward produced (stage boundaries of the Swin engine)
int vdk_rowscale_16(void* x16, int64_t T, int32_t C, const float* rs, int32_t rps, int opf, void* stream);

// conv.hip: every ConvNeXt block's weight preparation in one launch (depthwise weight tap-major, layer scale folded into fc2 in both orientations); in-library
struct CnPrepJob { const float* dw_w; float* dwt; const float* w2; const float* b2; const float* gamma; unsigned short* w2p; unsigned short* w2pt; float* b2p; int C, M;
                   float* rs; /* NULL: bf16 operands, gamma folded into both copies.  else f32 [2] = {r, 1 / r}: fp16 operands (see cn_prep_batch_kernel) */ };
int vdk_convnext_prep_blocks(const CnPrepJob* jobs, int n, void* stream);

// conv.hip kernels with the 16-bit output format as a parameter (in-library; the C-ABI names are the bf16 forms)
extern "C" int vdk_dwconv7_fwd_16(const float* in, const float* wt, const float* bias, const float* res, float* out, void* out16, int32_t B, int32_t H, int32_t W, int32_t C,
                                  int32_t flip, int32_t opf, void* stream);
extern "C" int vdk_avgpool_rows_f32_bwd_16(const float* dpool, float* dmap, void* dmap16, int32_t B, int32_t HW, int32_t C, int32_t opf, void* stream);
extern "C" int vdk_conv2x2_weight_prep_16(const float* w, void* wb, void* wtb, int32_t Co, int32_t Ci, int32_t opf, void* stream);
// vdk_colsum_f32 (csrc/gemm_f32.hip) with the reduction over its row splits left to the caller's vdk_reduce_rows_batch
int vdk_colsum_f32_deferred(const float* x, int64_t ld, int64_t T, int32_t N, float* out, void* ws, size_t ws_bytes, void* stream, VdkReduceJob* job);

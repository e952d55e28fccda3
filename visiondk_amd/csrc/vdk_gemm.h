// vdk_gemm.h — internal parameter block of the NT GEMM kernel (public descriptor: include/visiondk.h)
#pragma once
#include "visiondk.h"
#include "vdk_device.h"
#include "vdk_margin.h"

typedef VdkGemmDesc GemmDesc;

struct GemmParams {
  const bf16_t* A; const bf16_t* B; void* C;
  long lda, ldb, ldc;
  int M, N, K;
  int c_dtype;
  const float* bias;
  const float* residual; long ldr;
  int act;
  bf16_t* aux; long ldaux;
  float alpha;
  int row_group;             // > 0: token-row remap period (|VdkGemmDesc.row_group|)
  int row_shift;             // 1: one cls slot per group is skipped (desc.row_group > 0); 0: rows stay, only the residual row repeats per group (desc.row_group < 0)
  int a_row_group;
  int splitk; int k_per_split;
  float* slabs;              // split-K: [splitk][M][N] partial sums; stream-K: [2 * grid][256 * 256] raw accumulator slabs
  int* sk_cnt;               // stream-K: per-tile k-tile counters (zero between launches)
  float* ocs_part;           // 256-kernel, E_OCS: per (row tile, wave row) column sums of the stored bf16 output, f32 [2 * ceil(M/256)][N]
  float* colsum_part;        // 256-kernel, NT only: per (row tile, wave) column sums of A, f32 [ceil(M/256)*8][K] (bias gradient fused into the dgrad GEMM)
  // implicit-GEMM convolution (conv_on): A is an NHWC tensor gathered on the fly, see VdkConvGeom
  int conv_on, cCin, cH, cW, cOH, cOW, cKH, cKW, cstride, cpad, ctrans;
  int crows;                 // conv_on with TN (the weight gradient): valid contraction rows = batch * OH * OW; K is crows rounded up (VdkConvGeom.rows)
  // E_Q8: an fp8 copy of the stored bf16 output rides along (the A operand of the next fp8 GEMM: no separate quantisation pass over the tensor)
  unsigned char* q8; long ldq8; const float* q8_scale; float* q8_amax; int q8_fmt;
  MarginEpi me;              // E_MSTAT / E_MGRAD epilogues (margin-softmax head: cos tiles never leave the registers as fp32)
  unsigned long long* dbg;   // debug only: 4 cycle stamps per workgroup (start, operands landed, main loop done, end)
  int band_cw;               // gemm_w4_kernel / gemm_w4h_kernel: tile order = column bands of band_cw tile columns, row-major inside a band (0: plain row-major)
  int sk_xcd;                // gemm_w4_kernel, split-K: 1-D grid, a slab's tiles stay on one XCD (see the kernel); 0: grid (tiles, splits)
  int stagger, stagger_lo;   // gemm_w4h_kernel: workgroups stagger_lo .. 2 * stagger_lo - 1 start `stagger` shader cycles late (0: nobody)
  int opf;                   // operand format of A, B, a 16-bit C and aux: VDK_OPF_BF16 | VDK_OPF_F16 (VdkGemmDesc.ab_dtype)
  const float* cscale;       // VdkGemmDesc.col_scale: per-column factor of the accumulator, applied before bias / residual (fp32-output epilogues)
  const float* rscale; int rps;   // VdkGemmDesc.row_scale / rows_per_scale: C = residual + rscale[m / rps] * (acc + bias) (fp32-output epilogues with a residual)
};

// in-library launcher (no descriptor copy through the C ABI)
extern "C" int vdk_gemm_bf16_nt(const VdkGemmDesc* d, void* ws, size_t ws_bytes, void* stream);
extern "C" int vdk_transpose_bf16(const void* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, int32_t Rpad, int32_t in_row_group, float* colsum_partial, void* stream);
// in-library hint for the NEXT TN (trans = 1) problems of this thread: 1 = the 256x128 two-workgroup kernel where it serves, 0 = the dispatcher's own choice (four-wave).
// The Swin engine's weight gradients use it (tools/bench_gemm_swin.py: with half the splits the half-tile form wins for tiny and for very large outputs).
void vdk_gemm_tn_prefer_half(int on);
int vdk_transpose_16(const void* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, int32_t Rpad, int32_t in_row_group, float* colsum_partial, int opf, void* stream);

// gemm_w4.hip: the 4-wave (one wave per SIMD) 256x256 kernel.  serves(): the problem fits its 32-bit buffer offsets and asks for no by-product it lacks.
bool vdk_gemm_w4_serves(const GemmParams& p, bool trans);
bool vdk_gemm_w4_launch(const GemmParams& p, bool trans, int E, unsigned tiles, unsigned splitk, void* stream, void* ev0, void* ev1);
bool vdk_gemm_w4h_serves(const GemmParams& p, bool trans);
bool vdk_gemm_w4h_launch(const GemmParams& p, bool trans, int E, unsigned splitk, void* stream, void* ev0, void* ev1);
// the same kernels instantiated for fp16 operands (gemm_w4_f16.hip); GemmParams.opf selects between them in gemm.hip
bool vdk_gemm_w4_launch_f16(const GemmParams& p, bool trans, int E, unsigned tiles, unsigned splitk, void* stream, void* ev0, void* ev1);
bool vdk_gemm_w4h_launch_f16(const GemmParams& p, bool trans, int E, unsigned splitk, void* stream, void* ev0, void* ev1);
int vdk_gemm_w4_cus();      // workgroups of the persistent walk (CUs minus the reserve)

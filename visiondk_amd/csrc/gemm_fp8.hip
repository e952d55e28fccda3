// gemm_fp8.hip — C[M,N] = epilogue(s * A[M,K] . B[N,K]^T) with OCP fp8 operands on the block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, scales fixed to 2^0),
// the operand path of BASELINE.json configs[4] ("SigLIP ViT-L/14 336 ... fp8 MFMA").  The reference has no fp8 code: this is the Linear of timm's blocks
// (models/classifier/classify_model.py:49-54) under a per-tensor delayed-scaling recipe -- activations / weights in e4m3, gradients in e5m2, fp32 accumulation,
// the product of the two inverse scales applied to the accumulators before the usual fused epilogue.
//
// Same structure as the 256x256 bf16 kernel of gemm.hip, byte for byte: a K-tile is still 128 BYTES per row (128 fp8 elements instead of 64 bf16), so the LDS image,
// the LDS-DMA instructions (swizzle on the source address), the 4-interval read / MFMA schedule with counted vmcnt and the wave-row stagger are unchanged; a
// fragment is 32 bytes per lane (two ds_read_b128) and a segment issues 8 MFMAs of K = 64 (64 cycles each) where the bf16 kernel issues 16 of K = 16:
// the same matrix-pipe time per K-tile for twice the contraction length.  Any k order inside a lane's 32 bytes is fine: A and B fragments are read with the
// same addressing, and the MFMA pairs byte b of lane (row, half) with byte b of lane (col, half).
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_gemm.h"
#include "vdk_gemm_epilogue.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));

#define F_REGION 16384
#define F_TILEBUF (4 * F_REGION)
#define F_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))
#define F_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

struct Fp8Params {
  GemmParams g;
  const unsigned char* A; const unsigned char* B;   // fp8 operands, leading dimensions g.lda / g.ldb in bytes
  const float* a_scale_inv; const float* b_scale_inv;   // device scalars (delayed scaling): C = (1 / sa) (1 / sb) acc
};

// the 2 DMA instructions that fill one 16 KB region (128 rows x 128 bytes): LDS chunk c' of row r holds global chunk c' ^ ((r >> 1) & 7)
__device__ __forceinline__ void f_issue_region(unsigned char* region, const unsigned char* __restrict__ base, long ld, int row0, int nrows, int k0, bool is_b, int sub, int w,
                                               int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rr = j * 64 + w * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((rr >> 1) & 7);
    int trow;
    if (is_b) trow = (rr >> 5) * 64 + sub * 32 + (rr & 31);
    else trow = (rr >> 6) * 128 + sub * 64 + (rr & 63);
    int grow = row0 + trow;
    if (grow > nrows - 1) grow = nrows - 1;               // rows beyond the matrix are masked at the store
    const unsigned char* src = base + (long)grow * ld + k0 + c * 16;
    unsigned char* dst = region + (j * 512 + w * 64) * 16;
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(src), VDK_LDS_PTR(dst), 16, 0, 0);
  }
}
// fragment of 32 rows x one 64-deep k-step: lane (l31, hi) -> the 32 bytes at k = 64 ks + 32 hi
__device__ __forceinline__ i32x8 f_read_frag(const unsigned char* region, int rb, int ks, int l31, int hi) {
  const int rr = rb + l31;
  const int f = (rr >> 1) & 7;
  const u32x4 lo = *(const u32x4*)(region + rr * 128 + (((ks * 4 + hi * 2) ^ f) << 4));
  const u32x4 up = *(const u32x4*)(region + rr * 128 + (((ks * 4 + hi * 2 + 1) ^ f) << 4));
  return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)up[0], (int)up[1], (int)up[2], (int)up[3]};
}

#define F_MFMA(a, b, c) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), AF, BF, 0, 127, 0, 127)

// AF / BF: operand formats (0 = e4m3, 1 = e5m2)
template <int AF, int BF, int E>
__global__ __launch_bounds__(512) void gemm256_fp8_kernel(Fp8Params q) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * F_TILEBUF];   // 128 KB: operand buffers, reused by the epilogue
  GemmParams p = q.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3, hi = lane >> 5, l31 = lane & 31;
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  const int nwg = ntn * ntm;
  int bid = blockIdx.x;
  {
    int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
  }
  const int tn = bid % ntn, tm = bid / ntn;
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = p.K / 128;                               // launcher guarantees K % 128 == 0

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define F_REG(buf, id) (smem + (buf) * F_TILEBUF + (id) * F_REGION)   /* id: 0 RA0, 1 RA1, 2 RB0, 3 RB1 */
#define F_ISSUE_A(buf, sub, t) f_issue_region(F_REG(buf, sub), q.A, p.lda, m0, p.M, (t) * 128, false, sub, w, lane)
#define F_ISSUE_B(buf, sub, t) f_issue_region(F_REG(buf, 2 + (sub)), q.B, p.ldb, n0, p.N, (t) * 128, true, sub, w, lane)
#define F_FRAG_A(reg, rt, ks) f_read_frag(reg, wr * 64 + (rt) * 32, ks, l31, hi)
#define F_FRAG_B(reg, ks) f_read_frag(reg, wc * 32, ks, l31, hi)

  // prologue: tiles 0 and 1 completely
  if (nk > 0) { F_ISSUE_A(0, 0, 0); F_ISSUE_B(0, 0, 0); F_ISSUE_B(0, 1, 0); F_ISSUE_A(0, 1, 0); }
  if (nk > 1) { F_ISSUE_A(1, 0, 1); F_ISSUE_B(1, 0, 1); F_ISSUE_B(1, 1, 1); F_ISSUE_A(1, 1, 1); F_WAIT_VM(8); } else { F_WAIT_VM(0); }
  F_BAR();
  if (wr == 1) F_BAR();                                   // stagger: the second wave-row runs one interval behind

  // schedule of gemm.hip's 256x256 kernel (see there for the WAR / RAW argument): RA reads A0, B0, B1; MA multiplies A0 x (B0, B1) and issues RA1 of tile t+1;
  // RB reads A1; MB multiplies A1 x (B1, B0) and issues RA0, RB0, RB1 of tile t+2.  vmcnt is never 0 in steady state.
  i32x8 a0[2][2], a1[2][2], b0[2], b1[2];
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const unsigned char* RA0 = F_REG(cur, 0); const unsigned char* RA1 = F_REG(cur, 1);
    const unsigned char* RB0 = F_REG(cur, 2); const unsigned char* RB1 = F_REG(cur, 3);
    // ---- RA
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b0[ks] = F_FRAG_B(RB0, ks);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a0[rt][ks] = F_FRAG_A(RA0, rt, ks);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b1[ks] = F_FRAG_B(RB1, ks);
    if (t + 1 < nk) { F_WAIT_VM(6); } else { F_WAIT_VM(0); }
    F_BAR();
    // ---- MA
    VDK_PIN2(acc[0][0], acc[1][0]); VDK_PIN2(acc[0][1], acc[1][1]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      acc[0][0] = F_MFMA(a0[0][ks], b0[ks], acc[0][0]);
      acc[1][0] = F_MFMA(a0[1][ks], b0[ks], acc[1][0]);
      if (ks == 0 && t >= 1 && t + 1 < nk) {                  // (tile 1's RA1 was issued by the prologue)
        __builtin_amdgcn_sched_barrier(0); F_ISSUE_A(cur ^ 1, 1, t + 1); __builtin_amdgcn_sched_barrier(0);
      }
      acc[0][1] = F_MFMA(a0[0][ks], b1[ks], acc[0][1]);
      acc[1][1] = F_MFMA(a0[1][ks], b1[ks], acc[1][1]);
    }
    VDK_PIN2(acc[0][0], acc[1][0]); VDK_PIN2(acc[0][1], acc[1][1]);
    __builtin_amdgcn_s_setprio(0);
    F_BAR();
    // ---- RB
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a1[rt][ks] = F_FRAG_A(RA1, rt, ks);
    if (t + 1 < nk) { F_WAIT_VM(2); } else { F_WAIT_VM(0); }
    F_BAR();
    // ---- MB
    VDK_PIN2(acc[2][1], acc[3][1]); VDK_PIN2(acc[2][0], acc[3][0]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      acc[2][1] = F_MFMA(a1[0][ks], b1[ks], acc[2][1]);
      if (t + 2 < nk) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) F_ISSUE_A(cur, 0, t + 2); else F_ISSUE_B(cur, 1, t + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[3][1] = F_MFMA(a1[1][ks], b1[ks], acc[3][1]);
      acc[2][0] = F_MFMA(a1[0][ks], b0[ks], acc[2][0]);
      if (t + 2 < nk && ks == 0) {
        __builtin_amdgcn_sched_barrier(0); F_ISSUE_B(cur, 0, t + 2); __builtin_amdgcn_sched_barrier(0);
      }
      acc[3][0] = F_MFMA(a1[1][ks], b0[ks], acc[3][0]);
    }
    VDK_PIN2(acc[2][1], acc[3][1]); VDK_PIN2(acc[2][0], acc[3][0]);
    __builtin_amdgcn_s_setprio(0);
    F_BAR();
  }
  if (wr == 0) F_BAR();                                   // match the barrier count of the lagging wave-row

  // ---- epilogue: dequantise, then the shared fused epilogue (wave-private 16 KB slab, 64 rows x 64 fp32 at a time)
  p.alpha = p.alpha * (q.a_scale_inv ? q.a_scale_inv[0] : 1.0f) * (q.b_scale_inv ? q.b_scale_inv[0] : 1.0f);
  float* slab = (float*)(smem + w * 16384);
  const int ncol = n0 + wc * 64 + (lane & 7) * 8;
  float bias8[8], ocs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q8am = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if ((E & E_BIAS) && ncol < p.N) {
    f32x4 b0v = *(const f32x4*)(p.bias + ncol), b1v = *(const f32x4*)(p.bias + ncol + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias8[e] = b0v[e]; bias8[4 + e] = b1v[e]; }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + ct * 32 + l31] = acc[half * 2 + rt][ct][r] * p.alpha;
    __builtin_amdgcn_wave_barrier();
    h_epilogue_half<E>(p, slab, lane, (long)m0 + wr * 128 + half * 64, ncol, 0, bias8, ocs8, q8am);
    __builtin_amdgcn_wave_barrier();
  }
  if (E & E_OCS) {   // column sums of the stored bf16 output (c_colsum, as in gemm.hip): lanes with the same (lane & 7) hold the same 8 columns; one partial row per (row tile, wave row)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = ocs8[e];
      v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      ocs8[e] = v;
    }
    if (lane < 8 && ncol < p.N) {
      float* dst = p.ocs_part + ((long)(m0 >> 8) * 2 + wr) * p.N + ncol;
      *(f32x4*)dst = (f32x4){ocs8[0], ocs8[1], ocs8[2], ocs8[3]};
      *(f32x4*)(dst + 4) = (f32x4){ocs8[4], ocs8[5], ocs8[6], ocs8[7]};
    }
  }
  if ((E & E_Q8) && p.q8_amax) {
    // amax of the by-product: ONE atomic per workgroup, and only when it can raise the value (an atomic per wave and half was measured: 73 728 same-address atomics per
    // GEMM = +340 us on a 630 us kernel; the fp8 stores themselves are free).  The plain read may be stale, but the value only grows: skipping on "not larger" is safe.
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) q8am = fmaxf(q8am, __shfl_xor(q8am, off));
    __syncthreads();                                      // every wave is done with its slab (the reduction slots alias the first one)
    float* red = (float*)smem;
    if (lane == 0) red[w] = q8am;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = red[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i]);
      if (m > *(volatile float*)p.q8_amax) atomicMax((unsigned*)p.q8_amax, __float_as_uint(m));      // non-negative floats order like their bit patterns
    }
  }
#undef F_REG
#undef F_ISSUE_A
#undef F_ISSUE_B
#undef F_FRAG_A
#undef F_FRAG_B
}

// ---- quantisation: out = fp8(clamp(x * scale)), amax = max(amax, max |x|)  (delayed scaling: `scale` was derived from an earlier step's amax) -----------
// FMT 0: e4m3 (max 448), 1: e5m2 (max 57344).  x bf16 or f32, 16 elements per thread.
__device__ __forceinline__ unsigned f_pack4(float a, float b, float c, float d, int fmt) {
  int v = 0;
  if (fmt == 0) { v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false); v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true); }
  else { v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false); v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true); }
  return (unsigned)v;
}
template <bool BF>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const void* __restrict__ x, long n16, const float* __restrict__ scale, unsigned char* __restrict__ out, int fmt,
                                                        float* __restrict__ amax) {
  __shared__ float red[4];
  const float s = scale ? scale[0] : 1.0f;
  const float lim = fmt == 0 ? 448.0f : 57344.0f;
  float am = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    float v[16];
    if (BF) {
      const u32x4 u0 = *(const u32x4*)((const bf16_t*)x + i * 16), u1 = *(const u32x4*)((const bf16_t*)x + i * 16 + 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(u0[e]); v[2 * e + 1] = bf_hi(u0[e]); v[8 + 2 * e] = bf_lo(u1[e]); v[9 + 2 * e] = bf_hi(u1[e]); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const f32x4 t = *(const f32x4*)((const float*)x + i * 16 + 4 * j); v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3]; }
    }
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float c[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { am = fmaxf(am, fabsf(v[4 * j + e])); c[e] = fminf(fmaxf(v[4 * j + e] * s, -lim), lim); }
      o[j] = f_pack4(c[0], c[1], c[2], c[3], fmt);
    }
    if (out) *(u32x4*)(out + i * 16) = (u32x4){o[0], o[1], o[2], o[3]};      // out == NULL: amax pass only (current scaling: amax, scale update, then the real pass)
  }
  if (amax) {
    am = block_max<4>(am, red);
    if (threadIdx.x == 0) atomicMax((unsigned*)amax, __float_as_uint(am));     // non-negative floats order like their bit patterns
  }
}
// scale = fmt_max / max(amax, tiny) (a power-of-two-free per-tensor scale), scale_inv = 1 / scale; amax is reset for the next accumulation window
__global__ void fp8_scale_update_kernel(float* __restrict__ amax, float* __restrict__ scale, float* __restrict__ scale_inv, int n, float fmt_max, float margin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = amax[i];
  if (a > 0.f && a < 3.0e38f) { const float s = fmt_max / (a * margin); scale[i] = s; scale_inv[i] = 1.0f / s; }   // (non-finite amax: keep the previous scale, see vit_fp8_update_kernel)
  amax[i] = 0.f;
}

extern "C" {

// x (bf16 | f32, n elements, n % 16 == 0) -> fp8 (fmt 0 = e4m3, 1 = e5m2) with the device scalar `scale` (NULL = 1); amax (device scalar, may be NULL) accumulates max |x|
int vdk_quant_fp8(const void* x, int32_t x_dtype, int64_t n, const float* scale, void* out_fp8, int32_t fmt, float* amax, void* stream) {
  if (!x || (!out_fp8 && !amax) || n < 0 || (n & 15) || (fmt != 0 && fmt != 1) || (x_dtype != VDK_BF16 && x_dtype != VDK_F32)) return vdk_fail(VDK_EINVAL, "vdk_quant_fp8: bad argument (n % 16 == 0)");
  if (n == 0) return VDK_OK;
  const long n16 = n / 16;
  long blocks = (n16 + 255) / 256; if (blocks > 4096) blocks = 4096;
  if (x_dtype == VDK_BF16) hipLaunchKernelGGL(quant_fp8_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n16, scale, (unsigned char*)out_fp8, (int)fmt, amax);
  else hipLaunchKernelGGL(quant_fp8_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n16, scale, (unsigned char*)out_fp8, (int)fmt, amax);
  return vdk_check_launch("vdk_quant_fp8");
}
// delayed scaling: for each of n tensors scale = fmt_max / (margin * amax) (kept if amax == 0), scale_inv = 1 / scale, amax reset to 0
int vdk_fp8_scale_update(float* amax, float* scale, float* scale_inv, int32_t n, int32_t fmt, float margin, void* stream) {
  if (!amax || !scale || !scale_inv || n <= 0 || (fmt != 0 && fmt != 1) || !(margin >= 1.0f)) return vdk_fail(VDK_EINVAL, "vdk_fp8_scale_update: bad argument");
  hipLaunchKernelGGL(fp8_scale_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, amax, scale, scale_inv, (int)n, fmt == 0 ? 448.0f : 57344.0f, margin);
  return vdk_check_launch("vdk_fp8_scale_update");
}

// C = epilogue((1/sa)(1/sb) A . B^T): A [M,K], B [N,K] fp8 (a_fmt / b_fmt: 0 = e4m3, 1 = e5m2; combinations (0,0) and (1,0)), lda / ldb in elements (= bytes), % 16 == 0;
// M % 256 == 0 is not required (rows are masked), N % 8 == 0, K % 128 == 0, M, N >= 256.  Epilogue fields of the descriptor as for vdk_gemm_bf16_nt (bias, GELU + aux,
// DGELU, residual, c_dtype); splitk / trans / conv / row_group / a_colsum are not served (VDK_EUNSUPPORTED): the bf16 kernels keep those.
int vdk_gemm_fp8_nt(const VdkGemmDesc* d, int32_t a_fmt, int32_t b_fmt, const float* a_scale_inv, const float* b_scale_inv, void* stream_) {
  return vdk_gemm_fp8_nt_q8(d, a_fmt, b_fmt, a_scale_inv, b_scale_inv, nullptr, 0, 0, nullptr, nullptr, stream_);
}
// + an fp8 copy of the bf16 output (out8 [M, ldo8] bytes, out_fmt, device scalar out_scale, amax accumulated into out_amax): what vdk_quant_fp8 would make of C, written by
// the epilogue that stores C -- for the GELU and dGELU forms (bf16 C, N % 64 == 0), whose outputs are the A operands of the next fp8 GEMM
int vdk_gemm_fp8_nt_q8(const VdkGemmDesc* d, int32_t a_fmt, int32_t b_fmt, const float* a_scale_inv, const float* b_scale_inv, void* out8, int64_t ldo8, int32_t out_fmt,
                       const float* out_scale, float* out_amax, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !d->A || !d->B || !d->C) return vdk_fail(VDK_EINVAL, "vdk_gemm_fp8_nt: null pointer");
  if (d->M < 256 || d->N < 256 || d->K <= 0 || (d->K % 128) || (d->N & 7) || (d->lda & 15) || (d->ldb & 15) || (d->ldc & 7))
    return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_fp8_nt: needs M, N >= 256, K % 128 == 0, N % 8 == 0, lda / ldb % 16 == 0");
  if (d->splitk > 1 || d->trans || d->conv || d->row_group != 0 || d->a_colsum || (d->c_colsum && d->act != VDK_ACT_DGELU))
    return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_fp8_nt: split-K / TN / conv / row remap stay on the bf16 kernels (c_colsum: with the dGELU epilogue only)");
  if (!((a_fmt == 0 || a_fmt == 1) && b_fmt == 0)) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_fp8_nt: formats (e4m3, e4m3) and (e5m2, e4m3)");
  if (d->act < VDK_ACT_NONE || d->act > VDK_ACT_DGELU) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_fp8_nt: act NONE, GELU or DGELU");
  if ((d->act == VDK_ACT_DGELU && !d->aux) || (d->aux && (d->ldaux & 7))) return vdk_fail(VDK_EINVAL, "vdk_gemm_fp8_nt: bad aux");
  Fp8Params q;
  GemmParams& p = q.g;
  p.A = nullptr; p.B = nullptr; p.C = d->C; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.M = d->M; p.N = d->N; p.K = d->K; p.c_dtype = d->c_dtype;
  p.bias = (const float*)d->bias; p.residual = (const float*)d->residual; p.ldr = d->ldr; p.act = d->act; p.aux = (bf16_t*)d->aux; p.ldaux = d->ldaux;
  p.alpha = d->alpha; p.row_group = 0; p.row_shift = 0; p.a_row_group = 0; p.splitk = 1; p.k_per_split = d->K; p.slabs = nullptr; p.colsum_part = nullptr; p.ocs_part = (float*)d->c_colsum; p.conv_on = 0; p.dbg = nullptr; p.cscale = nullptr; p.rscale = nullptr; p.rps = 1; p.opf = 0;
  q.A = (const unsigned char*)d->A; q.B = (const unsigned char*)d->B; q.a_scale_inv = a_scale_inv; q.b_scale_inv = b_scale_inv;
  const bool bias = d->bias != nullptr, res = d->residual != nullptr, f32 = d->c_dtype == VDK_F32;
  int E = -1;
  if (!res && d->act == VDK_ACT_NONE && !f32) E = bias ? E_BIAS : 0;
  else if (!res && d->act == VDK_ACT_GELU && !f32 && bias && d->aux) E = E_BIAS | E_GELU;
  else if (!res && d->act == VDK_ACT_DGELU && !f32 && !bias) E = E_DGELU;
  else if (res && d->act == VDK_ACT_NONE && f32 && bias) E = E_BIAS | E_RES | E_F32;
  else if (!res && d->act == VDK_ACT_NONE && f32 && !bias) E = E_F32;
  if (E < 0) return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_fp8_nt: epilogue combination not instantiated");
  p.q8 = nullptr; p.ldq8 = 0; p.q8_scale = nullptr; p.q8_amax = nullptr; p.q8_fmt = 0;
  if (out8) {
    if ((E != (E_BIAS | E_GELU) && E != E_DGELU) || (d->N & 63) || (ldo8 & 7) || ldo8 < d->N || (out_fmt != 0 && out_fmt != 1))
      return vdk_fail(VDK_EUNSUPPORTED, "vdk_gemm_fp8_nt_q8: the fp8 copy rides with the GELU / dGELU epilogues only (bf16 C, N % 64 == 0, ldo8 % 8 == 0)");
    p.q8 = (unsigned char*)out8; p.ldq8 = ldo8; p.q8_scale = out_scale; p.q8_amax = out_amax; p.q8_fmt = out_fmt;
    E |= E_Q8;
  }
  if (d->c_colsum) E |= E_OCS;      // [2 * ceil(M / 256)][N] partial rows (vdk_gemm_c_colsum_rows), dGELU form only (checked above)
  const dim3 grid((unsigned)(((d->M + 255) / 256) * ((d->N + 255) / 256)));
#define L8(AFv, EE) hipLaunchKernelGGL((gemm256_fp8_kernel<AFv, 0, EE>), grid, dim3(512), 0, stream, q)
#define L8E(AFv)                                                   \
  switch (E) {                                                    \
    case 0: L8(AFv, 0); break;                                    \
    case E_BIAS: L8(AFv, E_BIAS); break;                          \
    case E_BIAS | E_GELU: L8(AFv, E_BIAS | E_GELU); break;        \
    case E_DGELU: L8(AFv, E_DGELU); break;                        \
    case E_BIAS | E_GELU | E_Q8: L8(AFv, E_BIAS | E_GELU | E_Q8); break; \
    case E_DGELU | E_Q8: L8(AFv, E_DGELU | E_Q8); break;          \
    case E_DGELU | E_OCS: L8(AFv, E_DGELU | E_OCS); break;        \
    case E_DGELU | E_OCS | E_Q8: L8(AFv, E_DGELU | E_OCS | E_Q8); break; \
    case E_BIAS | E_RES | E_F32: L8(AFv, E_BIAS | E_RES | E_F32); break; \
    default: L8(AFv, E_F32); break;                               \
  }
  if (a_fmt == 0) { L8E(0) } else { L8E(1) }
#undef L8E
#undef L8
  return vdk_check_launch("vdk_gemm_fp8_nt");
}

}  // extern "C"

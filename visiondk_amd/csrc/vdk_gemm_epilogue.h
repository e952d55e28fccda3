// vdk_gemm_epilogue.h — the fused epilogues shared by the bf16 GEMM kernels (gemm.hip) and the fp8 GEMM kernel (gemm_fp8.hip): bias, exact-erf GELU with the saved
// pre-activation, dGELU, fp32 residual, fp32 / bf16 output, split-K slabs, token-row remap.
#pragma once
#include "vdk_gemm.h"

// ---- shared fused epilogue for one 8-wide row chunk: v[8] = raw accumulators of C[mi][n..n+7] -----------------------
__device__ __forceinline__ void g_epilogue_store8(const GemmParams& p, long mi, int n, float (&v)[8], int z) {
    // token-row remap (patch embedding -> token buffer): output row skips one cls slot per group and the
    // residual (pos_embed) row repeats per group
    const long m = p.row_group > 0 ? mi + (mi / p.row_group + 1) * p.row_shift : mi;
    const long mr = p.row_group > 0 ? mi % p.row_group + p.row_shift : mi;
    if (p.splitk > 1) {  // raw fp32 partials; the reduce kernel finishes the job
      float* dst = p.slabs + ((long)z * p.M + mi) * p.N + n;
      *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      return;
    }
    if (p.alpha != 1.0f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
    }
    if (p.bias) {
      f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (p.act == VDK_ACT_GELU) {
      if (p.aux) {  // keep the pre-activation for the backward pass
        u32x4 u = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *(u32x4*)(p.aux + m * p.ldaux + n) = u;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
    } else if (p.act == VDK_ACT_DGELU) {  // dL/du = dL/dg * gelu'(u), u = saved pre-activation
      u32x4 u = *(const u32x4*)(p.aux + m * p.ldaux + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] *= gelu_grad_f(bf_lo(u[e]));
        v[2 * e + 1] *= gelu_grad_f(bf_hi(u[e]));
      }
    }
    if (p.residual) {
      const float* rs = p.residual + mr * p.ldr + n;
      f32x4 r0 = *(const f32x4*)rs, r1 = *(const f32x4*)(rs + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
    }
    if (p.c_dtype == VDK_F32) {
      float* dst = (float*)p.C + m * p.ldc + n;
      *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
      u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      *(u32x4*)((bf16_t*)p.C + m * p.ldc + n) = o;
    }
}


// ---- compile-time epilogue variants of the 256x256 kernel ------------------------------------------------------------------
// E_* flags select what the epilogue does; E_GENERIC falls back to the run-time-flag path (g_epilogue_store8).  The specialised
// paths have no branches, load the bias once per lane and issue every residual / pre-activation load of a 64-row half before
// the first use, so their latencies overlap instead of serialising per pass.
#define E_BIAS 1
#define E_GELU 2      /* exact GELU, pre-activation saved to aux */
#define E_DGELU 4     /* multiply by GELU'(aux) */
#define E_RES 8       /* + fp32 residual */
#define E_F32 16      /* fp32 output (else bf16) */
#define E_SPLITK 32   /* raw fp32 partial slab */
#define E_ROWGRP 64   /* token-row remap (patch embedding) */
#define E_OCS 128     /* by-product: column sums of the stored (bf16-rounded) output over this wave's 128 rows -> p.ocs_part (bias gradient of the next Linear back) */
#define E_GENERIC 0x1000

template <int E>
__device__ __forceinline__ void h_epilogue_half(const GemmParams& p, const float* slab, int lane, long mbase /* first row of this 64-row half */,
                                                int n, int z, const float (&bias8)[8], float (&ocs)[8]) {
  // this lane: rows mbase + pass*8 + (lane >> 3), pass = 0..7, columns n .. n+7
  const int rsub = lane >> 3, cc = (lane & 7) * 8;
  if (n >= p.N) return;
  long mo[8], mr[8];
  bool ok[8];
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    const long mi = mbase + ps * 8 + rsub;
    ok[ps] = mi < p.M;
    mo[ps] = (E & E_ROWGRP) ? mi + (mi / p.row_group + 1) * p.row_shift : mi;
    mr[ps] = (E & E_ROWGRP) ? mi % p.row_group + p.row_shift : mi;
  }
  f32x4 r0[8], r1[8];
  u32x4 ux[8];
  if (E & E_RES) {
#pragma unroll
    for (int ps = 0; ps < 8; ++ps)
      if (ok[ps]) { const float* rs = p.residual + mr[ps] * p.ldr + n; r0[ps] = *(const f32x4*)rs; r1[ps] = *(const f32x4*)(rs + 4); }
  }
  if (E & E_DGELU) {
#pragma unroll
    for (int ps = 0; ps < 8; ++ps)
      if (ok[ps]) ux[ps] = *(const u32x4*)(p.aux + mo[ps] * p.ldaux + n);
  }
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    if (!ok[ps]) continue;
    const int row = ps * 8 + rsub;
    float v[8];
    f32x4 x0 = *(const f32x4*)(slab + row * 64 + cc), x1 = *(const f32x4*)(slab + row * 64 + cc + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
    if (E & E_SPLITK) {
      float* dst = p.slabs + ((long)z * p.M + (mbase + row)) * p.N + n;
      *(f32x4*)dst = x0; *(f32x4*)(dst + 4) = x1;
      continue;
    }
    if (E & E_BIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias8[e];
    }
    if (E & E_GELU) {
      *(u32x4*)(p.aux + mo[ps] * p.ldaux + n) = (u32x4){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
    }
    if (E & E_DGELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] *= gelu_grad_f(bf_lo(ux[ps][e])); v[2 * e + 1] *= gelu_grad_f(bf_hi(ux[ps][e])); }
    }
    if (E & E_RES) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += r0[ps][e]; v[4 + e] += r1[ps][e]; }
    }
    if (E & E_F32) {
      float* dst = (float*)p.C + mo[ps] * p.ldc + n;
      *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
      const u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      *(u32x4*)((bf16_t*)p.C + mo[ps] * p.ldc + n) = o;
      if (E & E_OCS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ocs[2 * e] += bf_lo(o[e]); ocs[2 * e + 1] += bf_hi(o[e]); }
      }
    }
  }
}


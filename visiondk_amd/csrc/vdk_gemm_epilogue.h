// vdk_gemm_epilogue.h — the fused epilogues shared by the bf16 GEMM kernels (gemm.hip) and the fp8 GEMM kernel (gemm_fp8.hip): bias, exact-erf GELU with the saved
// pre-activation, dGELU, fp32 residual, fp32 / bf16 output, split-K slabs, token-row remap.
#pragma once
#include "vdk_gemm.h"

// ---- shared fused epilogue for one 8-wide row chunk: v[8] = raw accumulators of C[mi][n..n+7] -----------------------
template <int OF = 0>
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
    if (p.cscale) {
      f32x4 c0 = *(const f32x4*)(p.cscale + n), c1 = *(const f32x4*)(p.cscale + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] *= c0[e]; v[4 + e] *= c1[e]; }
    }
    if (p.bias) {
      f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (p.act == VDK_ACT_GELU) {
      if (p.aux) {  // keep the pre-activation for the backward pass
        u32x4 u = {pack_op2<OF>(v[0], v[1]), pack_op2<OF>(v[2], v[3]), pack_op2<OF>(v[4], v[5]), pack_op2<OF>(v[6], v[7])};
        *(u32x4*)(p.aux + m * p.ldaux + n) = u;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { const vdk_f32x2 g2 = gelu_f2((vdk_f32x2){v[2 * e], v[2 * e + 1]}); v[2 * e] = g2[0]; v[2 * e + 1] = g2[1]; }
    } else if (p.act == VDK_ACT_DGELU) {  // dL/du = dL/dg * gelu'(u), u = saved pre-activation
      u32x4 u = *(const u32x4*)(p.aux + m * p.ldaux + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const vdk_f32x2 d2 = gelu_grad_f2((vdk_f32x2){op_lo<OF>(u[e]), op_hi<OF>(u[e])});
        v[2 * e] *= d2[0];
        v[2 * e + 1] *= d2[1];
      }
    } else if (p.act == VDK_ACT_GELU_SAVE_GRAD) {  // GELU, its derivative saved for the backward pass
      float d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gelu_both_f(v[e], v[e], d[e]);
      *(u32x4*)(p.aux + m * p.ldaux + n) = (u32x4){pack_h2(d[0], d[1]), pack_h2(d[2], d[3]), pack_h2(d[4], d[5]), pack_h2(d[6], d[7])};
    } else if (p.act == VDK_ACT_MUL_AUX) {  // dL/du = dL/dg * saved gelu'(u) (fp16)
      u32x4 u = *(const u32x4*)(p.aux + m * p.ldaux + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] *= h_lo(u[e]); v[2 * e + 1] *= h_hi(u[e]); }
    }
    if (p.rscale) {      // VdkGemmDesc.row_scale: the branch (acc + bias) times its sample's factor, then the shortcut
      const float f = p.rscale[mi / p.rps];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= f;
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
      u32x4 o = {pack_op2<OF>(v[0], v[1]), pack_op2<OF>(v[2], v[3]), pack_op2<OF>(v[4], v[5]), pack_op2<OF>(v[6], v[7])};
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
#define E_MSTAT 256   /* margin head, pass 1: per-(row, 64-column slice) online-softmax partials of the margin logits; nothing is stored to C */
#define E_MGRAD 512   /* margin head, pass 2: C = bf16 d(loss)/d(cos) from the row statistics */
#define E_Q8 1024     /* by-product: fp8(clamp(stored bf16 value * q8_scale)) -> p.q8, max |value| -> p.q8_amax: bit-identical to vdk_quant_fp8 over the stored tensor */
#define E_AUXD 2048    /* with E_GELU: aux receives fp16 GELU'(pre-activation) instead of the bf16 pre-activation; with E_DGELU: aux holds that derivative (one multiplication) */
#define E_GENERIC 0x1000

// SWZ: the slab's 16-byte chunk c of row r lives at chunk position c ^ (r & 15) (bank-conflict-free for the row-per-lane writes of gemm_w4.hip)
template <int E, int NPS = 8 /* row passes: the slab holds NPS * 8 rows x 64 fp32 */, bool SWZ = false, int OF = 0 /* operand format of the 16-bit output / aux */>
__device__ __forceinline__ void h_epilogue_half(const GemmParams& p, const float* slab, int lane, long mbase /* first row of this slab */,
                                                int n, int z, const float (&bias8)[8], float (&ocs)[8], float& q8am) {
  // this lane: rows mbase + pass*8 + (lane >> 3), pass = 0..NPS-1, columns n .. n+7
  const int rsub = lane >> 3, cc = (lane & 7) * 8;
  if (E & E_MSTAT) {     // margin logits of this lane's 8 columns -> (max, sum exp, sum) merged over the 8 lanes that share a row -> one partial per (row, 64-column slice)
    const MarginEpi& me = p.me;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const long mi = mbase + ps * 8 + rsub;
      const bool rok = mi < me.B;
      const int row = ps * 8 + rsub, sw = SWZ ? (row & 15) : 0;
      f32x4 x0 = *(const f32x4*)(slab + row * 64 + (((cc >> 2) ^ sw) << 2)), x1 = *(const f32x4*)(slab + row * 64 + ((((cc >> 2) + 1) ^ sw) << 2));
      float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      MarginP P = me.P;
      long yt = -1; RowCtx R; R.thr = 0.f; R.final_gt = 0.f; R.dfinal = 1.f;
      if (rok) { margin_row_params(P, (int)mi); yt = me.labels[mi]; R = margin_row_ctx(P, me.gt ? me.gt[mi] : 0.f); }
      float m = -3.0e38f, sl = 0.f, lg[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lg[e] = -3.0e38f;
        if (rok && n + e < me.C) {
          float jc; margin_eval(P, R, v[e], (long)(n + e) == yt, lg[e], jc);
          sl += lg[e]; m = fmaxf(m, lg[e]);
          if ((long)(n + e) == yt) me.tlogit[mi] = lg[e];
        }
      }
      float se = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) se += (lg[e] > -1.0e38f) ? vdk_vexp(lg[e] - m) : 0.f;
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        const float m2 = __shfl_xor(m, off), s2 = __shfl_xor(se, off);
        sl += __shfl_xor(sl, off);
        const float M = fmaxf(m, m2);
        se = se * vdk_vexp(m - M) + s2 * vdk_vexp(m2 - M);      // (both factors are exp(0) = 1 or exp(-huge) = 0 when a side is empty: m = m2 = -3e38 gives 1 * 0 + 1 * 0)
        m = M;
      }
      if (rok && (lane & 7) == 0 && n < p.N) *(f32x4*)(me.stats + ((long)mi * me.nslice + (n >> 6)) * 4) = (f32x4){m, se, sl, 0.f};
    }
    return;
  }
  if (n >= p.N) return;
  long mo[NPS], mr[NPS];
  bool ok[NPS];
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const long mi = mbase + ps * 8 + rsub;
    ok[ps] = mi < p.M;
    mo[ps] = (E & E_ROWGRP) ? mi + (mi / p.row_group + 1) * p.row_shift : mi;
    mr[ps] = (E & E_ROWGRP) ? mi % p.row_group + p.row_shift : mi;
  }
  f32x4 r0[NPS], r1[NPS];
  u32x4 ux[NPS];
  float q8s = 1.0f, q8lim = 448.0f;      // (q8am: this lane's running max |stored value|, reduced by the kernel once per workgroup)
  if (E & E_Q8) { q8s = p.q8_scale ? p.q8_scale[0] : 1.0f; q8lim = p.q8_fmt == 0 ? 448.0f : 57344.0f; }
  if (E & E_RES) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
      if (ok[ps]) { const float* rs = p.residual + mr[ps] * p.ldr + n; r0[ps] = *(const f32x4*)rs; r1[ps] = *(const f32x4*)(rs + 4); }
  }
  if (E & E_DGELU) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
      if (ok[ps]) ux[ps] = *(const u32x4*)(p.aux + mo[ps] * p.ldaux + n);
  }
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    if (!ok[ps]) continue;
    const int row = ps * 8 + rsub, sw = SWZ ? (row & 15) : 0;
    float v[8];
    f32x4 x0 = *(const f32x4*)(slab + row * 64 + (((cc >> 2) ^ sw) << 2)), x1 = *(const f32x4*)(slab + row * 64 + ((((cc >> 2) + 1) ^ sw) << 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
    if (E & E_SPLITK) {
      float* dst = p.slabs + ((long)z * p.M + (mbase + row)) * p.N + n;
      *(f32x4*)dst = x0; *(f32x4*)(dst + 4) = x1;
      continue;
    }
    if ((E & E_F32) && p.cscale) {      // VdkGemmDesc.col_scale (fp32-output forms): acc * col_scale, then bias / residual
      const f32x4 c0 = *(const f32x4*)(p.cscale + n), c1 = *(const f32x4*)(p.cscale + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] *= c0[e]; v[4 + e] *= c1[e]; }
    }
    if (E & E_BIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias8[e];
    }
    if ((E & E_GELU) && (E & E_AUXD)) {
      float dv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gelu_both_f(v[e], v[e], dv[e]);
      *(u32x4*)(p.aux + mo[ps] * p.ldaux + n) = (u32x4){pack_h2(dv[0], dv[1]), pack_h2(dv[2], dv[3]), pack_h2(dv[4], dv[5]), pack_h2(dv[6], dv[7])};
    } else if (E & E_GELU) {
      *(u32x4*)(p.aux + mo[ps] * p.ldaux + n) = (u32x4){pack_op2<OF>(v[0], v[1]), pack_op2<OF>(v[2], v[3]), pack_op2<OF>(v[4], v[5]), pack_op2<OF>(v[6], v[7])};
#pragma unroll
      for (int e = 0; e < 4; ++e) { const vdk_f32x2 g2 = gelu_f2((vdk_f32x2){v[2 * e], v[2 * e + 1]}); v[2 * e] = g2[0]; v[2 * e + 1] = g2[1]; }
    }
    if ((E & E_DGELU) && (E & E_AUXD)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] *= h_lo(ux[ps][e]); v[2 * e + 1] *= h_hi(ux[ps][e]); }
    } else if (E & E_DGELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const vdk_f32x2 d2 = gelu_grad_f2((vdk_f32x2){op_lo<OF>(ux[ps][e]), op_hi<OF>(ux[ps][e])}); v[2 * e] *= d2[0]; v[2 * e + 1] *= d2[1]; }
    }
    if (E & E_RES) {
      if (p.rscale) {
        const float f = p.rscale[(mbase + row) / p.rps];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += r0[ps][e]; v[4 + e] += r1[ps][e]; }
    }
    if (E & E_MGRAD) {     // v = cos -> gscale * (softmax - eps / C - (1 - eps) [c is the target]) * d(logit)/d(cos); rows >= B and columns >= C are zero
      const MarginEpi& me = p.me;
      const long mi = mbase + row;
      if (mi < me.B) {
        MarginP P = me.P; margin_row_params(P, (int)mi);
        const long yt = me.labels[mi];
        const RowCtx R = margin_row_ctx(P, me.gt ? me.gt[mi] : 0.f);
        const float M = me.rowstat[2 * mi], inv = me.rowstat[2 * mi + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float lgv, jc; margin_eval(P, R, v[e], (long)(n + e) == yt, lgv, jc);
          float gd = vdk_vexp(lgv - M) * inv - me.epsc;
          if ((long)(n + e) == yt) gd -= (1.0f - me.smoothing);
          v[e] = (n + e < me.C) ? gd * me.gscale * jc : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
    }
    if (E & E_F32) {
      float* dst = (float*)p.C + mo[ps] * p.ldc + n;
      *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
      const u32x4 o = {pack_op2<OF>(v[0], v[1]), pack_op2<OF>(v[2], v[3]), pack_op2<OF>(v[4], v[5]), pack_op2<OF>(v[6], v[7])};
      *(u32x4*)((bf16_t*)p.C + mo[ps] * p.ldc + n) = o;
      if (E & E_OCS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ocs[2 * e] += op_lo<OF>(o[e]); ocs[2 * e + 1] += op_hi<OF>(o[e]); }
      }
      if (E & E_Q8) {
        float c[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = op_lo<OF>(o[e]), b = op_hi<OF>(o[e]);
          q8am = fmaxf(q8am, fmaxf(fabsf(a), fabsf(b)));
          c[2 * e] = fminf(fmaxf(a * q8s, -q8lim), q8lim); c[2 * e + 1] = fminf(fmaxf(b * q8s, -q8lim), q8lim);
        }
        int lo = 0, up = 0;
        if (p.q8_fmt == 0) {
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
          up = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], up, false); up = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], up, true);
        } else {
          lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], lo, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], lo, true);
          up = __builtin_amdgcn_cvt_pk_bf8_f32(c[4], c[5], up, false); up = __builtin_amdgcn_cvt_pk_bf8_f32(c[6], c[7], up, true);
        }
        *(u32x2*)(p.q8 + mo[ps] * p.ldq8 + n) = (u32x2){(unsigned)lo, (unsigned)up};
      }
    }
  }
}

// vdk_margin.h — the margin functions of the faceX heads (ArcFace / CircleLoss / MV-Softmax / MagFace), shared by the row kernels of margin_head.hip and the GEMM epilogues
// that apply them to cos tiles in registers (gemm.hip: E_MSTAT / E_MGRAD).  Reference: models/faceX/head/{arcface,circleloss,mv_softmax,magface}.py.
#pragma once
#include <math.h>
#include "visiondk.h"
#include "vdk_device.h"
#include "vdk_host.h"

struct MarginP {
  int mode;          // VDK_HEAD_ARCFACE / CIRCLE / MV_AM / MV_ARC
  float s;           // scale (arcface, mv) or gamma (circle)
  float m;           // margin
  float cos_m, sin_m, min_cos, m_am;   // arcface: cos(m), sin(m), cos(pi - m), margin_am
  float Op, On, dp, dn;                // circle: 1+m, -m, 1-m, m
  float t;                             // mv_weight
  const float* row_margin;             // arcface only, optional: per-row additive angular margin (MagFace's magnitude-aware margin), overrides m
};
// MagFace (models/faceX/head/magface.py:26-47): ArcFace whose margin is a function of the row's feature norm; the kernels take it per row
__device__ __forceinline__ void margin_row_params(MarginP& P, int row) {
  if (P.row_margin) {
    const float m = P.row_margin[row];
    P.m = m; P.cos_m = cosf(m); P.sin_m = sinf(m); P.min_cos = cosf(3.14159265358979323846f - m);
  }
}

struct RowCtx { float thr, final_gt, dfinal; };   // MV-Softmax per-row quantities derived from gt = cos[i][y_i]

__device__ __forceinline__ RowCtx margin_row_ctx(const MarginP& P, float gt) {
  RowCtx r; r.thr = 0.f; r.final_gt = gt; r.dfinal = 1.f;
  if (P.mode == VDK_HEAD_MV_AM) {
    r.thr = gt - P.m;
    r.final_gt = gt > P.m ? gt - P.m : gt;
  } else if (P.mode == VDK_HEAD_MV_ARC) {
    const float sn = sqrtf(1.0f - gt * gt);
    const float ctm = gt * P.cos_m - sn * P.sin_m;
    r.thr = ctm;
    if (gt > 0.0f) { r.final_gt = ctm; r.dfinal = P.cos_m + gt / sn * P.sin_m; }
  }
  return r;
}
// logit and d(logit)/d(cos) of one entry
__device__ __forceinline__ void margin_eval(const MarginP& P, const RowCtx& R, float c_raw, bool tgt, float& logit, float& jac) {
  if (P.mode == VDK_HEAD_ARCFACE) {
    const float c = fminf(fmaxf(c_raw, -1.0f), 1.0f);
    const float cg = (c_raw >= -1.0f && c_raw <= 1.0f) ? 1.0f : 0.0f;   // clamp backward
    if (!tgt) { logit = P.s * c; jac = P.s * cg; return; }
    if (c > P.min_cos) {
      const float sn = sqrtf(1.0f - c * c);
      logit = P.s * (c * P.cos_m - sn * P.sin_m);
      jac = P.s * (P.cos_m + c / sn * P.sin_m) * cg;
    } else { logit = P.s * (c - P.m_am); jac = P.s * cg; }
  } else if (P.mode == VDK_HEAD_CIRCLE) {
    const float c = fminf(fmaxf(c_raw, -1.0f), 1.0f);
    const float cg = (c_raw >= -1.0f && c_raw <= 1.0f) ? 1.0f : 0.0f;
    if (tgt) { const float a = fmaxf(P.Op - c, 0.f); logit = P.s * a * (c - P.dp); jac = P.s * a * cg; }
    else { const float a = fmaxf(c - P.On, 0.f); logit = P.s * a * (c - P.dn); jac = P.s * a * cg; }
  } else {  // MV-Softmax (no clamp)
    if (tgt) { logit = P.s * R.final_gt; jac = P.s * R.dfinal; }
    else if (c_raw > R.thr) { logit = P.s * (P.t * c_raw + P.t - 1.0f); jac = P.s * P.t; }
    else { logit = P.s * c_raw; jac = P.s; }
  }
}


// v_exp_f32 on x * log2(e): 2 instructions against libm expf's ~12; relative error ~2^-22 on |x| < 90, inside the rounding of the softmax sums
__device__ __forceinline__ float vdk_vexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// what the two margin epilogues of the cos GEMM need beside the tile (GemmParams.me)
struct MarginEpi {
  MarginP P;
  const long long* labels;     // [B]
  const float* gt;             // [B] target cosine per row (MV-Softmax thresholds) or NULL
  float* stats;                // E_MSTAT out: f32 [B][nslice][4] = (max logit, sum exp(logit - max), sum logit, -) per 64-column slice
  long nslice;
  float* tlogit;               // E_MSTAT out: [B] the target's logit (written by the one lane that meets column y_b)
  const float* rowstat;        // E_MGRAD in: f32 [B][2] = (max over the row, 1 / sum exp(logit - max))
  float smoothing, gscale, epsc;
  int B, C;
};

static inline int fill_params(const VdkMarginHead* h, MarginP* P) {
  if (!h) return vdk_fail(VDK_EINVAL, "margin head: null config");
  P->mode = h->mode; P->s = h->scale; P->m = h->margin; P->m_am = h->margin_am; P->t = h->mv_weight;
  P->cos_m = cosf(h->margin); P->sin_m = sinf(h->margin); P->min_cos = cosf(3.14159265358979323846f - h->margin);
  P->Op = 1.0f + h->margin; P->On = -h->margin; P->dp = 1.0f - h->margin; P->dn = h->margin;
  P->row_margin = h->mode == VDK_HEAD_ARCFACE ? h->row_margin : nullptr;
  if (h->mode < VDK_HEAD_ARCFACE || h->mode > VDK_HEAD_MV_ARC) return vdk_fail(VDK_EINVAL, "margin head: bad mode");
  return VDK_OK;
}


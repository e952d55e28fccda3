// vit_engine.hip — native orchestration of hot path A for timm's VisionTransformer
// (vit_base_patch16_224 & friends: class token, learned pos_embed, pre-norm blocks, LayerNorm eps 1e-6,
// fused qkv, exact-erf GELU MLP, final norm, token pooling, Linear head) as created by the reference at
// models/classifier/classify_model.py:49-54 (`timm.create_model(name, num_classes=...)`) and stepped by
// engine/procedure/train.py:177-215.  One C call runs the whole forward (saving what backward needs) and one
// the whole backward, as a fixed sequence of the library's own kernels on ONE stream over caller-owned flat
// buffers: no autograd graph, no per-tensor launches, no allocation.
//
// Data layout (everything in HBM, 288 GB makes recomputation pointless):
//   params / grads / momentum / ema : flat fp32, identical tensor offsets (timm state_dict order, see
//                                     vdk_vit_param_info); wb16 = same layout in bf16 (GEMM B operands,
//                                     refreshed by vdk_sgd_step); wt16 = per-Linear [in, out] bf16 copies for dgrad.
//   residual stream                 : fp32 [B*N, D] per block boundary (what timm keeps in fp32 under autocast)
//   GEMM operands / saved tensors   : bf16
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_gemm.h"

extern "C" {
int vdk_layernorm_fwd(const float*, int64_t, int32_t, int32_t, const float*, const float*, float, void*, int64_t, int32_t, float*, float*, void*);
int vdk_layernorm_bwd_workspace_bytes(int32_t, int32_t, size_t*);
int vdk_layernorm_bwd(const void*, int64_t, int32_t, const float*, int64_t, const float*, const float*, const float*, const float*, int64_t,
                      int32_t, int32_t, float*, int64_t, void*, int64_t, float*, float*, void*, size_t, void*);
int vdk_reduce_rows_f32(const float*, int64_t, int32_t, int64_t, float*, float, void*);
int vdk_colsum_bf16_workspace_bytes(int32_t, int32_t, size_t*);
int vdk_colsum_bf16(const void*, int64_t, int32_t, int32_t, float*, void*, size_t, void*);
int vdk_attention_fwd(const void*, int64_t, void*, int64_t, float*, int32_t, int32_t, int32_t, int32_t, float, void*);
int vdk_attention_bwd(const void*, int64_t, const void*, const void*, int64_t, const float*, void*, int64_t, float*, int32_t, int32_t, int32_t,
                      int32_t, float, void*);
int vdk_attention_fwd_dt(const void*, int64_t, void*, int64_t, float*, int32_t, int32_t, int32_t, int32_t, float, int32_t, void*);
int vdk_attention_bwd_dt(const void*, int64_t, const void*, const void*, int64_t, const float*, void*, int64_t, float*, int32_t, int32_t, int32_t, int32_t, float, int32_t, void*);
int vdk_attention_bwd_cs(const void*, int64_t, const void*, const void*, int64_t, const float*, void*, int64_t, float*, int32_t, int32_t, int32_t, int32_t, float, int32_t, float*, int32_t*,
                         void*);
int vdk_patchify_bf16(const float*, int32_t, int32_t, int32_t, int32_t, int32_t, void*, int32_t, void*);
int vdk_cls_rows(float*, int64_t, int32_t, int32_t, const float*, const float*, void*);
int vdk_cast_f32_bf16(const float*, void*, int64_t, void*);
int vdk_transpose_cast_f32_bf16(const float*, int64_t, int32_t, int32_t, void*, int64_t, int32_t, void*);
int vdk_gemm_f32_nt(const VdkGemmF32Desc*, void*);
int vdk_gemm_a_colsum_rows(int32_t, int32_t, int32_t);
int vdk_softmax_rows_f32(float*, int64_t, int64_t, int32_t, float, void*);
int vdk_patchify_f32(const float*, int32_t, int32_t, int32_t, int32_t, int32_t, float*, void*);
int vdk_quant_fp8(const void*, int32_t, int64_t, const float*, void*, int32_t, float*, void*);
int vdk_fp8_scale_update(float*, float*, float*, int32_t, int32_t, float, void*);
int vdk_gemm_fp8_nt(const VdkGemmDesc*, int32_t, int32_t, const float*, const float*, void*);
int vdk_gemm_fp8_nt_q8(const VdkGemmDesc*, int32_t, int32_t, const float*, const float*, void*, int64_t, int32_t, const float*, float*, void*);
int vdk_layernorm_fwd_q8(const float*, int64_t, int32_t, int32_t, const float*, const float*, float, void*, int64_t, float*, float*, void*, int64_t, int32_t, const float*, float*, void*);
}


// Operand format of the engine call in progress on this thread: every entry point sets it from VdkVitConfig.operand before it enqueues anything, the helpers below read
// it (an engine call runs to completion on its thread).  DT16 = the dtype code of the 16-bit tensors.
static thread_local int t_opf = VDK_OPF_BF16;
#define DT16 (t_opf ? VDK_F16 : VDK_BF16)
// What the MLP keeps for the backward pass in `u` [T, mlp_dim]: the pre-activation in the operand format (the backward evaluates GELU' of it inside the dfc2 epilogue), or --
// VDK_VIT_GELU_SAVED_GRAD=1 -- GELU'(pre-activation) evaluated once in the fc1 epilogue from the SAME erf / exp terms as GELU itself and stored as fp16 (the backward's
// epilogue is then one multiplication).  Measured per layer at ViT-B/16 (profiles/r03_gemm_ab.json): forward 305 -> 332 us, backward 321 -> 282 us.
// Round 5: DEFAULT with fp16 operands -- `u` is rounded to fp16 either way, so the saved derivative costs no accuracy there (logits are bit-identical: the forward's GELU value
// does not change), and the step gains what round 4 measured; bf16 operands keep the pre-activation.  VDK_VIT_GELU_SAVED_GRAD=0 / 1 forces it for both formats.
static bool gelu_saved_grad() {
  static const int v = [] { const char* e = getenv("VDK_VIT_GELU_SAVED_GRAD"); return e ? atoi(e) : -1; }();
  return v < 0 ? t_opf != 0 : v != 0;
}
#define ACT_FC1 (gelu_saved_grad() ? VDK_ACT_GELU_SAVE_GRAD : VDK_ACT_GELU)
#define ACT_DFC2 (gelu_saved_grad() ? VDK_ACT_MUL_AUX : VDK_ACT_DGELU)

static inline int64_t up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct VitDims {
  int B, img, ps, Cin, D, L, H, M, C, Cp, np, N, T, Kraw, Kpe, Tp, Bp, cls, pre /* VdkVitConfig.pre_norm */, fp8, opf /* operand format: VDK_OPF_BF16 | VDK_OPF_F16 (VdkVitConfig.operand) */;   // cls: 1 = class token in row 0 of every image's token block   // Kraw = in_chans * patch^2, Kpe = Kraw padded to 8 (patch 14: 588 -> 592)
  float eps;
};
static int vit_dims(const VdkVitConfig* c, VitDims* d) {
  if (!c) return vdk_fail(VDK_EINVAL, "vit: null config");
  d->B = c->batch; d->img = c->img_size; d->ps = c->patch_size; d->Cin = c->in_chans; d->D = c->dim; d->L = c->depth;
  d->H = c->heads; d->M = c->mlp_dim; d->C = c->num_classes; d->eps = c->ln_eps;
  if (d->B <= 0 || d->img <= 0 || d->ps <= 0 || d->img % d->ps || d->Cin <= 0 || d->D <= 0 || d->L <= 0 || d->H <= 0 || d->M <= 0 || d->C < 0)
    return vdk_fail(VDK_EINVAL, "vit: bad config value");
  if (d->D != d->H * 64) return vdk_fail(VDK_EUNSUPPORTED, "vit: head_dim must be 64 (dim == 64 * heads)");
  if ((d->D & 7) || (d->M & 7)) return vdk_fail(VDK_EUNSUPPORTED, "vit: dim and mlp_dim must be multiples of 8");
  d->Kraw = d->Cin * d->ps * d->ps;
  d->Kpe = (int)up(d->Kraw, 8);     // patch 14 (timm vit_*_patch14_*: K = 588): the GEMM operands are zero-padded copies, the parameter itself stays [D, Kraw]
  d->np = (d->img / d->ps) * (d->img / d->ps);
  d->cls = c->no_class_token ? 0 : 1;
  d->pre = c->pre_norm ? 1 : 0;
  if (d->pre && c->fp8) return vdk_fail(VDK_EUNSUPPORTED, "vit: pre_norm and the fp8 mode are not combined");
  if (!d->cls && d->C > 0) return vdk_fail(VDK_EUNSUPPORTED, "vit: no_class_token is feature mode only (num_classes == 0)");
  d->N = d->np + d->cls;
  d->T = d->B * d->N;
  d->Cp = (int)up(d->C, 8);
  d->Tp = (int)up(d->T, 64);
  d->Bp = (int)up(d->B, 64);
  d->fp8 = c->fp8;
  if (d->fp8 < 0 || d->fp8 > 2) return vdk_fail(VDK_EINVAL, "vit: fp8 must be 0, 1 or 2");
  if (c->operand != VDK_BF16 && c->operand != VDK_F16) return vdk_fail(VDK_EINVAL, "vit: operand must be VDK_BF16 or VDK_F16");
  d->opf = c->operand == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16;
  if (d->opf && d->fp8) return vdk_fail(VDK_EUNSUPPORTED, "vit: the fp8 mode goes with bf16 operands");
  if (d->fp8 && (d->D < 256 || (d->D % 128) || (d->M % 128))) return vdk_fail(VDK_EUNSUPPORTED, "vit: the fp8 mode needs dim >= 256, dim and mlp_dim % 128 == 0");
  return VDK_OK;
}

// ---------------------------------------------------------------------------- parameter layout
struct PEntry { char name[64]; int64_t off, numel; int64_t shape[4]; int ndim; };
struct PLayout {
  // offsets into the flat fp32/bf16 buffers
  int64_t cls, pos, pe_w, pe_b, npre_w, npre_b, norm_w, norm_b, head_w, head_b, total;      // (pre_norm: pe_b stays allocated and zero, npre_* = norm_pre)
  struct Blk { int64_t n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b; };
  Blk blk[64];
  // offsets into the transposed-weight buffer (bf16 elements)
  struct BlkT { int64_t qkv, proj, fc1, fc2; } blkT[64];
  int64_t headT, totalT;
};
static int64_t p_take(int64_t& cur, int64_t n) { int64_t o = cur; cur = up(cur + n, 64); return o; }
static int vit_layout(const VitDims& d, PLayout* p) {
  if (d.L > 64) return vdk_fail(VDK_EUNSUPPORTED, "vit: depth > 64");
  int64_t cur = 0;
  p->cls = d.cls ? p_take(cur, d.D) : 0;
  p->pos = p_take(cur, (int64_t)d.N * d.D);
  p->pe_w = p_take(cur, (int64_t)d.D * d.Kraw);
  p->pe_b = p_take(cur, d.D);
  p->npre_w = p->npre_b = 0;
  if (d.pre) { p->npre_w = p_take(cur, d.D); p->npre_b = p_take(cur, d.D); }
  for (int l = 0; l < d.L; ++l) {
    PLayout::Blk& b = p->blk[l];
    b.n1w = p_take(cur, d.D); b.n1b = p_take(cur, d.D);
    b.qkv_w = p_take(cur, (int64_t)3 * d.D * d.D); b.qkv_b = p_take(cur, 3 * d.D);
    b.proj_w = p_take(cur, (int64_t)d.D * d.D); b.proj_b = p_take(cur, d.D);
    b.n2w = p_take(cur, d.D); b.n2b = p_take(cur, d.D);
    b.fc1_w = p_take(cur, (int64_t)d.M * d.D); b.fc1_b = p_take(cur, d.M);
    b.fc2_w = p_take(cur, (int64_t)d.D * d.M); b.fc2_b = p_take(cur, d.D);
  }
  p->norm_w = p_take(cur, d.D); p->norm_b = p_take(cur, d.D);
  p->head_w = p->head_b = cur;
  if (d.C > 0) {
    p->head_w = p_take(cur, (int64_t)d.Cp * d.D);   // rows C..Cp-1 are zero padding (N % 8 for the GEMM)
    p->head_b = p_take(cur, d.Cp);
  }
  p->total = cur;
  int64_t t = 0;
  for (int l = 0; l < d.L; ++l) {
    p->blkT[l].qkv = p_take(t, (int64_t)d.D * 3 * d.D);
    p->blkT[l].proj = p_take(t, (int64_t)d.D * d.D);
    p->blkT[l].fc1 = p_take(t, (int64_t)d.D * d.M);
    p->blkT[l].fc2 = p_take(t, (int64_t)d.M * d.D);
  }
  p->headT = t;
  if (d.C > 0) p->headT = p_take(t, (int64_t)d.D * d.Cp);
  p->totalT = t;
  return VDK_OK;
}

static void pe_set(PEntry* e, const char* name, int64_t off, int ndim, int64_t s0, int64_t s1 = 1, int64_t s2 = 1, int64_t s3 = 1) {
  snprintf(e->name, sizeof(e->name), "%s", name);
  e->off = off; e->ndim = ndim; e->shape[0] = s0; e->shape[1] = s1; e->shape[2] = s2; e->shape[3] = s3;
  e->numel = s0 * s1 * s2 * s3;
}
// timm state_dict order and names; head rows are reported unpadded ([C, D]; the padding rows
// follow in memory and must stay zero).
static int vit_entry(const VitDims& d, const PLayout& p, int idx, PEntry* e) {
  const int per = 12, nb0 = d.pre ? 5 : 4, ntens = nb0 + per * d.L + (d.C > 0 ? 4 : 2);   // feature mode (num_classes = 0) has no head; pre_norm: norm_pre.{weight, bias} instead of patch_embed.proj.bias
  if (!d.cls) { if (idx < 0) return -1; ++idx; }                    // class_token=False: the state dict starts at pos_embed
  if (idx < 0 || idx >= ntens) return -1;
  char nm[64];
  if (idx == 0) { pe_set(e, "cls_token", p.cls, 3, 1, 1, d.D); return 0; }
  if (idx == 1) { pe_set(e, "pos_embed", p.pos, 3, 1, d.N, d.D); return 0; }
  if (idx == 2) { pe_set(e, "patch_embed.proj.weight", p.pe_w, 4, d.D, d.Cin, d.ps, d.ps); return 0; }
  if (idx == 3 && !d.pre) { pe_set(e, "patch_embed.proj.bias", p.pe_b, 1, d.D); return 0; }
  if (idx == 3) { pe_set(e, "norm_pre.weight", p.npre_w, 1, d.D); return 0; }
  if (idx == 4 && d.pre) { pe_set(e, "norm_pre.bias", p.npre_b, 1, d.D); return 0; }
  if (idx < nb0 + per * d.L) {
    int l = (idx - nb0) / per, k = (idx - nb0) % per;
    const PLayout::Blk& b = p.blk[l];
    static const char* suffix[12] = {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                                     "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"};
    snprintf(nm, sizeof(nm), "blocks.%d.%s", l, suffix[k]);
    switch (k) {
      case 0: pe_set(e, nm, b.n1w, 1, d.D); break;
      case 1: pe_set(e, nm, b.n1b, 1, d.D); break;
      case 2: pe_set(e, nm, b.qkv_w, 2, 3 * d.D, d.D); break;
      case 3: pe_set(e, nm, b.qkv_b, 1, 3 * d.D); break;
      case 4: pe_set(e, nm, b.proj_w, 2, d.D, d.D); break;
      case 5: pe_set(e, nm, b.proj_b, 1, d.D); break;
      case 6: pe_set(e, nm, b.n2w, 1, d.D); break;
      case 7: pe_set(e, nm, b.n2b, 1, d.D); break;
      case 8: pe_set(e, nm, b.fc1_w, 2, d.M, d.D); break;
      case 9: pe_set(e, nm, b.fc1_b, 1, d.M); break;
      case 10: pe_set(e, nm, b.fc2_w, 2, d.D, d.M); break;
      default: pe_set(e, nm, b.fc2_b, 1, d.D); break;
    }
    return 0;
  }
  int k = idx - nb0 - per * d.L;
  if (k == 0) pe_set(e, "norm.weight", p.norm_w, 1, d.D);
  else if (k == 1) pe_set(e, "norm.bias", p.norm_b, 1, d.D);
  else if (k == 2) pe_set(e, "head.weight", p.head_w, 2, d.C, d.D);
  else pe_set(e, "head.bias", p.head_b, 1, d.C);
  return 0;
}

// ---------------------------------------------------------------------------- workspace plan
struct WsPlan {
  size_t total;
  size_t patches;            // bf16 [B*np, Kpe]
  size_t pepad, dwpe;        // Kraw != Kpe only: bf16 [D, Kpe] zero-padded copy of patch_embed.proj.weight, f32 [D, Kpe] its padded gradient
  size_t X;                  // fp32 (2L+1) x [T, D]  : X[2l] block input, X[2l+1] after attention, X[2L] output
  size_t xe, pstats;         // pre_norm only: fp32 [T, D] the embedding in front of norm_pre, fp32 2 x [T] its row statistics
  size_t stats;              // fp32 L x 4 x [T] (mean1, rstd1, mean2, rstd2) + 2 x [B]
  size_t h1, qkv, lse, o, h2, u, g;   // per-layer strides below
  size_t s_h, s_qkv, s_lse, s_u;      // per-layer sizes in bytes
  size_t hf;                 // bf16 [Bp, D] (zero padded rows)
  // backward scratch
  size_t dxa, dxm;               // fp32 [T,D] x2 (main stream only)
  size_t dxab[3], dxmb[2];       // bf16 [T,D]: dxab(l) is written at the END of layer l+1 -> 3 buffers; dxmb ping-pongs on the layer parity
  size_t du[2], dqkv[2];         // bf16 [T,M], [T,3D], ping-pong per layer parity
  size_t dsm;                    // bf16 [T, D]           (dh / do)
  size_t dvec;                   // fp32 [B,H,N]
  size_t tA, tB;                 // bf16 [max(M,3D,Cp), Tp] each
  size_t slabs, slabs_bytes;
  size_t lnws, lnws_bytes, csws, csws_bytes;   // lnws: 2 buffers of lnws_bytes, csws: 5 of csws_bytes (a block's deferred reductions read them at its end)
  size_t dhf;                    // bf16 [B, D]
  size_t dposall;                // fp32 [N, D]
  size_t a8;                     // fp8 mode: the quantised A operand of the GEMM about to run, [T, max(M, 3D)] bytes
  size_t a8b;                    // fp8 mode: the fp8 copy a GELU / dGELU epilogue writes for the NEXT GEMM (while a8 is still being read), [T, M] bytes
};
static size_t w_take(size_t& cur, size_t n) { size_t o = cur; cur = (cur + n + 255) & ~(size_t)255; return o; }
static int wgrad_splitk(int M, int N, int K) {
  if (K < 4096) return 1;
  int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int s = (1024 + tiles - 1) / tiles;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return s;
}
// TN LDS-DMA kernel: 256x256 tiles, one workgroup per CU -> pick the split count that makes tiles * splits land just under a
// whole number of 256-CU rounds (a second, nearly empty round would halve the efficiency)
static int wgrad_splitk_tn(int M, int N, int K) {
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  int s = 256 / tiles;
  if (s < 1) s = 1;
  const int kt = K / 64;
  if (s > kt / 4) s = kt / 4;            // keep >= 4 k-tiles per split
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}
static int vit_plan(const VitDims& d, WsPlan* w) {
  size_t cur = 0;
  const size_t T = d.T, D = d.D, M = d.M, L = d.L;
  w->patches = w_take(cur, (size_t)d.B * d.np * d.Kpe * 2);
  w->pepad = w->dwpe = 0;
  if (d.Kraw != d.Kpe) { w->pepad = w_take(cur, D * d.Kpe * 2); w->dwpe = w_take(cur, D * d.Kpe * 4); }
  w->X = w_take(cur, (2 * L + 1) * T * D * 4);
  w->xe = w->pstats = 0;
  if (d.pre) { w->xe = w_take(cur, T * D * 4); w->pstats = w_take(cur, 2 * T * 4); }
  w->stats = w_take(cur, (L * 4 * T + 2 * T) * 4);   // + final norm: B rows (token pooling) or all T rows (feature mode)
  w->s_h = T * D * 2; w->s_qkv = T * 3 * D * 2; w->s_lse = (size_t)d.B * d.H * d.N * 4; w->s_u = T * M * 2;
  w->h1 = w_take(cur, L * w->s_h); w->qkv = w_take(cur, L * w->s_qkv); w->lse = w_take(cur, L * w->s_lse);
  w->o = w_take(cur, L * w->s_h); w->h2 = w_take(cur, L * w->s_h); w->u = w_take(cur, L * w->s_u); w->g = w_take(cur, L * w->s_u);
  w->hf = w_take(cur, (size_t)d.Bp * D * 2);
  w->dxa = w_take(cur, T * D * 4); w->dxm = w_take(cur, T * D * 4);
  size_t big = M > 3 * D ? M : 3 * D;
  for (int i = 0; i < 3; ++i) w->dxab[i] = w_take(cur, T * D * 2);
  for (int i = 0; i < 2; ++i) {
    w->dxmb[i] = w_take(cur, T * D * 2);
    w->du[i] = w_take(cur, T * M * 2); w->dqkv[i] = w_take(cur, T * 3 * D * 2);
  }
  w->dsm = w_take(cur, T * D * 2);
  w->dvec = w_take(cur, w->s_lse);
  size_t trows = big > (size_t)d.Cp ? big : (size_t)d.Cp;
  if (trows < (size_t)d.Kpe) trows = d.Kpe;
  size_t tcols = d.Tp > d.Bp ? d.Tp : d.Bp;
  w->tA = w_take(cur, trows * tcols * 2); w->tB = w_take(cur, trows * tcols * 2);
  size_t sl = 0;
  {
    int sh[5][3] = {{(int)M, (int)D, d.T}, {(int)D, (int)M, d.T}, {3 * (int)D, (int)D, d.T}, {(int)D, (int)D, d.T}, {(int)D, d.Kpe, d.B * d.np}};
    for (auto& s : sh) {
      int k1 = wgrad_splitk(s[0], s[1], s[2]), k2 = wgrad_splitk_tn(s[0], s[1], s[2]);
      size_t b = (size_t)(k1 > k2 ? k1 : k2) * s[0] * s[1] * 4; if (b > sl) sl = b;
    }
  }
  w->slabs_bytes = sl; w->slabs = w_take(cur, sl);
  vdk_layernorm_bwd_workspace_bytes(d.T, d.D, &w->lnws_bytes); w->lnws_bytes = (w->lnws_bytes + 255) & ~(size_t)255; w->lnws = w_take(cur, 2 * w->lnws_bytes);
  size_t cs = (size_t)((tcols + 63) / 64) * trows * 4;   // per-row-tile column sums written by the dY transposes ...
  { size_t cs2 = 0; vdk_colsum_bf16_workspace_bytes(d.T, (int)trows, &cs2); if (cs2 > cs) cs = cs2; }   // ... or by vdk_colsum_bf16
  { size_t cs3 = (size_t)2 * ((d.T + 255) / 256) * trows * 4; if (cs3 > cs) cs = cs3; }                  // ... or by a dgrad GEMM's a_colsum / c_colsum by-product
  w->csws_bytes = (cs + 255) & ~(size_t)255; w->csws = w_take(cur, 5 * w->csws_bytes);   // slots 0..3: a block's four fused bias-gradient partials (pending until its end), slot 4: immediate users
  w->dhf = w_take(cur, (size_t)d.B * D * 2);
  w->dposall = w_take(cur, (size_t)d.N * D * 4);
  w->a8 = d.fp8 ? w_take(cur, T * (M > 3 * D ? M : 3 * D)) : 0;
  w->a8b = d.fp8 ? w_take(cur, T * M) : 0;
  w->total = cur;
  return VDK_OK;
}

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// ---- fp8 mode (VdkVitConfig.fp8): OCP e4m3 / e5m2 operands for the forward and input-gradient GEMMs of the block Linears ------------------------------------
struct F8 {
  int mode; unsigned char* w8; unsigned char* wt8; float* amax; float* sc; float* si; unsigned char* a8; unsigned char* a8b;
};
static int f8_init(const VdkVitConfig* cfg, const VitDims& d, const PLayout& p, F8* f, unsigned char* a8, unsigned char* a8b) {
  f->mode = d.fp8; f->a8 = a8; f->a8b = a8b;
  if (!d.fp8) return VDK_OK;
  if (!cfg->fp8_w || !cfg->fp8_state) return vdk_fail(VDK_EINVAL, "vit: fp8 mode without fp8_w / fp8_state");
  if (a8 && d.T < 256) return vdk_fail(VDK_EUNSUPPORTED, "vit: the fp8 mode needs batch * tokens >= 256");
  f->w8 = (unsigned char*)cfg->fp8_w; f->wt8 = f->w8 + p.total;
  const int S = 12 * d.L;
  f->amax = cfg->fp8_state; f->sc = f->amax + S; f->si = f->sc + S;
  return VDK_OK;
}
// x (bf16 | f32, n values) -> fp8 bytes with the slot's scale; records amax for the next scale.  mode 2: the scale is refreshed from THIS tensor first (current scaling)
static int f8_quant(hipStream_t s, const F8& f, const void* x, int xdt, long n, int slot, int fmt, unsigned char* out, int force_current = 0) {
  if (f.mode == 2 || force_current) {
    RC(vdk_quant_fp8(x, xdt, n, nullptr, nullptr, fmt, f.amax + slot, s));
    RC(vdk_fp8_scale_update(f.amax + slot, f.sc + slot, f.si + slot, 1, fmt, 1.0f, s));
  }
  return vdk_quant_fp8(x, xdt, n, f.sc + slot, out, fmt, f.amax + slot, s);
}
// C = epilogue(A8[M, K] . W8[N, K]^T / (scale_a scale_w)): `a` is quantised into the scratch operand first -- unless the GEMM that produced it already wrote its fp8 copy
// (a_pre: f.a8b made by a GEMM epilogue, or f.a8 filled by a LayerNorm kernel, with slot_a's scale).  q8_slot >= 0: this GEMM's epilogue writes the fp8 copy of ITS bf16 output for the next one (delayed
// scaling only: the scale must be known before the values are; the calibration step and mode 2 keep the separate pass).
static bool f8_fused(const F8& f) {
  const char* e = getenv("VDK_FP8_FUSED_QUANT");                        // A/B and tests: 0 keeps the separate quantisation passes (read per call: the tests switch it in-process)
  return f.mode == 1 && !(e && e[0] == '0');
}
static int gemm8(hipStream_t s, const F8& f, const void* a, int slot_a, int a_fmt, const unsigned char* w8, int slot_w, int64_t ldb, void* Cc, int64_t ldc, int M, int N,
                 int K, int cdt, const float* bias, const float* res, int64_t ldr, int act, void* aux, int64_t ldaux, const unsigned char* a_pre = nullptr, int q8_slot = -1,
                 int q8_fmt = 0, float* c_colsum = nullptr) {
  if (!a_pre) RC(f8_quant(s, f, a, VDK_BF16, (long)M * K, slot_a, a_fmt, f.a8));
  VdkGemmDesc g = {};
  g.A = a_pre ? a_pre : f.a8; g.lda = K; g.B = w8; g.ldb = ldb; g.C = Cc; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.c_dtype = cdt; g.bias = bias; g.residual = res; g.ldr = ldr;
  g.act = act; g.aux = aux; g.ldaux = ldaux; g.alpha = 1.0f; g.splitk = 1; g.c_colsum = c_colsum;
  if (q8_slot >= 0) return vdk_gemm_fp8_nt_q8(&g, a_fmt, 0, f.si + slot_a, f.si + slot_w, f.a8b, N, q8_fmt, f.sc + q8_slot, f.amax + q8_slot, s);
  return vdk_gemm_fp8_nt(&g, a_fmt, 0, f.si + slot_a, f.si + slot_w, s);
}
__global__ void vit_fp8_update_kernel(float* __restrict__ amax, float* __restrict__ sc, float* __restrict__ si, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = amax[i], fmax = (i % 12) >= 8 ? 57344.0f : 448.0f;       // slots 8..11 of a layer are e5m2 gradients
  if (a > 0.f && a < 3.0e38f) { const float v = fmax / a; sc[i] = v; si[i] = 1.0f / v; }   // a non-finite amax (one overflowed step) keeps the previous scale instead of poisoning it: scale 0 / inf would make every later output NaN, fmaxf would then drop the NaNs and the a > 0 guard would keep scale 0 forever
  amax[i] = 0.f;
}

// Events that order the main (dgrad) stream and the side (wgrad) stream of vdk_vit_backward.  Created once per (thread, device),
// timing disabled, re-recorded on every call (the only library-owned state; streams and memory stay the caller's).
#include <map>
#include <vector>
static thread_local std::map<int, std::vector<hipEvent_t>> g_ev_dev;      // per calling thread and per device (an event belongs to the device current at its creation)
static int ev_get(size_t i, hipEvent_t* e) {
  int dev = 0; (void)hipGetDevice(&dev);
  std::vector<hipEvent_t>& g_ev = g_ev_dev[dev];
  while (g_ev.size() <= i) {
    hipEvent_t x;
    if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vit: hipEventCreate failed");
    g_ev.push_back(x);
  }
  *e = g_ev[i];
  return VDK_OK;
}
// "everything enqueued so far on `from` happens before anything enqueued later on `to`"
static int ev_order(size_t slot, hipStream_t from, hipStream_t to) {
  if (from == to) return VDK_OK;
  hipEvent_t e; RC(ev_get(slot, &e));
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vit: event record/wait failed");
  return VDK_OK;
}

static int gemm(hipStream_t s, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, int cdt,
                const float* bias, const float* res, int64_t ldr, int act, void* aux, int64_t ldaux, int splitk, int row_group, void* ws,
                size_t wsb) {
  VdkGemmDesc g = {};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.c_dtype = cdt == VDK_F32 ? VDK_F32 : DT16; g.bias = bias;
  g.residual = res; g.ldr = ldr; g.act = act; g.aux = aux; g.ldaux = ldaux; g.alpha = 1.0f; g.splitk = splitk; g.row_group = row_group; g.ab_dtype = DT16;
  return vdk_gemm_bf16_nt(&g, ws, wsb, s);
}

extern "C" {

int vdk_vit_param_count(const VdkVitConfig* cfg, int64_t* n_floats, int32_t* n_tensors, int64_t* n_transposed) {
  VitDims d; RC(vit_dims(cfg, &d));
  PLayout p; RC(vit_layout(d, &p));
  if (n_floats) *n_floats = p.total;
  if (n_tensors) *n_tensors = 3 + d.pre + d.cls + 12 * d.L + (d.C > 0 ? 4 : 2);
  if (n_transposed) *n_transposed = p.totalT;
  return VDK_OK;
}

int vdk_vit_param_info(const VdkVitConfig* cfg, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel,
                       int64_t* shape4, int32_t* ndim) {
  VitDims d; RC(vit_dims(cfg, &d));
  PLayout p; RC(vit_layout(d, &p));
  PEntry e;
  if (vit_entry(d, p, index, &e)) return vdk_fail(VDK_EINVAL, "vdk_vit_param_info: index out of range");
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", e.name);
  if (offset) *offset = e.off;
  if (numel) *numel = e.numel;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = e.shape[i];
  if (ndim) *ndim = e.ndim;
  return VDK_OK;
}

int vdk_vit_workspace_bytes(const VdkVitConfig* cfg, size_t* bytes) {
  VitDims d; RC(vit_dims(cfg, &d));
  WsPlan w; RC(vit_plan(d, &w));
  if (!bytes) return vdk_fail(VDK_EINVAL, "null");
  *bytes = w.total;
  return VDK_OK;
}

// (Re)build the bf16 operand copies from the fp32 master weights: wb16 (same layout) unless `skip_wb16`
// (vdk_sgd_step already refreshed it), and the [in, out] transposes the dgrad GEMMs read.
int vdk_vit_refresh_weights(const VdkVitConfig* cfg, const float* params, void* wb16, void* wt16, int32_t skip_wb16, void* stream) {
  VitDims d; RC(vit_dims(cfg, &d));
  PLayout p; RC(vit_layout(d, &p));
  if (!params || !wb16 || !wt16) return vdk_fail(VDK_EINVAL, "vdk_vit_refresh_weights: null pointer");
  t_opf = d.opf;
  if (!skip_wb16) RC(vdk_cast_f32_16(params, wb16, p.total, t_opf, stream));
  bf16_t* wt = (bf16_t*)wt16;
  std::vector<VdkTcItem> jobs;      // every [in, out] transposed operand copy of the step in one launch (49 jobs for 12 layers)
  auto add = [&](const float* in, int ldi, int R, int Cc, bf16_t* out, int ldo, int Rpad) { jobs.push_back(VdkTcItem{in, out, ldi, R, Cc, ldo, Rpad}); };
  for (int l = 0; l < d.L; ++l) {
    add(params + p.blk[l].qkv_w, d.D, 3 * d.D, d.D, wt + p.blkT[l].qkv, 3 * d.D, 3 * d.D);
    add(params + p.blk[l].proj_w, d.D, d.D, d.D, wt + p.blkT[l].proj, d.D, d.D);
    add(params + p.blk[l].fc1_w, d.D, d.M, d.D, wt + p.blkT[l].fc1, d.M, d.M);
    add(params + p.blk[l].fc2_w, d.M, d.D, d.M, wt + p.blkT[l].fc2, d.D, d.D);
  }
  if (d.C > 0) add(params + p.head_w, d.D, d.Cp, d.D, wt + p.headT, d.Cp, d.Cp);
  RC(vdk_transpose_cast_batch(jobs.data(), (int)jobs.size(), stream, t_opf));
  if (d.fp8) {     // e4m3 copies of the block Linears' weights, both orientations, one scale per tensor taken from the weights themselves (current scaling)
    F8 f; RC(f8_init(cfg, d, p, &f, nullptr, nullptr));
    hipStream_t s = (hipStream_t)stream;
    for (int l = 0; l < d.L; ++l) {
      const PLayout::Blk& b = p.blk[l]; const PLayout::BlkT& bt = p.blkT[l];
      const int64_t off[4] = {b.qkv_w, b.proj_w, b.fc1_w, b.fc2_w}, offT[4] = {bt.qkv, bt.proj, bt.fc1, bt.fc2};
      const long n[4] = {(long)3 * d.D * d.D, (long)d.D * d.D, (long)d.M * d.D, (long)d.D * d.M};
      for (int k = 0; k < 4; ++k) {
        RC(f8_quant(s, f, params + off[k], VDK_F32, n[k], 12 * l + 4 + k, 0, f.w8 + off[k], 1));
        RC(vdk_quant_fp8(wt + offT[k], VDK_BF16, n[k], f.sc + 12 * l + 4 + k, f.wt8 + offT[k], 0, nullptr, s));
      }
    }
  }
  return VDK_OK;
}

int vdk_vit_fp8_update(const VdkVitConfig* cfg, void* stream) {
  VitDims d; RC(vit_dims(cfg, &d));
  if (!d.fp8) return VDK_OK;
  if (!cfg->fp8_state) return vdk_fail(VDK_EINVAL, "vdk_vit_fp8_update: null state");
  const int S = 12 * d.L;
  hipLaunchKernelGGL(vit_fp8_update_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cfg->fp8_state, cfg->fp8_state + S, cfg->fp8_state + 2 * S, S);
  return vdk_check_launch("vdk_vit_fp8_update");
}

// x: f32 [B, Cin, img, img] -> logits f32 [B, Cp] (columns C..Cp-1 are padding); feature mode (num_classes = 0): `logits`
// receives the final-normed tokens f32 [B*N, D].  Saves activations in ws.
int vdk_vit_forward(const VdkVitConfig* cfg, const float* x, const float* params, const void* wb16, void* ws, size_t ws_bytes,
                    float* logits, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  VitDims d; RC(vit_dims(cfg, &d));
  PLayout p; RC(vit_layout(d, &p));
  WsPlan w; RC(vit_plan(d, &w));
  if (!x || !params || !wb16 || !ws || !logits) return vdk_fail(VDK_EINVAL, "vdk_vit_forward: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_vit_forward: workspace too small");
  t_opf = d.opf;
  char* base = (char*)ws;
  const bf16_t* wb = (const bf16_t*)wb16;
  const int T = d.T, D = d.D, M = d.M;
  float* X = (float*)(base + w.X);
  float* stats = (float*)(base + w.stats);
  const size_t XS = (size_t)T * D;

  // patch embedding: im2col-free operand + GEMM whose epilogue drops each patch row at its token slot and adds pos_embed
  bf16_t* patches = (bf16_t*)(base + w.patches);
  RC(vdk_patchify_16(x, d.B, d.Cin, d.img, d.img, d.ps, patches, d.Kpe, t_opf, s));
  const bf16_t* pew = wb + p.pe_w;
  if (d.Kraw != d.Kpe) {           // rows of 588 bf16 are not 16-byte aligned: the GEMM reads a zero-padded [D, Kpe] copy rebuilt from the master weights
    RC(vdk_cast_pad_rows(params + p.pe_w, d.Kraw, D, d.Kraw, base + w.pepad, d.Kpe, s, t_opf));
    pew = (const bf16_t*)(base + w.pepad);
  }
  float* const X0 = d.pre ? (float*)(base + w.xe) : X;      // pre_norm: the embedding lands in front of norm_pre, whose output is block 0's input
  RC(gemm(s, patches, d.Kpe, pew, d.Kpe, X0, D, d.B * d.np, D, d.Kpe, VDK_F32, params + p.pe_b, params + p.pos, D, VDK_ACT_NONE,
          nullptr, 0, 1, d.cls ? d.np : -d.np, nullptr, 0));      // (pre_norm: the bias slot holds zeros -- timm builds that patch embedding without a bias)
  if (d.cls) RC(vdk_cls_rows(X0, (int64_t)d.N * D, d.B, D, params + p.cls, params + p.pos, s));
  if (d.pre) {
    float* ps = (float*)(base + w.pstats);
    RC(vdk_layernorm_fwd(X0, D, T, D, params + p.npre_w, params + p.npre_b, d.eps, X, D, VDK_F32, ps, ps + T, s));
  }

  const float scale = 0.125f;  // head_dim ** -0.5, head_dim == 64
  F8 f8; RC(f8_init(cfg, d, p, &f8, (unsigned char*)(base + w.a8), (unsigned char*)(base + w.a8b)));
  for (int l = 0; l < d.L; ++l) {
    const PLayout::Blk& b = p.blk[l];
    float* xin = X + (size_t)(2 * l) * XS; float* xmid = xin + XS; float* xout = xmid + XS;
    float* mean1 = stats + (size_t)l * 4 * T; float* rstd1 = mean1 + T; float* mean2 = rstd1 + T; float* rstd2 = mean2 + T;
    bf16_t* h1 = (bf16_t*)(base + w.h1 + l * w.s_h); bf16_t* qkv = (bf16_t*)(base + w.qkv + l * w.s_qkv);
    float* lse = (float*)(base + w.lse + l * w.s_lse); bf16_t* o = (bf16_t*)(base + w.o + l * w.s_h);
    bf16_t* h2 = (bf16_t*)(base + w.h2 + l * w.s_h); bf16_t* u = (bf16_t*)(base + w.u + l * w.s_u); bf16_t* g = (bf16_t*)(base + w.g + l * w.s_u);
    // x = x + proj(attn(norm1(x)))
    if (f8.mode) {
      // fp8 mode, delayed scaling: the LayerNorm kernels write the fp8 copy of h1 / h2 straight into the operand scratch, the fc1 + GELU epilogue the one of g
      const int sl = 12 * l;
      const bool fl = f8_fused(f8) && D > 128 && D <= 1024, fq = f8_fused(f8) && (M % 64) == 0;
      if (fl) RC(vdk_layernorm_fwd_q8(xin, D, T, D, params + b.n1w, params + b.n1b, d.eps, h1, D, mean1, rstd1, f8.a8, D, 0, f8.sc + sl + 0, f8.amax + sl + 0, s));
      else RC(vdk_layernorm_fwd(xin, D, T, D, params + b.n1w, params + b.n1b, d.eps, h1, D, VDK_BF16, mean1, rstd1, s));
      RC(gemm8(s, f8, h1, sl + 0, 0, f8.w8 + b.qkv_w, sl + 4, D, qkv, 3 * D, T, 3 * D, D, VDK_BF16, params + b.qkv_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0, fl ? f8.a8 : nullptr));
      RC(vdk_attention_fwd(qkv, 3 * D, o, D, lse, d.B, d.N, d.H, 64, scale, s));
      RC(gemm8(s, f8, o, sl + 1, 0, f8.w8 + b.proj_w, sl + 5, D, xmid, D, T, D, D, VDK_F32, params + b.proj_b, xin, D, VDK_ACT_NONE, nullptr, 0));
      if (fl) RC(vdk_layernorm_fwd_q8(xmid, D, T, D, params + b.n2w, params + b.n2b, d.eps, h2, D, mean2, rstd2, f8.a8, D, 0, f8.sc + sl + 2, f8.amax + sl + 2, s));
      else RC(vdk_layernorm_fwd(xmid, D, T, D, params + b.n2w, params + b.n2b, d.eps, h2, D, VDK_BF16, mean2, rstd2, s));
      RC(gemm8(s, f8, h2, sl + 2, 0, f8.w8 + b.fc1_w, sl + 6, D, g, M, T, M, D, VDK_BF16, params + b.fc1_b, nullptr, 0, VDK_ACT_GELU, u, M, fl ? f8.a8 : nullptr, fq ? sl + 3 : -1, 0));
      RC(gemm8(s, f8, g, sl + 3, 0, f8.w8 + b.fc2_w, sl + 7, M, xout, D, T, D, M, VDK_F32, params + b.fc2_b, xmid, D, VDK_ACT_NONE, nullptr, 0, fq ? f8.a8b : nullptr));
      continue;
    }
    RC(vdk_layernorm_fwd(xin, D, T, D, params + b.n1w, params + b.n1b, d.eps, h1, D, DT16, mean1, rstd1, s));
    RC(gemm(s, h1, D, wb + b.qkv_w, D, qkv, 3 * D, T, 3 * D, D, DT16, params + b.qkv_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));
    RC(vdk_attention_fwd_dt(qkv, 3 * D, o, D, lse, d.B, d.N, d.H, 64, scale, DT16, s));
    RC(gemm(s, o, D, wb + b.proj_w, D, xmid, D, T, D, D, VDK_F32, params + b.proj_b, xin, D, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));
    // x = x + fc2(gelu(fc1(norm2(x))))
    RC(vdk_layernorm_fwd(xmid, D, T, D, params + b.n2w, params + b.n2b, d.eps, h2, D, DT16, mean2, rstd2, s));
    RC(gemm(s, h2, D, wb + b.fc1_w, D, g, M, T, M, D, DT16, params + b.fc1_b, nullptr, 0, ACT_FC1, u, M, 1, 0, nullptr, 0));
    RC(gemm(s, g, M, wb + b.fc2_w, M, xout, D, T, D, M, VDK_F32, params + b.fc2_b, xmid, D, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));
  }
  float* xl = X + (size_t)(2 * d.L) * XS;
  float* meanf = stats + (size_t)d.L * 4 * T; float* rstdf = meanf + T;
  if (d.C == 0) {
    // feature mode (timm num_classes=0, global_pool=''): `logits` receives norm(x) for ALL tokens, f32 [B*N, D]
    RC(vdk_layernorm_fwd(xl, D, T, D, params + p.norm_w, params + p.norm_b, d.eps, logits, D, VDK_F32, meanf, rstdf, s));
    return VDK_OK;
  }
  // final norm on the class-token rows only (LayerNorm is per token; global_pool='token' keeps token 0), then the head
  bf16_t* hf = (bf16_t*)(base + w.hf);
  RC(vdk_layernorm_fwd(xl, (int64_t)d.N * D, d.B, D, params + p.norm_w, params + p.norm_b, d.eps, hf, D, DT16, meanf, rstdf, s));
  RC(gemm(s, hf, D, wb + p.head_w, D, logits, d.Cp, d.B, d.Cp, D, VDK_F32, params + p.head_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));
  return VDK_OK;
}

// wgrad of one Linear: dW[out,in] = dY^T X  and db = colsum(dY); dY bf16 [rows, out], X bf16 [rows, in].
// Fast path (rows % 64 == 0): the TN LDS-DMA GEMM reads dY and X as they lie (hardware transpose reads) and a vectorised
// column-sum kernel produces db.  Otherwise (ragged token counts): explicit bf16 transposes feed the NT kernel and db
// rides along with the dY transpose.
static int linear_wgrad(hipStream_t s, const VitDims& d, const WsPlan& w, char* base, const bf16_t* dY, int64_t lddy, const bf16_t* Xa,
                        int64_t ldx, int rows, int rows_pad, int out, int in, float* dW, float* db, int dy_row_group) {
  // (fp16 operands: the token-row remap of dY -- the patch embedding's weight gradient reads the rows of the [B, 1 + np, D] gradient minus the cls rows -- is a feature of
  //  the eight-wave bf16 TN kernel only; that one GEMM per step takes the transposing path below: two passes over [B np, D] tensors, ~0.2 % of a ViT-B/16 step)
  if ((rows % 64) == 0 && (out % 8) == 0 && (in % 8) == 0 && out >= 8 && in >= 8 && !(t_opf && dy_row_group != 0)) {
    const int sk = wgrad_splitk_tn(out, in, rows);
    VdkGemmDesc g = {};
    g.A = dY; g.lda = lddy; g.B = Xa; g.ldb = ldx; g.C = dW; g.ldc = in; g.M = out; g.N = in; g.K = rows; g.c_dtype = VDK_F32;
    g.alpha = 1.0f; g.splitk = sk; g.trans = 1; g.a_row_group = dy_row_group; g.ab_dtype = DT16;
    RC(vdk_gemm_bf16_nt(&g, base + w.slabs, w.slabs_bytes, s));
    if (db && dy_row_group == 0) RC(vdk_colsum_16(dY, lddy, rows, out, db, base + w.csws + 4 * w.csws_bytes, w.csws_bytes, t_opf, s));
    return VDK_OK;
  }
  bf16_t* tA = (bf16_t*)(base + w.tA); bf16_t* tB = (bf16_t*)(base + w.tB);
  float* csp = (db && dy_row_group == 0) ? (float*)(base + w.csws + 4 * w.csws_bytes) : nullptr;   // bias gradient rides along with the dY transpose
  RC(vdk_transpose_16(dY, lddy, rows, out, tA, rows_pad, rows_pad, dy_row_group, csp, t_opf, s));
  RC(vdk_transpose_16(Xa, ldx, rows, in, tB, rows_pad, rows_pad, 0, nullptr, t_opf, s));
  const int sk = wgrad_splitk(out, in, rows_pad);
  RC(gemm(s, tA, rows_pad, tB, rows_pad, dW, in, out, in, rows_pad, VDK_F32, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, sk, 0, base + w.slabs,
          w.slabs_bytes));
  if (csp) RC(vdk_reduce_rows_f32(csp, out, (rows_pad + 63) / 64, out, db, 1.0f, s));
  return VDK_OK;
}

// dgrad GEMM  dX[rows, in] = act'(dY[rows, out] . Wt[in, out]^T)  that also delivers db = colsum(dY) for the same Linear: when the 256x256 NT kernel
// serves the problem the column sums are a by-product of its A tiles (no extra pass over dY); returns 1 in *fused then, else the caller runs vdk_colsum_bf16.
// The partial sums land in column-sum buffer `slot` (0..3) and their reduction is appended to `jobs` for the block's one batched launch.
// dbx (optional): bias gradient of the Linear whose dY is this GEMM's OUTPUT dX (the previous Linear in the backward order) = column sums of the stored bf16 dX,
// accumulated in the epilogue (c_colsum by-product, partials in buffer `slot + 4`... the caller passes a distinct slot); *fusedx says whether that happened.
// qkv.bias gradient as its own column-sum pass (see the call site); VDK_VIT_QKVB_PASS=0 restores the GEMM by-product form
static bool qkvb_pass(int T, int N, size_t csws_bytes) {
  static const int env = getenv("VDK_VIT_QKVB_PASS") ? atoi(getenv("VDK_VIT_QKVB_PASS")) : -1;
  if (env == 0) return false;
  size_t need = 0;
  if (vdk_colsum_bf16_workspace_bytes(T, N, &need) != VDK_OK || need > csws_bytes) return false;
  return true;      // measured on one box (two runs each): 37.47 -> 37.24 ms per ViT-B/16 step, GEMM average 184.8 -> 179.4 us per launch
}

static int dgrad_with_bias(hipStream_t s, const WsPlan& w, char* base, const bf16_t* dY, int64_t lddy, const bf16_t* Wt, int64_t ldw, void* dX, int64_t lddx, int rows,
                           int in, int out, int act, void* aux, int64_t ldaux, float* db, int* fused, int slot, VdkReduceJob* jobs, int* njobs,
                           float* dbx = nullptr, int* fusedx = nullptr, int slotx = 0) {
  const int prow = db ? vdk_gemm_a_colsum_rows(rows, in, out) : 0;
  *fused = (db && !t_opf && prow > 0 && (size_t)prow * out * 4 <= w.csws_bytes) ? 1 : 0;      // (the A-tile column sums are a by-product of the eight-wave bf16 kernel only)
  const int xrow = dbx ? vdk_gemm_c_colsum_rows(rows, in, out) : 0;
  const int fx = (dbx && !*fused && xrow > 0 && (size_t)xrow * in * 4 <= w.csws_bytes) ? 1 : 0;
  if (fusedx) *fusedx = fx;
  float* const part = (float*)(base + w.csws + (size_t)slot * w.csws_bytes);
  float* const partx = (float*)(base + w.csws + (size_t)slotx * w.csws_bytes);
  VdkGemmDesc g = {};
  g.A = dY; g.lda = lddy; g.B = Wt; g.ldb = ldw; g.C = dX; g.ldc = lddx; g.M = rows; g.N = in; g.K = out; g.c_dtype = DT16; g.act = act; g.aux = aux;
  g.ldaux = ldaux; g.alpha = 1.0f; g.splitk = 1; g.ab_dtype = DT16;
  if (*fused) g.a_colsum = part;
  if (fx) g.c_colsum = partx;
  RC(vdk_gemm_bf16_nt(&g, nullptr, 0, s));
  if (*fused) jobs[(*njobs)++] = VdkReduceJob{part, (long)out, prow, (long)out, db, 1.0f};
  if (fx) jobs[(*njobs)++] = VdkReduceJob{partx, (long)in, xrow, (long)in, dbx, 1.0f};
  return VDK_OK;
}

// dlogits: bf16 [B, Cp] (what vdk_softmax_ce writes, padding columns zero); in feature mode (num_classes = 0): f32 [B*N, D].  grads: flat fp32, param layout,
// fully overwritten (zero_grad semantics).  on_ready(user, offset, numel) is called on the host right after the
// kernels producing grads[offset, offset+numel) have been enqueued (reverse layer order) so that a data-parallel
// caller can start that bucket's all-reduce on another stream; may be NULL.
int vdk_vit_backward(const VdkVitConfig* cfg, const void* dlogits, const float* params, const void* wb16, const void* wt16, void* ws,
                     size_t ws_bytes, float* grads, vdk_grad_ready_fn on_ready, void* user, void* stream_, void* side_stream_) {
  hipStream_t s = (hipStream_t)stream_;
  // s2: where the weight-gradient GEMMs + bias column sums go.  They are compute-bound with tiny outputs and independent of
  // the dgrad chain (whose short-K GEMMs have HBM-bound epilogues): interleaving the two across CUs overlaps MFMA with HBM.
  hipStream_t s2 = side_stream_ ? (hipStream_t)side_stream_ : s;
  VitDims d; RC(vit_dims(cfg, &d));
  PLayout p; RC(vit_layout(d, &p));
  WsPlan w; RC(vit_plan(d, &w));
  if (!dlogits || !params || !wb16 || !wt16 || !ws || !grads) return vdk_fail(VDK_EINVAL, "vdk_vit_backward: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_vit_backward: workspace too small");
  t_opf = d.opf;
  char* base = (char*)ws;
  const bf16_t* wt = (const bf16_t*)wt16;
  const int T = d.T, D = d.D, M = d.M;
  float* X = (float*)(base + w.X);
  float* stats = (float*)(base + w.stats);
  const size_t XS = (size_t)T * D;
  float* dxa = (float*)(base + w.dxa); float* dxm = (float*)(base + w.dxm);
  // bf16 gradient operands rotate over the layers (dxab: 3 buffers, the others: parity): the side stream may still be reading
  // layer l+1's copies while the main stream writes layer l's.  Main waits for "side finished layer l+2" before it starts layer l.
#define DXAB(l) ((bf16_t*)(base + w.dxab[((l) + 3) % 3]))
#define DXMB(l) ((bf16_t*)(base + w.dxmb[((l) + 2) & 1]))
#define DU(l) ((bf16_t*)(base + w.du[((l) + 2) & 1]))
#define DQKV(l) ((bf16_t*)(base + w.dqkv[((l) + 2) & 1]))
  bf16_t* dsm = (bf16_t*)(base + w.dsm);
  float* dvec = (float*)(base + w.dvec);
  void* lnws = base + w.lnws;
  const size_t EV_SIDE_DONE = 8;   // slots [8, 8 + L + 2): side stream finished layer l (index l + 1; 0 = embeddings)
  size_t ev_p = EV_SIDE_DONE + d.L + 4;   // producer events, a fresh slot per use
  RC(ev_order(0, s, s2));           // the side stream starts after whatever precedes this call on the main stream

  // ---- head + final norm -------------------------------------------------------------------------
  bool last_fc2_bias_done = false;   // fc2.bias gradient of block l is produced with DXAB(l): by the final norm's backward (l = L-1) or by block l+1's norm1 backward
  if (d.C == 0) {
    // feature mode: `dlogits` is dL/d norm(x) for all tokens, f32 [B*N, D]
    float* xl = X + (size_t)(2 * d.L) * XS;
    float* meanf = stats + (size_t)d.L * 4 * T; float* rstdf = meanf + T;
    RC(vdk_layernorm_bwd_deferred(dlogits, D, VDK_F32, xl, D, meanf, rstdf, params + p.norm_w, nullptr, 0, T, D, dxa, D, DXAB(d.L - 1), D, grads + p.norm_w,
                                  grads + p.norm_b, lnws, w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf));
    if (on_ready) on_ready(user, p.norm_w, p.total - p.norm_w);
  } else {
    const bf16_t* dl = (const bf16_t*)dlogits;
    bf16_t* hf = (bf16_t*)(base + w.hf);
    RC(linear_wgrad(s, d, w, base, dl, d.Cp, hf, D, d.B, d.Bp, d.Cp, D, grads + p.head_w, grads + p.head_b, 0));
    bf16_t* dhf = (bf16_t*)(base + w.dhf);
    RC(gemm(s, dl, d.Cp, wt + p.headT, d.Cp, dhf, D, d.B, D, d.Cp, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));
    if (hipMemsetAsync(dxa, 0, XS * 4, s) != hipSuccess || hipMemsetAsync(DXAB(d.L - 1), 0, XS * 2, s) != hipSuccess)
      return vdk_fail(VDK_ELAUNCH, "vdk_vit_backward: memset failed");
    float* xl = X + (size_t)(2 * d.L) * XS;
    float* meanf = stats + (size_t)d.L * 4 * T; float* rstdf = meanf + T;
    if (s2 == s && D <= 1024) {   // db of the last block's fc2 = column sums of DXAB(L-1) (only the cls rows are non-zero): by-product of this kernel
      VdkReduceJob fj[2];
      RC(vdk_layernorm_bwd_deferred(dhf, D, DT16, xl, (int64_t)d.N * D, meanf, rstdf, params + p.norm_w, nullptr, 0, d.B, D, dxa, (int64_t)d.N * D, DXAB(d.L - 1),
                                    (int64_t)d.N * D, grads + p.norm_w, grads + p.norm_b, lnws, w.lnws_bytes, s, &fj[0], grads + p.blk[d.L - 1].fc2_b, &fj[1]));
      if (fj[0].in) RC(vdk_reduce_rows_batch(fj, 2, s)); else RC(vdk_reduce_rows_batch(fj + 1, 1, s));
      last_fc2_bias_done = true;
    } else
    RC(vdk_layernorm_bwd(dhf, D, DT16, xl, (int64_t)d.N * D, meanf, rstdf, params + p.norm_w, nullptr, 0, d.B, D, dxa, (int64_t)d.N * D, DXAB(d.L - 1),
                         (int64_t)d.N * D, grads + p.norm_w, grads + p.norm_b, lnws, w.lnws_bytes, s));
    if (on_ready) on_ready(user, p.norm_w, p.total - p.norm_w);
  }
  // ---- blocks, last to first ------------------------------------------------------------------------
  bool fc2_bias_from_norm1 = false, dxab8_ready = false;
  F8 f8; RC(f8_init(cfg, d, p, &f8, (unsigned char*)(base + w.a8), (unsigned char*)(base + w.a8b)));
  for (int l = d.L - 1; l >= 0; --l) {
    const PLayout::Blk& b = p.blk[l];
    float* xin = X + (size_t)(2 * l) * XS; float* xmid = xin + XS;
    float* mean1 = stats + (size_t)l * 4 * T; float* rstd1 = mean1 + T; float* mean2 = rstd1 + T; float* rstd2 = mean2 + T;
    bf16_t* h1 = (bf16_t*)(base + w.h1 + l * w.s_h); bf16_t* qkv = (bf16_t*)(base + w.qkv + l * w.s_qkv);
    float* lse = (float*)(base + w.lse + l * w.s_lse); bf16_t* o = (bf16_t*)(base + w.o + l * w.s_h);
    bf16_t* h2 = (bf16_t*)(base + w.h2 + l * w.s_h); bf16_t* u = (bf16_t*)(base + w.u + l * w.s_u); bf16_t* g = (bf16_t*)(base + w.g + l * w.s_u);
    bf16_t* dxab = DXAB(l); bf16_t* dxmb = DXMB(l); bf16_t* du = DU(l); bf16_t* dqkv = DQKV(l);
    // WAR: this layer overwrites the parity buffers the side stream read two layers ago
    if (s2 != s && l + 2 <= d.L - 1) { hipEvent_t e; RC(ev_get(EV_SIDE_DONE + l + 3, &e)); if (hipStreamWaitEvent(s, e, 0) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vit: wait failed"); }
    // Every Linear: dgrad GEMM on the main stream (its bias gradient is a by-product of the A tiles when the 256x256 kernel serves it), weight gradient
    // straight from dY / X on the side stream.  The column sums live in csws, which the side stream's fallback paths also use: fused mode is main-stream only.
    int fz = 0;
    const bool one_stream = (s2 == s);
    VdkReduceJob jobs[8]; int nj = 0;          // this block's small reductions (LayerNorm dgamma | dbeta, Linear bias gradients): one launch at its end
    char* const lnws0 = base + w.lnws; char* const lnws1 = lnws0 + w.lnws_bytes;
    // MLP branch: dxa / dxab hold dL/dx_out
    RC(ev_order(ev_p++, s, s2));
    if (f8.mode) {
      // fp8 operands (e5m2 gradient x e4m3 transposed weight) for the two input-gradient GEMMs; the weight gradients stay bf16 TN GEMMs from the same bf16 tensors.
      // fc2.bias comes with DXAB(l) when the norm backward above produced it; fc1.bias is a column-sum pass over du (the fp8 kernel has no by-products).
      const int sl = 12 * l;
      const bool have_fc2b = one_stream && ((l == d.L - 1) ? last_fc2_bias_done : fc2_bias_from_norm1);
      const bool fq = f8_fused(f8) && (M % 64) == 0;                 // du's e5m2 copy comes out of the dGELU epilogue
      // fc1.bias = column sums of du, accumulated by the dGELU epilogue that stores it (c_colsum, as in the bf16 path) -- the column-sum pass over the [T, 4D] tensor goes
      const int xrow = one_stream ? 2 * ((T + 255) / 256) : 0;
      const bool fo = xrow > 0 && (size_t)xrow * M * 4 <= w.csws_bytes;
      float* const partx = (float*)(base + w.csws + (size_t)1 * w.csws_bytes);
      RC(gemm8(s, f8, dxab, sl + 8, 1, f8.wt8 + p.blkT[l].fc2, sl + 7, D, du, M, T, M, D, VDK_BF16, nullptr, nullptr, 0, VDK_ACT_DGELU, u, M, dxab8_ready ? f8.a8 : nullptr,
               fq ? sl + 9 : -1, 1, fo ? partx : nullptr));   // du (DXAB(l)'s e5m2 copy is in the operand scratch when the norm1 backward of the block above wrote it)
      if (fo) jobs[nj++] = VdkReduceJob{partx, (long)M, xrow, (long)M, grads + b.fc1_b, 1.0f};
      RC(linear_wgrad(s2, d, w, base, dxab, D, g, M, T, d.Tp, D, M, grads + b.fc2_w, have_fc2b ? nullptr : grads + b.fc2_b, 0));
      RC(gemm8(s, f8, du, sl + 9, 1, f8.wt8 + p.blkT[l].fc1, sl + 6, M, dsm, D, T, D, M, VDK_BF16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, fq ? f8.a8b : nullptr));      // dh2
      RC(linear_wgrad(s2, d, w, base, du, M, h2, D, T, d.Tp, M, D, grads + b.fc1_w, fo ? nullptr : grads + b.fc1_b, 0));
    } else if (one_stream) {
      // Bias gradients ride with the PRODUCER of each dY (it sums what it stores): fc2.bias with DXAB(l) (norm backward of the block above), fc1.bias with du
      // (dGELU epilogue below), proj.bias with dxmb (norm2 backward below); only qkv.bias still comes from the A tiles of the dh1 GEMM (dqkv is attention's output).
      int fx = 0;
      const bool have_fc2b = (l == d.L - 1) ? last_fc2_bias_done : fc2_bias_from_norm1;
      RC(dgrad_with_bias(s, w, base, dxab, D, wt + p.blkT[l].fc2, D, du, M, T, M, D, ACT_DFC2, u, M, have_fc2b ? nullptr : grads + b.fc2_b, &fz, 0, jobs, &nj,
                         grads + b.fc1_b, &fx, 1));   // du
      RC(linear_wgrad(s2, d, w, base, dxab, D, g, M, T, d.Tp, D, M, grads + b.fc2_w, (fz || have_fc2b) ? nullptr : grads + b.fc2_b, 0));
      if (fx) {
        RC(gemm(s, du, M, wt + p.blkT[l].fc1, M, dsm, D, T, D, M, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));   // dh2
        fz = 1;
      } else {
        RC(dgrad_with_bias(s, w, base, du, M, wt + p.blkT[l].fc1, M, dsm, D, T, D, M, VDK_ACT_NONE, nullptr, 0, grads + b.fc1_b, &fz, 1, jobs, &nj));   // dh2
      }
      RC(linear_wgrad(s2, d, w, base, du, M, h2, D, T, d.Tp, M, D, grads + b.fc1_w, fz ? nullptr : grads + b.fc1_b, 0));
    } else {
      RC(linear_wgrad(s2, d, w, base, dxab, D, g, M, T, d.Tp, D, M, grads + b.fc2_w, grads + b.fc2_b, 0));
      RC(gemm(s, dxab, D, wt + p.blkT[l].fc2, D, du, M, T, M, D, DT16, nullptr, nullptr, 0, ACT_DFC2, u, M, 1, 0, nullptr, 0));   // du
      RC(ev_order(ev_p++, s, s2));
      RC(linear_wgrad(s2, d, w, base, du, M, h2, D, T, d.Tp, M, D, grads + b.fc1_w, grads + b.fc1_b, 0));
      RC(gemm(s, du, M, wt + p.blkT[l].fc1, M, dsm, D, T, D, M, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));   // dh2
    }
    const bool ocs_ln = one_stream && D <= 1024;
    // fp16 operands: the residual-gradient stream travels in 16 bits.  Every norm backward writes dL/dx twice -- fp32 for the next norm backward's residual term, 16-bit
    // for the next GEMM -- and reads the fp32 one back: 310 of its 620 MB per launch at ViT-B/16 (the kernel streams at HBM rate, so bytes are its time).  With g16 the
    // 16-bit copy is the stream: the sum is formed in fp32 registers and rounded once per norm (24 roundings of 2^-11 down the trunk; measured against the fp32 oracle
    // in tests/test_fp16_operands.py and bench.py's parity block, bound 5e-3 on every gradient).  bf16 (8 mantissa bits) keeps the fp32 stream.
    static const bool g16_on = !(getenv("VDK_VIT_G16") && atoi(getenv("VDK_VIT_G16")) == 0);
    const bool g16 = g16_on && ocs_ln && !f8.mode && t_opf == VDK_OPF_F16 && D > 512 && D <= 1024;
    const bool lq = f8.mode && f8_fused(f8) && ocs_ln;               // the norm backward kernels write the e5m2 copies of dxmb / DXAB(l - 1) into the operand scratch
    const LnQ8 q8m = {f8.a8, (long)D, f8.sc + 12 * l + 10, f8.amax + 12 * l + 10, 1};
    if (g16) RC(vdk_layernorm_bwd_deferred(dsm, D, DT16, xmid, D, mean2, rstd2, params + b.n2w, nullptr, D, T, D, nullptr, D, dxmb, D, grads + b.n2w, grads + b.n2b, lnws0,
                                           w.lnws_bytes, s, &jobs[nj], grads + b.proj_b, &jobs[nj + 1], nullptr, 0, nullptr, nullptr, 1, dxab));
    else
    RC(vdk_layernorm_bwd_deferred(dsm, D, DT16, xmid, D, mean2, rstd2, params + b.n2w, dxa, D, T, D, dxm, D, dxmb, D, grads + b.n2w, grads + b.n2b, lnws0,
                                  w.lnws_bytes, s, &jobs[nj], ocs_ln ? grads + b.proj_b : nullptr, ocs_ln ? &jobs[nj + 1] : nullptr, lq ? &q8m : nullptr));
    nj += ocs_ln ? 2 : 1;
    // attention branch: dxm / dxmb hold dL/dx_mid
    RC(ev_order(ev_p++, s, s2));
    if (f8.mode) {
      RC(gemm8(s, f8, dxmb, 12 * l + 10, 1, f8.wt8 + p.blkT[l].proj, 12 * l + 5, D, dsm, D, T, D, D, VDK_BF16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, lq ? f8.a8 : nullptr));   // do
      RC(linear_wgrad(s2, d, w, base, dxmb, D, o, D, T, d.Tp, D, D, grads + b.proj_w, ocs_ln ? nullptr : grads + b.proj_b, 0));
    } else if (one_stream && ocs_ln) {
      RC(gemm(s, dxmb, D, wt + p.blkT[l].proj, D, dsm, D, T, D, D, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));   // do
      RC(linear_wgrad(s2, d, w, base, dxmb, D, o, D, T, d.Tp, D, D, grads + b.proj_w, nullptr, 0));
    } else if (one_stream) {
      RC(dgrad_with_bias(s, w, base, dxmb, D, wt + p.blkT[l].proj, D, dsm, D, T, D, D, VDK_ACT_NONE, nullptr, 0, grads + b.proj_b, &fz, 2, jobs, &nj));   // do
      RC(linear_wgrad(s2, d, w, base, dxmb, D, o, D, T, d.Tp, D, D, grads + b.proj_w, fz ? nullptr : grads + b.proj_b, 0));
    } else {
      RC(linear_wgrad(s2, d, w, base, dxmb, D, o, D, T, d.Tp, D, D, grads + b.proj_w, grads + b.proj_b, 0));
      RC(gemm(s, dxmb, D, wt + p.blkT[l].proj, D, dsm, D, T, D, D, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));   // do
    }
    // dqkv; with the one-pass small-N kernel the qkv.bias gradient's per-image partials (column sums of the dq | dk | dv rows it stores) come out of the same launch
    // (round 5: the separate pass over dqkv was 13 x 42 us per ViT-B/16 step); VDK_VIT_QKVB_ATTN=0 keeps that pass
    int32_t qkvb_done = 0;
    {
      static const bool fuse_on = !(getenv("VDK_VIT_QKVB_ATTN") && atoi(getenv("VDK_VIT_QKVB_ATTN")) == 0);
      float* csp = (fuse_on && !f8.mode && one_stream && (size_t)d.B * 3 * D * 4 <= w.csws_bytes) ? (float*)(base + w.csws + (size_t)3 * w.csws_bytes) : nullptr;
      RC(vdk_attention_bwd_cs(qkv, 3 * D, o, dsm, D, lse, dqkv, 3 * D, dvec, d.B, d.N, d.H, 64, 0.125f, DT16, csp, &qkvb_done, s));
      if (qkvb_done) { jobs[nj] = VdkReduceJob{csp, (long)3 * D, d.B, (long)3 * D, grads + b.qkv_b, 1.0f}; ++nj; }
    }
    RC(ev_order(ev_p++, s, s2));
    if (f8.mode) {
      // dqkv is attention's output: its column sums (qkv.bias) and its e5m2 copy (operand of the dh1 GEMM) come from ONE pass over it
      size_t csneed = 0; vdk_colsum_bf16_workspace_bytes(T, 3 * D, &csneed);
      const bool fcq = f8_fused(f8) && one_stream && csneed <= w.csws_bytes;
      if (fcq) {
        const LnQ8 q8d = {f8.a8, (long)3 * D, f8.sc + 12 * l + 11, f8.amax + 12 * l + 11, 1};
        RC(vdk_colsum_bf16_deferred(dqkv, 3 * D, T, 3 * D, grads + b.qkv_b, base + w.csws + (size_t)2 * w.csws_bytes, w.csws_bytes, s, &jobs[nj], &q8d)); ++nj;
      }
      RC(gemm8(s, f8, dqkv, 12 * l + 11, 1, f8.wt8 + p.blkT[l].qkv, 12 * l + 4, 3 * D, dsm, D, T, D, 3 * D, VDK_BF16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0,
               fcq ? f8.a8 : nullptr));   // dh1
      RC(linear_wgrad(s2, d, w, base, dqkv, 3 * D, h1, D, T, d.Tp, 3 * D, D, grads + b.qkv_w, fcq ? nullptr : grads + b.qkv_b, 0));
    } else if (one_stream && (qkvb_done || qkvb_pass(T, 3 * D, w.csws_bytes))) {
      // qkv.bias = column sums of dqkv (attention's output: no producing GEMM epilogue to ride on) as one pass over it, so that the dh1 GEMM is the plain
      // one-wave-per-SIMD kernel instead of the eight-wave kernel with the A-tile column-sum by-product (A/B on one box: see DESIGN.md)
      if (!qkvb_done) { RC(vdk_colsum_bf16_deferred(dqkv, 3 * D, T, 3 * D, grads + b.qkv_b, base + w.csws + (size_t)3 * w.csws_bytes, w.csws_bytes, s, &jobs[nj], nullptr, t_opf)); ++nj; }
      RC(gemm(s, dqkv, 3 * D, wt + p.blkT[l].qkv, 3 * D, dsm, D, T, D, 3 * D, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));   // dh1
      RC(linear_wgrad(s2, d, w, base, dqkv, 3 * D, h1, D, T, d.Tp, 3 * D, D, grads + b.qkv_w, nullptr, 0));
    } else if (one_stream) {
      RC(dgrad_with_bias(s, w, base, dqkv, 3 * D, wt + p.blkT[l].qkv, 3 * D, dsm, D, T, D, 3 * D, VDK_ACT_NONE, nullptr, 0, grads + b.qkv_b, &fz, 3, jobs, &nj));   // dh1
      RC(linear_wgrad(s2, d, w, base, dqkv, 3 * D, h1, D, T, d.Tp, 3 * D, D, grads + b.qkv_w, fz ? nullptr : grads + b.qkv_b, 0));
    } else {
      RC(linear_wgrad(s2, d, w, base, dqkv, 3 * D, h1, D, T, d.Tp, 3 * D, D, grads + b.qkv_w, grads + b.qkv_b, 0));
      RC(gemm(s, dqkv, 3 * D, wt + p.blkT[l].qkv, 3 * D, dsm, D, T, D, 3 * D, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, 0, nullptr, 0));  // dh1
    }
    const bool ocs_n1 = ocs_ln && l > 0;      // DXAB(l - 1) is dY of block l-1's fc2 (for l == 0 it feeds the patch embedding, whose bias comes from d pos_embed)
    const bool lq1 = lq && ocs_n1;
    const LnQ8 q8a = {f8.a8, (long)D, f8.sc + 12 * (l - 1) + 8, f8.amax + 12 * (l - 1) + 8, 1};
    if (g16 && l > 0) RC(vdk_layernorm_bwd_deferred(dsm, D, DT16, xin, D, mean1, rstd1, params + b.n1w, nullptr, D, T, D, nullptr, D, DXAB(l - 1), D, grads + b.n1w, grads + b.n1b, lnws1,
                                                    w.lnws_bytes, s, &jobs[nj], grads + p.blk[l - 1].fc2_b, &jobs[nj + 1], nullptr, 0, nullptr, nullptr, 1, dxmb));
    else if (g16) {      // block 0: the embeddings' gradients want the fp32 tensor (d pos_embed sums it over the batch); the column-sum by-product has no taker here
      float* const dummy_cs = (float*)(base + w.csws);      // (scratch: the sums are not used)
      RC(vdk_layernorm_bwd_deferred(dsm, D, DT16, xin, D, mean1, rstd1, params + b.n1w, nullptr, D, T, D, dxa, D, DXAB(l - 1), D, grads + b.n1w, grads + b.n1b, lnws1,
                                    w.lnws_bytes, s, &jobs[nj], dummy_cs, &jobs[nj + 1], nullptr, 0, nullptr, nullptr, 1, dxmb));
    } else
    RC(vdk_layernorm_bwd_deferred(dsm, D, DT16, xin, D, mean1, rstd1, params + b.n1w, dxm, D, T, D, dxa, D, DXAB(l - 1), D, grads + b.n1w, grads + b.n1b, lnws1,
                                  w.lnws_bytes, s, &jobs[nj], ocs_n1 ? grads + p.blk[l - 1].fc2_b : nullptr, ocs_n1 ? &jobs[nj + 1] : nullptr, lq1 ? &q8a : nullptr));
    dxab8_ready = lq1;
    nj += ocs_n1 ? 2 : 1;
    fc2_bias_from_norm1 = ocs_n1;
    RC(vdk_reduce_rows_batch(jobs, nj, s));
    if (s2 != s) { hipEvent_t e; RC(ev_get(EV_SIDE_DONE + l + 1, &e)); if (hipEventRecord(e, s2) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vit: record failed"); }
    if (on_ready) {
      if (s2 != s) { hipEvent_t e; RC(ev_get(EV_SIDE_DONE + l + 1, &e)); if (hipStreamWaitEvent(s, e, 0) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vit: wait failed"); }
      on_ready(user, b.n1w, (l + 1 < d.L ? p.blk[l + 1].n1w : p.norm_w) - b.n1w);
    }
  }
  // ---- embeddings -------------------------------------------------------------------------------------
  {
    float* dposall = (float*)(base + w.dposall);
    const float* dx0 = dxa;      // dL/d(embedding)
    if (d.pre) {      // norm_pre backward: dxa = dL/d(block 0's input) -> dxm = dL/d(embedding), its 16-bit copy (the patch embedding's dY) over DXAB(-1), norm_pre's own gradients
      const float* ps = (const float*)(base + w.pstats);
      RC(vdk_layernorm_bwd_deferred(dxa, D, VDK_F32, (const float*)(base + w.xe), D, ps, ps + T, params + p.npre_w, nullptr, 0, T, D, dxm, D, DXAB(-1), D, grads + p.npre_w,
                                    grads + p.npre_b, lnws, w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf));
      dx0 = dxm;
    }
    // d pos_embed[n] = sum_b dx0[b, n];  d cls = d pos_embed[0];  d patch bias = sum_{n >= 1} d pos_embed[n] (pre_norm: there is no such bias, its slot's gradient stays zero)
    RC(vdk_reduce_rows_f32(dx0, (int64_t)d.N * D, d.B, (int64_t)d.N * D, grads + p.pos, 1.0f, s));
    if (d.cls && hipMemcpyAsync(grads + p.cls, grads + p.pos, (size_t)D * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return vdk_fail(VDK_ELAUNCH, "vdk_vit_backward: memcpy failed");
    if (!d.pre) RC(vdk_reduce_rows_f32(grads + p.pos + (size_t)d.cls * D, D, d.np, D, grads + p.pe_b, 1.0f, s));
    else if (hipMemsetAsync(grads + p.pe_b, 0, (size_t)D * 4, s) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_vit_backward: memset failed");
    (void)dposall;
    bf16_t* patches = (bf16_t*)(base + w.patches);
    const int rows = d.B * d.np, rows_pad = (int)up(rows, 64);
    RC(ev_order(ev_p++, s, s2));
    float* dwpe = d.Kraw != d.Kpe ? (float*)(base + w.dwpe) : grads + p.pe_w;
    RC(linear_wgrad(s2, d, w, base, DXAB(-1), D, patches, d.Kpe, rows, rows_pad, D, d.Kpe, dwpe, nullptr, d.cls ? d.np : 0));
    if (d.Kraw != d.Kpe && hipMemcpy2DAsync(grads + p.pe_w, (size_t)d.Kraw * 4, dwpe, (size_t)d.Kpe * 4, (size_t)d.Kraw * 4, D, hipMemcpyDeviceToDevice, s2) != hipSuccess)
      return vdk_fail(VDK_ELAUNCH, "vdk_vit_backward: memcpy2d failed");   // drop the padding columns
    RC(ev_order(ev_p++, s2, s));      // join: everything the side stream produced is ordered before what follows on the main stream
    if (on_ready) on_ready(user, 0, p.blk[0].n1w);
  }
#undef DXAB
#undef DXMB
#undef DU
#undef DQKV
  return vdk_check_launch("vdk_vit_backward");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// PRECISE forward (evaluation / embedding extraction): the same network with every contraction on the fp32 MFMA
// (csrc/gemm_f32.hip), fp32 activations throughout, attention with materialised fp32 scores.  Reads the fp32 master weights
// directly; nothing is saved for a backward.  Logits / tokens agree with the reference's PyTorch-CPU fp32 path to ~1e-6.
struct WsF32 { size_t total, patches, xa, xb, h, qkv, S, o, u; int Np; };
static void vit_plan_f32(const VitDims& d, WsF32* w) {
  size_t cur = 0;
  const size_t T = d.T, D = d.D;
  w->Np = (int)up(d.N, 4);
  w->patches = w_take(cur, (size_t)d.B * d.np * d.Kpe * 4);
  w->xa = w_take(cur, T * D * 4); w->xb = w_take(cur, T * D * 4); w->h = w_take(cur, T * D * 4);
  w->qkv = w_take(cur, (T + 8) * 3 * D * 4);                      // + pad rows: the P V contraction reads K = Np >= N value rows
  w->S = w_take(cur, (size_t)d.B * d.H * d.N * w->Np * 4);
  w->o = w_take(cur, T * D * 4);
  w->u = w_take(cur, T * (size_t)d.M * 4);
  w->total = cur;
}
static int gemm32(hipStream_t s, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                  const float* res, int64_t ldr, int act) {
  VdkGemmF32Desc g = {};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.residual = res; g.ldr = ldr; g.act = act;
  g.alpha = 1.0f;
  return vdk_gemm_f32_nt(&g, s);
}

extern "C" {

int vdk_vit_workspace_f32_bytes(const VdkVitConfig* cfg, size_t* bytes) {
  VitDims d; RC(vit_dims(cfg, &d));
  if (!bytes) return vdk_fail(VDK_EINVAL, "null");
  WsF32 w; vit_plan_f32(d, &w);
  *bytes = w.total;
  return VDK_OK;
}

// x f32 [B, Cin, img, img] -> logits f32 [B, Cp] (num_classes > 0) or final-normed tokens f32 [B*N, D] (num_classes == 0)
int vdk_vit_forward_f32(const VdkVitConfig* cfg, const float* x, const float* params, void* ws, size_t ws_bytes, float* logits, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  VitDims d; RC(vit_dims(cfg, &d));
  if (d.Kraw != d.Kpe) return vdk_fail(VDK_EUNSUPPORTED, "vdk_vit_forward_f32: in_chans*patch*patch must be a multiple of 8 on the fp32 path");
  PLayout p; RC(vit_layout(d, &p));
  WsF32 w; vit_plan_f32(d, &w);
  if (!x || !params || !ws || !logits) return vdk_fail(VDK_EINVAL, "vdk_vit_forward_f32: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_vit_forward_f32: workspace too small");
  if ((long)d.B * d.H > 65535) return vdk_fail(VDK_EUNSUPPORTED, "vdk_vit_forward_f32: batch * heads <= 65535");
  char* base = (char*)ws;
  const int T = d.T, D = d.D, M = d.M, N = d.N, Np = w.Np;
  float* patches = (float*)(base + w.patches);
  float* xa = (float*)(base + w.xa); float* xb = (float*)(base + w.xb); float* h = (float*)(base + w.h);
  float* qkv = (float*)(base + w.qkv); float* S = (float*)(base + w.S); float* o = (float*)(base + w.o); float* u = (float*)(base + w.u);
  if (hipMemsetAsync(qkv + (size_t)T * 3 * D, 0, (size_t)8 * 3 * D * 4, s) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_vit_forward_f32: memset failed");
  // patch embedding per image (batch index = image): rows 1..N-1 of its token block, + pos_embed; then the cls rows
  RC(vdk_patchify_f32(x, d.B, d.Cin, d.img, d.img, d.ps, patches, s));
  {
    VdkGemmF32Desc g = {};
    g.A = patches; g.lda = d.Kpe; g.B = params + p.pe_w; g.ldb = d.Kpe; g.C = xa + (size_t)d.cls * D; g.ldc = D; g.M = d.np; g.N = D; g.K = d.Kpe;
    g.bias = params + p.pe_b; g.residual = params + p.pos + (size_t)d.cls * D; g.ldr = D; g.alpha = 1.0f;
    g.batch1 = d.B; g.batch2 = 1; g.sa1 = (int64_t)d.np * d.Kpe; g.sc1 = (int64_t)N * D;
    if (d.B > 65535) return vdk_fail(VDK_EUNSUPPORTED, "vdk_vit_forward_f32: batch <= 65535");
    RC(vdk_gemm_f32_nt(&g, s));
  }
  if (d.cls) RC(vdk_cls_rows(xa, (int64_t)N * D, d.B, D, params + p.cls, params + p.pos, s));
  if (d.pre) {      // norm_pre (the bias slot of the patch embedding holds zeros in this mode)
    RC(vdk_layernorm_fwd(xa, D, T, D, params + p.npre_w, params + p.npre_b, d.eps, xb, D, VDK_F32, nullptr, nullptr, s));
    float* t = xa; xa = xb; xb = t;
  }
  for (int l = 0; l < d.L; ++l) {
    const PLayout::Blk& b = p.blk[l];
    RC(vdk_layernorm_fwd(xa, D, T, D, params + b.n1w, params + b.n1b, d.eps, h, D, VDK_F32, nullptr, nullptr, s));
    RC(gemm32(s, h, D, params + b.qkv_w, D, qkv, 3 * D, T, 3 * D, D, params + b.qkv_b, nullptr, 0, VDK_ACT_NONE));
    {   // S[b, head] = Q K^T (scale applied inside the softmax), then P = softmax(S / 8), then O = P V
      VdkGemmF32Desc g = {};
      g.A = qkv; g.lda = 3 * D; g.B = qkv + D; g.ldb = 3 * D; g.C = S; g.ldc = Np; g.M = N; g.N = N; g.K = 64; g.alpha = 1.0f;
      g.batch1 = d.B; g.batch2 = d.H; g.sa1 = (int64_t)N * 3 * D; g.sa2 = 64; g.sb1 = g.sa1; g.sb2 = 64; g.sc1 = (int64_t)d.H * N * Np; g.sc2 = (int64_t)N * Np;
      RC(vdk_gemm_f32_nt(&g, s));
      RC(vdk_softmax_rows_f32(S, Np, (int64_t)d.B * d.H * N, N, 0.125f, s));
      VdkGemmF32Desc v = {};
      v.A = S; v.lda = Np; v.B = qkv + 2 * D; v.ldb = 3 * D; v.b_kmajor = 1; v.C = o; v.ldc = D; v.M = N; v.N = 64; v.K = Np; v.alpha = 1.0f;
      v.batch1 = d.B; v.batch2 = d.H; v.sa1 = g.sc1; v.sa2 = g.sc2; v.sb1 = (int64_t)N * 3 * D; v.sb2 = 64; v.sc1 = (int64_t)N * D; v.sc2 = 64;
      RC(vdk_gemm_f32_nt(&v, s));
    }
    RC(gemm32(s, o, D, params + b.proj_w, D, xb, D, T, D, D, params + b.proj_b, xa, D, VDK_ACT_NONE));
    RC(vdk_layernorm_fwd(xb, D, T, D, params + b.n2w, params + b.n2b, d.eps, h, D, VDK_F32, nullptr, nullptr, s));
    RC(gemm32(s, h, D, params + b.fc1_w, D, u, M, T, M, D, params + b.fc1_b, nullptr, 0, VDK_ACT_GELU));
    RC(gemm32(s, u, M, params + b.fc2_w, M, xa, D, T, D, M, params + b.fc2_b, xb, D, VDK_ACT_NONE));
  }
  if (d.C == 0) {
    RC(vdk_layernorm_fwd(xa, D, T, D, params + p.norm_w, params + p.norm_b, d.eps, logits, D, VDK_F32, nullptr, nullptr, s));
    return vdk_check_launch("vdk_vit_forward_f32");
  }
  RC(vdk_layernorm_fwd(xa, (int64_t)N * D, d.B, D, params + p.norm_w, params + p.norm_b, d.eps, h, D, VDK_F32, nullptr, nullptr, s));
  RC(gemm32(s, h, D, params + p.head_w, D, logits, d.Cp, d.B, d.Cp, D, params + p.head_b, nullptr, 0, VDK_ACT_NONE));
  return vdk_check_launch("vdk_vit_forward_f32");
}

}  // extern "C"

// convnext_engine.hip — native forward/backward of timm ConvNeXt in feature mode (num_classes=0, global_pool=''), the CNN backbone
// the reference's face / CBIR path builds at models/faceX/backbone/timm_wrapper.py:16-21 (`convnext_base` in configs/faceX/cbir.yaml:4-8).
// Semantics restated from timm 0.9.16 (un-vendored; see oracle/convnext_ref.py, pinned against transformers.ConvNextModel):
//   stem Conv 4x4/4 + LayerNorm2d -> 4 stages [ (LayerNorm2d + Conv 2x2/2 for stages 1-3) + depth x ConvNeXtBlock ] -> head.norm (LayerNorm2d)
//   ConvNeXtBlock: x + gamma * fc2(GELU(fc1(LayerNorm(dwconv7x7(x)))))
//
// MI355X design: activations are NHWC rows (f32 residual stream, bf16 GEMM operands) exactly like the transformer's token rows, so
// the stem (patchify), the downsamples (space-to-depth) and the pointwise convs all run on the LDS-DMA MFMA GEMM with fused
// bias / GELU / residual epilogues, LayerNorm2d is the row LayerNorm kernel, and only the depthwise 7x7 (csrc/conv.hip) is new.
// The layer scale is folded into fc2 at weight-refresh time (W2' = gamma (.) W2), so it costs nothing on the activation path; its
// gradient comes from weight-sized tensors (vdk_layerscale_grad).  One C call per forward, one per backward; no allocation, no sync.
//
// Operand format (VdkConvNextConfig.operand): bf16 (above), or fp16 -- 8x smaller operand rounding, what brings the embeddings of the face / CBIR path inside 1e-3 of the
// reference's fp32 arithmetic (engine/procedure/train.py:217-227 runs that loop without autocast) at the bf16 step's speed; gradients then carry GradScaler's loss scale
// (train.py:205-211).  fp16 cannot hold gamma (.) W2 at timm's gamma = 1e-6, so in that mode fc2 multiplies the PLAIN weight and applies gamma in its epilogue
// (VdkGemmDesc.col_scale), and the backward runs the branch at a per-block power-of-two scale r (conv.hip: cn_prep_batch_kernel): du' = r du, dh' = r dh in fp16,
// LayerNorm backward multiplies 1 / r back in (fp32), the fc1 gradients are rescaled in place.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_gemm.h"

extern "C" {
int vdk_gemm_bf16_nt(const VdkGemmDesc*, void*, size_t, void*);
int vdk_layernorm_fwd(const float*, int64_t, int32_t, int32_t, const float*, const float*, float, void*, int64_t, int32_t, float*, float*, void*);
int vdk_layernorm_bwd_workspace_bytes(int32_t, int32_t, size_t*);
int vdk_layernorm_bwd(const void*, int64_t, int32_t, const float*, int64_t, const float*, const float*, const float*, const float*, int64_t, int32_t, int32_t,
                      float*, int64_t, void*, int64_t, float*, float*, void*, size_t, void*);
int vdk_reduce_rows_f32(const float*, int64_t, int32_t, int64_t, float*, float, void*);
int vdk_colsum_bf16_workspace_bytes(int32_t, int32_t, size_t*);
int vdk_colsum_bf16(const void*, int64_t, int32_t, int32_t, float*, void*, size_t, void*);
int vdk_patchify_bf16(const float*, int32_t, int32_t, int32_t, int32_t, int32_t, void*, int32_t, void*);
int vdk_cast_f32_bf16(const float*, void*, int64_t, void*);
int vdk_transpose_cast_f32_bf16(const float*, int64_t, int32_t, int32_t, void*, int64_t, int32_t, void*);
int vdk_transpose_bf16(const void*, int64_t, int32_t, int32_t, void*, int64_t, int32_t, int32_t, float*, void*);
int vdk_dwconv7_fwd(const float*, const float*, const float*, const float*, float*, void*, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_dwconv7_wgrad_workspace_bytes(int32_t, int32_t, int32_t, int32_t, size_t*);
int vdk_dwconv7_wgrad(const float*, const float*, float*, float*, int32_t, int32_t, int32_t, int32_t, void*, size_t, void*);
int vdk_dwconv7_weight_prep(const float*, float*, int32_t, void*);
int vdk_space_to_depth2_bf16(const void*, void*, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_conv2x2_weight_prep(const float*, void*, void*, int32_t, int32_t, void*);
int vdk_conv2x2_wgrad_unpermute(const float*, float*, int32_t, int32_t, void*);
int vdk_layerscale_weight_prep(const float*, const float*, const float*, void*, void*, float*, int32_t, int32_t, void*);
int vdk_layerscale_grad(const float*, const float*, const float*, const float*, const float*, float*, float*, float*, int32_t, int32_t, void*);
int vdk_gemm_f32_nt(const VdkGemmF32Desc*, void*);
int vdk_gemm_a_colsum_rows(int32_t, int32_t, int32_t);
int vdk_gemm_c_colsum_rows(int32_t, int32_t, int32_t);
int vdk_patchify_f32(const float*, int32_t, int32_t, int32_t, int32_t, int32_t, float*, void*);
int vdk_space_to_depth2_f32(const float*, float*, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_gelu_f32(const float*, float*, int64_t, void*);
int vdk_dgelu_f32(float*, const float*, int64_t, void*);
int vdk_rowscale_f32(const float*, const float*, float*, int64_t, int64_t, void*);
int vdk_colsum_f32_workspace_bytes(int64_t, int32_t, size_t*);
int vdk_colsum_f32(const float*, int64_t, int64_t, int32_t, float*, void*, size_t, void*);
int vdk_depth_to_space2_f32(const float*, float*, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_avgpool_rows_f32_fwd(const float*, float*, int32_t, int32_t, int32_t, void*);
int vdk_avgpool_rows_f32_bwd(const float*, float*, void*, int32_t, int32_t, int32_t, void*);
int vdk_scale_dev_f32(float*, int64_t, const float*, int32_t, void*);
}

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
static inline int64_t up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

namespace {

// operand format of the running call (VdkConvNextConfig.operand), set at every entry point as in vit_engine.hip / swin_engine.hip; DT16 = the dtype code of the 16-bit tensors
thread_local int t_opf = VDK_OPF_BF16;
#define DT16 (t_opf ? VDK_F16 : VDK_BF16)
// What a block keeps in `u` for the backward (as csrc/vit_engine.hip): the pre-activation in the operand format, or fp16 GELU'(pre-activation) from the fc1 epilogue, so that
// the dfc2 epilogue is one multiplication.  At ConvNeXt's K = 128 .. 1024 the GELU GEMMs are epilogue-bound and the trade pays: cfg3 109.55 -> 108.53 ms in two interleaved
// rounds on one box.  DEFAULT with fp16 operands, where it costs no accuracy (u is rounded to fp16 either way; the parity at C = 10^6 is unchanged: worst gradient 4.09e-3
// against 4.07e-3); bf16 operands keep rounds 1-4's arithmetic.  VDK_CN_GELU_SAVED_GRAD=0 / 1 forces it off / on for both formats.
bool cn_saved_grad() {
  static const int v = [] { const char* e = getenv("VDK_CN_GELU_SAVED_GRAD"); return e ? atoi(e) : -1; }();
  return v < 0 ? t_opf != 0 : v != 0;
}
#define CN_ACT_FC1 (cn_saved_grad() ? VDK_ACT_GELU_SAVE_GRAD : VDK_ACT_GELU)
#define CN_ACT_DFC2 (cn_saved_grad() ? VDK_ACT_MUL_AUX : VDK_ACT_DGELU)

struct CnDims {
  int opf;                // VDK_OPF_BF16 | VDK_OPF_F16
  int B, img, Cin, Kst;   // Kst = Cin * 16: K of the stem GEMM
  int depth[4], C[4], H[4], R[4];
  int nblk;
  float eps;
  int ncls, Cp, Bp;        // classifier mode (ncls > 0): timm's head = global average pool -> head.norm -> head.fc; Cp = classes padded to 8, Bp = batch padded to 64
};
int cn_dims(const VdkConvNextConfig* c, CnDims* d) {
  if (!c) return vdk_fail(VDK_EINVAL, "convnext: null config");
  if (c->batch <= 0 || c->img_size <= 0 || (c->img_size % 32) || c->in_chans <= 0) return vdk_fail(VDK_EINVAL, "convnext: bad config (img_size % 32 == 0)");
  d->B = c->batch; d->img = c->img_size; d->Cin = c->in_chans; d->Kst = c->in_chans * 16; d->eps = c->ln_eps; d->nblk = 0;
  if (c->num_classes < 0) return vdk_fail(VDK_EINVAL, "convnext: num_classes < 0");
  if (c->operand != VDK_BF16 && c->operand != VDK_F16) return vdk_fail(VDK_EINVAL, "convnext: operand must be VDK_BF16 or VDK_F16");
  d->opf = c->operand == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16;
  d->ncls = c->num_classes; d->Cp = (int)up(c->num_classes, 8); d->Bp = (int)up(c->batch, 64);
  for (int i = 0; i < 4; ++i) {
    if (c->depths[i] <= 0 || c->dims[i] <= 0 || (c->dims[i] & 7)) return vdk_fail(VDK_EINVAL, "convnext: bad config (dims % 8 == 0, depths > 0)");
    d->depth[i] = c->depths[i]; d->C[i] = c->dims[i]; d->H[i] = c->img_size >> (2 + i);
    const int64_t r = (int64_t)c->batch * d->H[i] * d->H[i];
    if (r > 0x7fffffffLL) return vdk_fail(VDK_EINVAL, "convnext: batch x resolution too large");
    d->R[i] = (int)r;
    d->nblk += c->depths[i];
  }
  return VDK_OK;
}

struct PEntry { char name[64]; int64_t off, numel; int64_t shape[4]; int ndim; };
struct BlkP { int64_t gamma, dw_w, dw_b, nw, nb, fc1_w, fc1_b, fc2_w, fc2_b; };
struct StageP { int64_t ds_nw, ds_nb, ds_w, ds_b; std::vector<BlkP> blk; };
struct PLayout {
  int64_t stem_w, stem_b, stem_nw, stem_nb, head_nw, head_nb, fc_w, fc_b, total;
  StageP st[4];
  std::vector<PEntry> entries;
};
int64_t p_take(int64_t& cur, int64_t n) { int64_t o = cur; cur = up(cur + n, 64); return o; }
void add_entry(PLayout* p, const char* name, int64_t off, int ndim, int64_t s0, int64_t s1 = 1, int64_t s2 = 1, int64_t s3 = 1) {
  PEntry e; memset(&e, 0, sizeof(e));
  snprintf(e.name, sizeof(e.name), "%s", name);
  e.off = off; e.ndim = ndim; e.shape[0] = s0; e.shape[1] = s1; e.shape[2] = s2; e.shape[3] = s3; e.numel = s0 * s1 * s2 * s3;
  p->entries.push_back(e);
}
// timm state_dict order and names
void cn_layout(const CnDims& d, PLayout* p) {
  int64_t cur = 0;
  char nm[64];
  p->entries.clear();
  p->stem_w = p_take(cur, (int64_t)d.C[0] * d.Kst); add_entry(p, "stem.0.weight", p->stem_w, 4, d.C[0], d.Cin, 4, 4);
  p->stem_b = p_take(cur, d.C[0]); add_entry(p, "stem.0.bias", p->stem_b, 1, d.C[0]);
  p->stem_nw = p_take(cur, d.C[0]); add_entry(p, "stem.1.weight", p->stem_nw, 1, d.C[0]);
  p->stem_nb = p_take(cur, d.C[0]); add_entry(p, "stem.1.bias", p->stem_nb, 1, d.C[0]);
  for (int i = 0; i < 4; ++i) {
    StageP& s = p->st[i];
    const int C = d.C[i], M = 4 * C;
    s.ds_nw = s.ds_nb = s.ds_w = s.ds_b = -1;
    if (i > 0) {
      const int Ci = d.C[i - 1];
      s.ds_nw = p_take(cur, Ci); snprintf(nm, 64, "stages.%d.downsample.0.weight", i); add_entry(p, nm, s.ds_nw, 1, Ci);
      s.ds_nb = p_take(cur, Ci); snprintf(nm, 64, "stages.%d.downsample.0.bias", i); add_entry(p, nm, s.ds_nb, 1, Ci);
      s.ds_w = p_take(cur, (int64_t)C * Ci * 4); snprintf(nm, 64, "stages.%d.downsample.1.weight", i); add_entry(p, nm, s.ds_w, 4, C, Ci, 2, 2);
      s.ds_b = p_take(cur, C); snprintf(nm, 64, "stages.%d.downsample.1.bias", i); add_entry(p, nm, s.ds_b, 1, C);
    }
    s.blk.resize(d.depth[i]);
    for (int j = 0; j < d.depth[i]; ++j) {
      BlkP& b = s.blk[j];
      b.gamma = p_take(cur, C); snprintf(nm, 64, "stages.%d.blocks.%d.gamma", i, j); add_entry(p, nm, b.gamma, 1, C);
      b.dw_w = p_take(cur, (int64_t)C * 49); snprintf(nm, 64, "stages.%d.blocks.%d.conv_dw.weight", i, j); add_entry(p, nm, b.dw_w, 4, C, 1, 7, 7);
      b.dw_b = p_take(cur, C); snprintf(nm, 64, "stages.%d.blocks.%d.conv_dw.bias", i, j); add_entry(p, nm, b.dw_b, 1, C);
      b.nw = p_take(cur, C); snprintf(nm, 64, "stages.%d.blocks.%d.norm.weight", i, j); add_entry(p, nm, b.nw, 1, C);
      b.nb = p_take(cur, C); snprintf(nm, 64, "stages.%d.blocks.%d.norm.bias", i, j); add_entry(p, nm, b.nb, 1, C);
      b.fc1_w = p_take(cur, (int64_t)M * C); snprintf(nm, 64, "stages.%d.blocks.%d.mlp.fc1.weight", i, j); add_entry(p, nm, b.fc1_w, 2, M, C);
      b.fc1_b = p_take(cur, M); snprintf(nm, 64, "stages.%d.blocks.%d.mlp.fc1.bias", i, j); add_entry(p, nm, b.fc1_b, 1, M);
      b.fc2_w = p_take(cur, (int64_t)C * M); snprintf(nm, 64, "stages.%d.blocks.%d.mlp.fc2.weight", i, j); add_entry(p, nm, b.fc2_w, 2, C, M);
      b.fc2_b = p_take(cur, C); snprintf(nm, 64, "stages.%d.blocks.%d.mlp.fc2.bias", i, j); add_entry(p, nm, b.fc2_b, 1, C);
    }
  }
  p->head_nw = p_take(cur, d.C[3]); add_entry(p, "head.norm.weight", p->head_nw, 1, d.C[3]);
  p->head_nb = p_take(cur, d.C[3]); add_entry(p, "head.norm.bias", p->head_nb, 1, d.C[3]);
  p->fc_w = p->fc_b = 0;
  if (d.ncls > 0) {          // rows >= ncls of the stored [Cp, C3] weight are padding (zero gradient, untouched by state_dict I/O)
    p->fc_w = p_take(cur, (int64_t)d.Cp * d.C[3]); add_entry(p, "head.fc.weight", p->fc_w, 2, d.ncls, d.C[3]);
    p->fc_b = p_take(cur, d.Cp); add_entry(p, "head.fc.bias", p->fc_b, 1, d.ncls);
  }
  p->total = cur;
}

// derived operand copies (`wx`, byte offsets): per block the tap-major depthwise weight, fc1^T, the layer-scale-folded fc2 and its
// transpose + bias; per downsample the (ky,kx,cin)-ordered weight and its transpose
struct BlkX { size_t dwt, fc1t, fc2p, fc2pt, b2p, rs; };   // rs: f32 {r, 1 / r}, the block's branch scale under fp16 operands
struct XLayout { size_t total; size_t dsw[4], dswt[4]; std::vector<BlkX> blk[4]; size_t fct; };   // fct: head.fc^T [C3, Cp] bf16 (classifier mode)
size_t w_take(size_t& cur, size_t n) { size_t o = cur; cur = (cur + n + 255) & ~(size_t)255; return o; }
void cn_xlayout(const CnDims& d, XLayout* x) {
  size_t cur = 0;
  for (int i = 0; i < 4; ++i) {
    const size_t C = d.C[i], M = 4 * C;
    x->dsw[i] = x->dswt[i] = 0;
    if (i > 0) { x->dsw[i] = w_take(cur, C * 4 * d.C[i - 1] * 2); x->dswt[i] = w_take(cur, C * 4 * d.C[i - 1] * 2); }
    x->blk[i].resize(d.depth[i]);
    for (int j = 0; j < d.depth[i]; ++j) {
      BlkX& b = x->blk[i][j];
      b.dwt = w_take(cur, 49 * C * 4); b.fc1t = w_take(cur, C * M * 2); b.fc2p = w_take(cur, C * M * 2); b.fc2pt = w_take(cur, C * M * 2); b.b2p = w_take(cur, C * 4); b.rs = w_take(cur, 8);
    }
  }
  x->fct = d.ncls > 0 ? w_take(cur, (size_t)d.C[3] * d.Cp * 2) : 0;
  x->total = cur;
}

int wgrad_splitk(int M, int N, int K) {
  if (K < 4096) return 1;
  int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int s = (1024 + tiles - 1) / tiles;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}
int wgrad_splitk_tn(int M, int N, int K) {   // see csrc/vit_engine.hip: tiles * splits just under a whole number of 256-CU rounds
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  int s = 256 / tiles;
  if (s < 1) s = 1;
  const int kt = K / 64;
  if (s > kt / 4) s = kt / 4;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

struct BlkW { size_t t, stats, h, u, g; };
struct WsPlan {
  size_t total;
  size_t patches, y0, stats0;
  size_t X[4];                       // f32 (depth + 1) x [R, C]
  size_t ds_stats[4], ds_h[4], ds_A[4];
  std::vector<BlkW> blk[4];
  size_t head_stats;
  size_t pooled, pstats, feat, dfeat, dpool;   // classifier mode: pooled f32 [B, C3], its LayerNorm statistics, normed bf16 [Bp, C3], and their gradients
  // backward scratch (sized by the largest stage)
  size_t dxa, dxb, dt, du, dh, dA, dhds, dw2p, db2p, dwdsp;
  size_t slabs, slabs_bytes, lnws, lnws_bytes, csws, csws_bytes, csws2, dwws, dwws_bytes, tA, tB;
};
void cn_plan(const CnDims& d, WsPlan* w) {
  size_t cur = 0;
  w->patches = w_take(cur, (size_t)d.R[0] * d.Kst * 2);
  w->y0 = w_take(cur, (size_t)d.R[0] * d.C[0] * 4);
  w->stats0 = w_take(cur, (size_t)d.R[0] * 2 * 4);
  size_t rc = 0, rm = 0, ra = 0, rh = 0, cm = 0, wds = 0, sl = 0, ln = 0, cs = 0, dww = 0, tr = 0;
  auto wg = [&](int out, int in, int rows) {
    const int k1 = wgrad_splitk(out, in, (int)up(rows, 64)), k2 = wgrad_splitk_tn(out, in, rows);
    const size_t b = (size_t)(k1 > k2 ? k1 : k2) * out * in * 4; if (b > sl) sl = b;
    size_t c2 = 0; vdk_colsum_bf16_workspace_bytes(rows, out, &c2);
    const size_t c1 = (size_t)((up(rows, 64) + 63) / 64) * out * 4;
    if (c2 > cs) cs = c2;
    if (c1 > cs) cs = c1;
    { const size_t c3 = (size_t)2 * ((rows + 255) / 256) * out * 4; if (c3 > cs) cs = c3; }   // a_colsum / c_colsum by-products of the dgrad GEMMs
    { const size_t c4 = (size_t)1024 * out * 4; if (c4 > cs) cs = c4; }                         // vdk_colsum_f32_deferred: at most 1024 row splits (fp16 operands: bias gradients from the fp32 stream)
    if (rows % 64) { const size_t t = (size_t)(out > in ? out : in) * up(rows, 64) * 2; if (t > tr) tr = t; }
  };
  wg(d.C[0], d.Kst, d.R[0]);
  for (int i = 0; i < 4; ++i) {
    const size_t R = d.R[i], C = d.C[i], M = 4 * C;
    w->ds_stats[i] = w->ds_h[i] = w->ds_A[i] = 0;
    if (i > 0) {
      const size_t Rp = d.R[i - 1], Ci = d.C[i - 1];
      w->ds_stats[i] = w_take(cur, Rp * 2 * 4); w->ds_h[i] = w_take(cur, Rp * Ci * 2); w->ds_A[i] = w_take(cur, R * 4 * Ci * 2);
      if (R * 4 * Ci > ra) ra = R * 4 * Ci;
      if (Rp * Ci > rh) rh = Rp * Ci;
      if (C * 4 * Ci > wds) wds = C * 4 * Ci;
      wg((int)C, (int)(4 * Ci), (int)R);
      size_t l = 0; vdk_layernorm_bwd_workspace_bytes((int)Rp, (int)Ci, &l); if (l > ln) ln = l;
    }
    w->X[i] = w_take(cur, (size_t)(d.depth[i] + 1) * R * C * 4);
    w->blk[i].resize(d.depth[i]);
    for (int j = 0; j < d.depth[i]; ++j) {
      BlkW& b = w->blk[i][j];
      b.t = w_take(cur, R * C * 4); b.stats = w_take(cur, R * 2 * 4); b.h = w_take(cur, R * C * 2); b.u = w_take(cur, R * M * 2); b.g = w_take(cur, R * M * 2);
    }
    if (R * C > rc) rc = R * C;
    if (R * M > rm) rm = R * M;
    if (C * M > cm) cm = C * M;
    wg((int)C, (int)M, (int)R); wg((int)M, (int)C, (int)R);
    size_t l = 0; vdk_layernorm_bwd_workspace_bytes((int)R, (int)C, &l); if (l > ln) ln = l;
    size_t q = 0; vdk_dwconv7_wgrad_workspace_bytes(d.B, d.H[i], d.H[i], (int)C, &q); if (q > dww) dww = q;
  }
  w->head_stats = w_take(cur, (size_t)d.R[3] * 2 * 4);
  w->pooled = w->pstats = w->feat = w->dfeat = w->dpool = 0;
  if (d.ncls > 0) {
    w->pooled = w_take(cur, (size_t)d.B * d.C[3] * 4); w->pstats = w_take(cur, (size_t)d.B * 2 * 4); w->feat = w_take(cur, (size_t)d.Bp * d.C[3] * 2);
    w->dfeat = w_take(cur, (size_t)d.Bp * d.C[3] * 4); w->dpool = w_take(cur, (size_t)d.B * d.C[3] * 4);
    wg(d.Cp, d.C[3], d.Bp);
    size_t l = 0; vdk_layernorm_bwd_workspace_bytes(d.B, d.C[3], &l); if (l > ln) ln = l;
  }
  w->dxa = w_take(cur, rc * 4); w->dxb = w_take(cur, rc * 2); w->dt = w_take(cur, rc * 4); w->du = w_take(cur, rm * 2); w->dh = w_take(cur, rc * 2);
  w->dA = w_take(cur, ra * 2 + 256); w->dhds = w_take(cur, rh * 2 + 256);
  w->dw2p = w_take(cur, cm * 4); w->db2p = w_take(cur, 4096 * 4); w->dwdsp = w_take(cur, wds * 4 + 256);
  w->slabs_bytes = sl; w->slabs = w_take(cur, sl + 256);
  w->lnws_bytes = ln; w->lnws = w_take(cur, ln + 256);
  w->csws_bytes = cs; w->csws = w_take(cur, cs + 256); w->csws2 = w_take(cur, cs + 256);   // two: a block's deferred reductions keep both alive until its batch launch
  w->dwws_bytes = dww; w->dwws = w_take(cur, dww + 256);
  w->tA = w_take(cur, tr + 256); w->tB = w_take(cur, tr + 256);
  w->total = cur;
}

// cdt: VDK_F32, or VDK_BF16 = "the 16-bit operand format" (translated to the running call's DT16)
int gemm(hipStream_t s, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, int cdt, const float* bias,
         const float* res, int64_t ldr, int act, void* aux, int64_t ldaux, const float* col_scale = nullptr) {
  VdkGemmDesc g = {};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.c_dtype = cdt == VDK_F32 ? VDK_F32 : DT16; g.bias = bias;
  g.residual = res; g.ldr = ldr; g.act = act; g.aux = aux; g.ldaux = ldaux; g.alpha = 1.0f; g.splitk = 1; g.ab_dtype = DT16; g.col_scale = col_scale;
  return vdk_gemm_bf16_nt(&g, nullptr, 0, s);
}
// dW[out,in] = dY^T X, db = colsum(dY)   (dY bf16 [rows,out], X bf16 [rows,in]); TN LDS-DMA kernel when rows % 64 == 0, else explicit transposes
int linear_wgrad(hipStream_t s, const WsPlan& w, char* base, const bf16_t* dY, const bf16_t* X, int rows, int out, int in, float* dW, float* db) {
  if ((rows % 64) == 0) {
    VdkGemmDesc g = {};
    g.A = dY; g.lda = out; g.B = X; g.ldb = in; g.C = dW; g.ldc = in; g.M = out; g.N = in; g.K = rows; g.c_dtype = VDK_F32;
    g.alpha = 1.0f; g.splitk = wgrad_splitk_tn(out, in, rows); g.trans = 1; g.ab_dtype = DT16;
    RC(vdk_gemm_bf16_nt(&g, base + w.slabs, w.slabs_bytes, s));
    return db ? vdk_colsum_16(dY, out, rows, out, db, base + w.csws, w.csws_bytes, t_opf, s) : VDK_OK;
  }
  const int rp = (int)up(rows, 64);
  bf16_t* tA = (bf16_t*)(base + w.tA); bf16_t* tB = (bf16_t*)(base + w.tB);
  float* csp = db ? (float*)(base + w.csws) : nullptr;
  RC(vdk_transpose_16(dY, out, rows, out, tA, rp, rp, 0, csp, t_opf, s));
  RC(vdk_transpose_16(X, in, rows, in, tB, rp, rp, 0, nullptr, t_opf, s));
  VdkGemmDesc g = {};
  g.A = tA; g.lda = rp; g.B = tB; g.ldb = rp; g.C = dW; g.ldc = in; g.M = out; g.N = in; g.K = rp; g.c_dtype = VDK_F32; g.alpha = 1.0f; g.ab_dtype = DT16;
  g.splitk = wgrad_splitk(out, in, rp);
  RC(vdk_gemm_bf16_nt(&g, base + w.slabs, w.slabs_bytes, s));
  return db ? vdk_reduce_rows_f32(csp, out, (rp + 63) / 64, out, db, 1.0f, s) : VDK_OK;
}


// dgrad GEMM dX[rows, in] = act'(dY[rows, out] . Wt[in, out]^T) that also delivers db' = colsum(dY) as a by-product of its A tiles when the 256x256 kernel serves
// the problem (see csrc/vit_engine.hip); *fused tells the caller whether db was produced
int dgrad_with_bias(hipStream_t s, const WsPlan& w, char* base, const void* dY, const void* Wt, void* dX, int rows, int in, int out, int act, void* aux, float* db,
                    int* fused) {
  const int prow = vdk_gemm_a_colsum_rows(rows, in, out);
  *fused = (db && !t_opf && prow > 0 && (size_t)prow * out * 4 <= w.csws_bytes) ? 1 : 0;      // (the A-tile column sums are a by-product of the eight-wave bf16 kernel only)
  VdkGemmDesc g = {};
  g.A = dY; g.lda = out; g.B = Wt; g.ldb = out; g.C = dX; g.ldc = in; g.M = rows; g.N = in; g.K = out; g.c_dtype = DT16; g.act = act; g.aux = aux; g.ldaux = in;
  g.alpha = 1.0f; g.splitk = 1; g.ab_dtype = DT16;
  if (*fused) g.a_colsum = (float*)(base + w.csws);
  RC(vdk_gemm_bf16_nt(&g, nullptr, 0, s));
  if (*fused) RC(vdk_reduce_rows_f32((const float*)(base + w.csws), out, prow, out, db, 1.0f, s));
  return VDK_OK;
}

}  // namespace

extern "C" {

int vdk_convnext_param_count(const VdkConvNextConfig* cfg, int64_t* n_floats, int32_t* n_tensors, size_t* wx_bytes) {
  CnDims d; RC(cn_dims(cfg, &d));
  PLayout p; cn_layout(d, &p);
  XLayout x; cn_xlayout(d, &x);
  if (n_floats) *n_floats = p.total;
  if (n_tensors) *n_tensors = (int32_t)p.entries.size();
  if (wx_bytes) *wx_bytes = x.total;
  return VDK_OK;
}

int vdk_convnext_param_info(const VdkConvNextConfig* cfg, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int64_t* shape4,
                            int32_t* ndim) {
  CnDims d; RC(cn_dims(cfg, &d));
  PLayout p; cn_layout(d, &p);
  if (index < 0 || index >= (int32_t)p.entries.size()) return vdk_fail(VDK_EINVAL, "vdk_convnext_param_info: index out of range");
  const PEntry& e = p.entries[index];
  if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", e.name);
  if (offset) *offset = e.off;
  if (numel) *numel = e.numel;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = e.shape[i];
  if (ndim) *ndim = e.ndim;
  return VDK_OK;
}

int vdk_convnext_workspace_bytes(const VdkConvNextConfig* cfg, size_t* bytes) {
  CnDims d; RC(cn_dims(cfg, &d));
  WsPlan w; cn_plan(d, &w);
  if (!bytes) return vdk_fail(VDK_EINVAL, "null");
  *bytes = w.total;
  return VDK_OK;
}

// wb16 = bf16 copy of the flat parameters (stem and fc1 operands are read from it as they lie) unless skip_wb16; wx = derived copies
int vdk_convnext_refresh_weights(const VdkConvNextConfig* cfg, const float* params, void* wb16, void* wx, int32_t skip_wb16, void* stream) {
  CnDims d; RC(cn_dims(cfg, &d));
  PLayout p; cn_layout(d, &p);
  XLayout x; cn_xlayout(d, &x);
  if (!params || !wb16 || !wx) return vdk_fail(VDK_EINVAL, "vdk_convnext_refresh_weights: null pointer");
  t_opf = d.opf;
  if (!skip_wb16) RC(vdk_cast_f32_16(params, wb16, p.total, t_opf, stream));
  char* xb = (char*)wx;
  std::vector<VdkTcItem> jobs;      // the fc1^T (and head.fc^T) copies of all blocks in one launch
  std::vector<CnPrepJob> prep;      // ... and every block's depthwise tap-major copy + layer-scale fold in another (conv.hip: cn_prep_batch_kernel)
  for (int i = 0; i < 4; ++i) {
    const int C = d.C[i], M = 4 * C;
    if (i > 0) RC(vdk_conv2x2_weight_prep_16(params + p.st[i].ds_w, xb + x.dsw[i], xb + x.dswt[i], C, d.C[i - 1], t_opf, stream));
    for (int j = 0; j < d.depth[i]; ++j) {
      const BlkP& b = p.st[i].blk[j]; const BlkX& bx = x.blk[i][j];
      jobs.push_back(VdkTcItem{params + b.fc1_w, xb + bx.fc1t, C, M, C, M, M});
      prep.push_back(CnPrepJob{params + b.dw_w, (float*)(xb + bx.dwt), params + b.fc2_w, params + b.fc2_b, params + b.gamma, (bf16_t*)(xb + bx.fc2p), (bf16_t*)(xb + bx.fc2pt),
                               (float*)(xb + bx.b2p), C, M, t_opf ? (float*)(xb + bx.rs) : nullptr});
    }
  }
  RC(vdk_convnext_prep_blocks(prep.data(), (int)prep.size(), stream));
  if (d.ncls > 0) jobs.push_back(VdkTcItem{params + p.fc_w, xb + x.fct, d.C[3], d.Cp, d.C[3], d.Cp, d.Cp});
  return vdk_transpose_cast_batch(jobs.data(), (int)jobs.size(), stream, t_opf);
}

// x f32 [B, Cin, img, img] (NCHW, as the reference's dataloader hands it) -> out f32 [B * (img/32)^2, dims[3]]: the head-normed map in NHWC rows
int vdk_convnext_forward(const VdkConvNextConfig* cfg, const float* x, const float* params, const void* wb16, const void* wx, void* ws, size_t ws_bytes,
                         float* out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  CnDims d; RC(cn_dims(cfg, &d));
  PLayout p; cn_layout(d, &p);
  XLayout xl; cn_xlayout(d, &xl);
  WsPlan w; cn_plan(d, &w);
  if (!x || !params || !wb16 || !wx || !ws || !out) return vdk_fail(VDK_EINVAL, "vdk_convnext_forward: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_convnext_forward: workspace too small");
  char* base = (char*)ws;
  const bf16_t* wb = (const bf16_t*)wb16;
  const char* xb = (const char*)wx;
  t_opf = d.opf;
  // stem: Conv2d(Cin, C0, 4, stride 4) as patchify + GEMM, then LayerNorm2d
  bf16_t* patches = (bf16_t*)(base + w.patches);
  float* y0 = (float*)(base + w.y0);
  float* st0 = (float*)(base + w.stats0);
  RC(vdk_patchify_16(x, d.B, d.Cin, d.img, d.img, 4, patches, d.Kst, t_opf, s));
  RC(gemm(s, patches, d.Kst, wb + p.stem_w, d.Kst, y0, d.C[0], d.R[0], d.C[0], d.Kst, VDK_F32, params + p.stem_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0));
  RC(vdk_layernorm_fwd(y0, d.C[0], d.R[0], d.C[0], params + p.stem_nw, params + p.stem_nb, d.eps, base + w.X[0], d.C[0], VDK_F32, st0, st0 + d.R[0], s));
  for (int i = 0; i < 4; ++i) {
    const int R = d.R[i], C = d.C[i], M = 4 * C, H = d.H[i];
    float* X = (float*)(base + w.X[i]);
    const size_t XS = (size_t)R * C;
    if (i > 0) {
      // downsample: LayerNorm2d -> Conv2d(Ci, C, 2, stride 2) as space-to-depth + GEMM
      const int Rp = d.R[i - 1], Ci = d.C[i - 1];
      const float* xprev = (const float*)(base + w.X[i - 1]) + (size_t)d.depth[i - 1] * Rp * Ci;
      float* dst = (float*)(base + w.ds_stats[i]);
      RC(vdk_layernorm_fwd(xprev, Ci, Rp, Ci, params + p.st[i].ds_nw, params + p.st[i].ds_nb, d.eps, base + w.ds_h[i], Ci, DT16, dst, dst + Rp, s));
      RC(vdk_space_to_depth2_bf16(base + w.ds_h[i], base + w.ds_A[i], d.B, d.H[i - 1], d.H[i - 1], Ci, 0, s));
      RC(gemm(s, base + w.ds_A[i], 4 * Ci, xb + xl.dsw[i], 4 * Ci, X, C, R, C, 4 * Ci, VDK_F32, params + p.st[i].ds_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0));
    }
    for (int j = 0; j < d.depth[i]; ++j) {
      const BlkP& b = p.st[i].blk[j]; const BlkX& bx = xl.blk[i][j]; const BlkW& bw = w.blk[i][j];
      float* xin = X + (size_t)j * XS; float* xout = xin + XS;
      float* t = (float*)(base + bw.t); float* st = (float*)(base + bw.stats);
      RC(vdk_dwconv7_fwd(xin, (const float*)(xb + bx.dwt), params + b.dw_b, nullptr, t, nullptr, d.B, H, H, C, 0, s));
      RC(vdk_layernorm_fwd(t, C, R, C, params + b.nw, params + b.nb, d.eps, base + bw.h, C, DT16, st, st + R, s));
      RC(gemm(s, base + bw.h, C, wb + b.fc1_w, C, base + bw.g, M, R, M, C, VDK_BF16, params + b.fc1_b, nullptr, 0, CN_ACT_FC1, base + bw.u, M));
      // fp16 operands: plain W2, gamma applied to the accumulators (col_scale), b2p = gamma (.) b2; bf16: gamma folded into fc2p
      RC(gemm(s, base + bw.g, M, xb + bx.fc2p, M, xout, C, R, C, M, VDK_F32, (const float*)(xb + bx.b2p), xin, C, VDK_ACT_NONE, nullptr, 0, t_opf ? params + b.gamma : nullptr));
    }
  }
  const float* xlast = (const float*)(base + w.X[3]) + (size_t)d.depth[3] * d.R[3] * d.C[3];
  if (d.ncls > 0) {
    // timm ConvNeXt head with num_classes > 0 (NormMlpClassifierHead): global average pool -> LayerNorm (head.norm) -> Linear (head.fc); out = logits f32 [B, Cp]
    float* pooled = (float*)(base + w.pooled); float* ps = (float*)(base + w.pstats);
    RC(vdk_avgpool_rows_f32_fwd(xlast, pooled, d.B, d.H[3] * d.H[3], d.C[3], s));
    if (d.Bp > d.B && hipMemsetAsync(base + w.feat + (size_t)d.B * d.C[3] * 2, 0, (size_t)(d.Bp - d.B) * d.C[3] * 2, s) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_convnext_forward: memset");
    RC(vdk_layernorm_fwd(pooled, d.C[3], d.B, d.C[3], params + p.head_nw, params + p.head_nb, d.eps, base + w.feat, d.C[3], DT16, ps, ps + d.B, s));
    RC(gemm(s, base + w.feat, d.C[3], wb + p.fc_w, d.C[3], out, d.Cp, d.B, d.Cp, d.C[3], VDK_F32, params + p.fc_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0));
    return vdk_check_launch("vdk_convnext_forward");
  }
  float* hs = (float*)(base + w.head_stats);
  RC(vdk_layernorm_fwd(xlast, d.C[3], d.R[3], d.C[3], params + p.head_nw, params + p.head_nb, d.eps, out, d.C[3], VDK_F32, hs, hs + d.R[3], s));
  return vdk_check_launch("vdk_convnext_forward");
}

// dout f32 [B * (img/32)^2, dims[3]] (feature mode) or dlogits bf16 [up(B, 64), up(num_classes, 8)] (classifier mode) -> grads (flat fp32, param layout, fully overwritten).  on_ready(user, offset, numel): see vdk_vit_backward.
int vdk_convnext_backward(const VdkConvNextConfig* cfg, const void* dout_, const float* params, const void* wb16, const void* wx, void* ws, size_t ws_bytes,
                          float* grads, vdk_grad_ready_fn on_ready, void* user, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  const float* dout = (const float*)dout_;
  CnDims d; RC(cn_dims(cfg, &d));
  PLayout p; cn_layout(d, &p);
  XLayout xl; cn_xlayout(d, &xl);
  WsPlan w; cn_plan(d, &w);
  if (!dout || !params || !wb16 || !wx || !ws || !grads) return vdk_fail(VDK_EINVAL, "vdk_convnext_backward: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_convnext_backward: workspace too small");
  char* base = (char*)ws;
  const char* xb = (const char*)wx;
  t_opf = d.opf;
  float* dxa = (float*)(base + w.dxa); bf16_t* dxb = (bf16_t*)(base + w.dxb);
  float* dt = (float*)(base + w.dt); bf16_t* du = (bf16_t*)(base + w.du); bf16_t* dh = (bf16_t*)(base + w.dh);
  float* dw2p = (float*)(base + w.dw2p); float* db2p = (float*)(base + w.db2p);
  void* lnws = base + w.lnws;
  if (d.ncls > 0) {
    // dout = dlogits bf16 [Bp, Cp] (rows >= B and columns >= num_classes zero): fc weight / bias gradient, feature gradient, head.norm backward on the
    // pooled rows, then the pooled gradient spread over the map (1 / HW each)
    const bf16_t* dl = (const bf16_t*)dout_;
    RC(linear_wgrad(s, w, base, dl, (const bf16_t*)(base + w.feat), d.Bp, d.Cp, d.C[3], grads + p.fc_w, grads + p.fc_b));
    RC(gemm(s, dl, d.Cp, xb + xl.fct, d.Cp, base + w.dfeat, d.C[3], d.B, d.C[3], d.Cp, VDK_F32, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0));
    const float* ps = (const float*)(base + w.pstats);
    RC(vdk_layernorm_bwd(base + w.dfeat, d.C[3], VDK_F32, (const float*)(base + w.pooled), d.C[3], ps, ps + d.B, params + p.head_nw, nullptr, 0, d.B, d.C[3],
                         (float*)(base + w.dpool), d.C[3], nullptr, 0, grads + p.head_nw, grads + p.head_nb, lnws, w.lnws_bytes, s));
    RC(vdk_avgpool_rows_f32_bwd_16((const float*)(base + w.dpool), dxa, dxb, d.B, d.H[3] * d.H[3], d.C[3], t_opf, s));
    if (on_ready) on_ready(user, p.head_nw, p.total - p.head_nw);
  } else {
    const float* xlast = (const float*)(base + w.X[3]) + (size_t)d.depth[3] * d.R[3] * d.C[3];
    const float* hs = (const float*)(base + w.head_stats);
    RC(vdk_layernorm_bwd_deferred(dout, d.C[3], VDK_F32, xlast, d.C[3], hs, hs + d.R[3], params + p.head_nw, nullptr, 0, d.R[3], d.C[3], dxa, d.C[3], dxb, d.C[3],
                                  grads + p.head_nw, grads + p.head_nb, lnws, w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf));
    if (on_ready) on_ready(user, p.head_nw, p.total - p.head_nw);
  }
  for (int i = 3; i >= 0; --i) {
    const int R = d.R[i], C = d.C[i], M = 4 * C, H = d.H[i];
    const float* X = (const float*)(base + w.X[i]);
    const size_t XS = (size_t)R * C;
    if (C > 4096) return vdk_fail(VDK_EUNSUPPORTED, "vdk_convnext_backward: dims <= 4096");
    for (int j = d.depth[i] - 1; j >= 0; --j) {
      const BlkP& b = p.st[i].blk[j]; const BlkX& bx = xl.blk[i][j]; const BlkW& bw = w.blk[i][j];
      const float* xin = X + (size_t)j * XS;
      const float* st = (const float*)(base + bw.stats);
      // dxa / dxb = dL/d(block output).  MLP branch with the layer scale folded into fc2; the bias gradients ride along with the dgrad GEMMs when possible
      int fz = 0;
      // fp16 operands: fc2pt holds ((gamma r) (.) W2)^T, so du, dh and the fc1 gradients come out r times their size (r = the block's power-of-two branch scale, rs[0]);
      // LayerNorm backward multiplies rs[1] = 1 / r into dh as it loads it, and the fc1 weight / bias gradients (adjacent in the flat buffer) are rescaled in place.
      const float* bsc = t_opf ? (const float*)(xb + bx.rs) + 1 : nullptr;
      auto fc1_unscale = [&]() -> int { return bsc ? vdk_scale_dev_f32(grads + b.fc1_w, b.fc2_w - b.fc1_w, bsc, 0, s) : VDK_OK; };
      // fc1.bias = column sums of du, accumulated by the dGELU epilogue that stores du (c_colsum: the PRODUCER sums what it writes); the fc2' bias is a column-sum pass over
      // the [R, C] gradient dxb (4x narrower than du).  Both input-gradient GEMMs then run the plain 225-register variants instead of the a_colsum ones (256 + spills).
      const int xrow = vdk_gemm_c_colsum_rows(R, M, C);
      if (xrow > 0 && (size_t)xrow * M * 4 <= w.csws_bytes) {
        VdkGemmDesc g = {};
        g.A = dxb; g.lda = C; g.B = xb + bx.fc2pt; g.ldb = C; g.C = du; g.ldc = M; g.M = R; g.N = M; g.K = C; g.c_dtype = DT16; g.act = CN_ACT_DFC2; g.aux = base + bw.u;
        g.ldaux = M; g.alpha = 1.0f; g.splitk = 1; g.c_colsum = (float*)(base + w.csws); g.ab_dtype = DT16;
        RC(vdk_gemm_bf16_nt(&g, nullptr, 0, s));
        // The block's four small reductions (fc1.bias from the dGELU epilogue's column sums, the fc2' bias column sums, LayerNorm dgamma | dbeta, depthwise dw | db) run as
        // ONE launch after the depthwise weight-gradient kernel; the layer-scale kernel that needs db2p follows it.
        VdkReduceJob jobs[4]; int nj = 0;
        jobs[nj++] = VdkReduceJob{(float*)(base + w.csws), (long)M, xrow, (long)M, grads + b.fc1_b, 1.0f};
        const bool tn = (R % 64) == 0 && b.dw_b == b.dw_w + (int64_t)C * 49;
        if (tn) {
          RC(linear_wgrad(s, w, base, dxb, (const bf16_t*)(base + bw.g), R, C, M, dw2p, nullptr));
          // fc2' bias gradient = column sums of the block-output gradient.  The neck's BatchNorm makes those sums cancel (sum over rows of a BatchNorm input gradient
          // is zero; LayerNorm2d only perturbs that): |sum| ~ 0.03 x the l2 of the column, so summing the 16-bit COPY costs 30 x its rounding (5.6e-3 in fp16, 4e-2 in
          // bf16 on stages.3's last block).  fp16 operands sum the fp32 stream dxa instead; bf16 keeps rounds 1-4's arithmetic.
          if (t_opf) RC(vdk_colsum_f32_deferred(dxa, C, R, C, db2p, base + w.csws2, w.csws_bytes, s, &jobs[nj]));
          else RC(vdk_colsum_bf16_deferred(dxb, C, R, C, db2p, base + w.csws2, w.csws_bytes, s, &jobs[nj], nullptr, t_opf));
          ++nj;
        } else {
          RC(linear_wgrad(s, w, base, dxb, (const bf16_t*)(base + bw.g), R, C, M, dw2p, t_opf ? nullptr : db2p));        // db2p by vdk_colsum_bf16 / the transposes' by-product inside
          if (t_opf) RC(vdk_colsum_f32(dxa, C, R, C, db2p, base + w.csws2, w.csws_bytes, s));
        }
        RC(gemm(s, du, M, xb + bx.fc1t, M, dh, C, R, C, M, VDK_BF16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0));
        RC(linear_wgrad(s, w, base, du, (const bf16_t*)(base + bw.h), R, M, C, grads + b.fc1_w, nullptr));
        if (tn) {
          RC(vdk_layernorm_bwd_deferred(dh, C, DT16, (const float*)(base + bw.t), C, st, st + R, params + b.nw, nullptr, 0, R, C, dt, C, nullptr, 0, grads + b.nw,
                                        grads + b.nb, lnws, w.lnws_bytes, s, &jobs[nj], nullptr, nullptr, nullptr, t_opf, bsc)); ++nj;
          RC(vdk_dwconv7_wgrad_deferred(xin, dt, grads + b.dw_w, grads + b.dw_b, d.B, H, H, C, base + w.dwws, w.dwws_bytes, s, &jobs[nj])); ++nj;
          RC(vdk_reduce_rows_batch(jobs, nj, s));
          RC(fc1_unscale());
          RC(vdk_layerscale_grad(dw2p, db2p, params + b.fc2_w, params + b.fc2_b, params + b.gamma, grads + b.fc2_w, grads + b.fc2_b, grads + b.gamma, C, M, s));
          RC(vdk_dwconv7_fwd_16(dt, (const float*)(xb + bx.dwt), nullptr, dxa, dxa, dxb, d.B, H, H, C, 1, t_opf, s));
          if (on_ready) {
            const int64_t end = (j + 1 < d.depth[i]) ? p.st[i].blk[j + 1].gamma : (i < 3 ? p.st[i + 1].ds_nw : p.head_nw);
            on_ready(user, b.gamma, end - b.gamma);
          }
          continue;
        }
        RC(vdk_reduce_rows_batch(jobs, nj, s));
        RC(fc1_unscale());
        RC(vdk_layerscale_grad(dw2p, db2p, params + b.fc2_w, params + b.fc2_b, params + b.gamma, grads + b.fc2_w, grads + b.fc2_b, grads + b.gamma, C, M, s));
      } else {
      RC(dgrad_with_bias(s, w, base, dxb, xb + bx.fc2pt, du, R, M, C, CN_ACT_DFC2, base + bw.u, db2p, &fz));
      RC(linear_wgrad(s, w, base, dxb, (const bf16_t*)(base + bw.g), R, C, M, dw2p, (fz || t_opf) ? nullptr : db2p));
      if (t_opf) RC(vdk_colsum_f32(dxa, C, R, C, db2p, base + w.csws2, w.csws_bytes, s));
      RC(vdk_layerscale_grad(dw2p, db2p, params + b.fc2_w, params + b.fc2_b, params + b.gamma, grads + b.fc2_w, grads + b.fc2_b, grads + b.gamma, C, M, s));
      RC(dgrad_with_bias(s, w, base, du, xb + bx.fc1t, dh, R, C, M, VDK_ACT_NONE, nullptr, grads + b.fc1_b, &fz));
      RC(linear_wgrad(s, w, base, du, (const bf16_t*)(base + bw.h), R, M, C, grads + b.fc1_w, fz ? nullptr : grads + b.fc1_b));
      RC(fc1_unscale());
      }
      RC(vdk_layernorm_bwd_deferred(dh, C, DT16, (const float*)(base + bw.t), C, st, st + R, params + b.nw, nullptr, 0, R, C, dt, C, nullptr, 0, grads + b.nw,
                                    grads + b.nb, lnws, w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf, bsc));
      // depthwise conv: weight/bias gradient, then input gradient + shortcut gradient (in place on dxa) and its bf16 copy
      RC(vdk_dwconv7_wgrad(xin, dt, grads + b.dw_w, grads + b.dw_b, d.B, H, H, C, base + w.dwws, w.dwws_bytes, s));
      RC(vdk_dwconv7_fwd_16(dt, (const float*)(xb + bx.dwt), nullptr, dxa, dxa, dxb, d.B, H, H, C, 1, t_opf, s));
      if (on_ready) {
        const int64_t end = (j + 1 < d.depth[i]) ? p.st[i].blk[j + 1].gamma : (i < 3 ? p.st[i + 1].ds_nw : p.head_nw);
        on_ready(user, b.gamma, end - b.gamma);
      }
    }
    if (i > 0) {
      // dxa / dxb = dL/d(downsample conv output) [R, C]
      const int Rp = d.R[i - 1], Ci = d.C[i - 1];
      float* dwdsp = (float*)(base + w.dwdsp);
      RC(linear_wgrad(s, w, base, dxb, (const bf16_t*)(base + w.ds_A[i]), R, C, 4 * Ci, dwdsp, t_opf ? nullptr : grads + p.st[i].ds_b));
      if (t_opf) RC(vdk_colsum_f32(dxa, C, R, C, grads + p.st[i].ds_b, base + w.csws2, w.csws_bytes, s));      // the downsample conv's bias gradient from the fp32 stream (see the blocks)
      RC(vdk_conv2x2_wgrad_unpermute(dwdsp, grads + p.st[i].ds_w, C, Ci, s));
      RC(gemm(s, dxb, C, xb + xl.dswt[i], C, base + w.dA, 4 * Ci, R, 4 * Ci, C, VDK_BF16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0));      // (VDK_BF16 = "the operand format")
      RC(vdk_space_to_depth2_bf16(base + w.dA, base + w.dhds, d.B, d.H[i - 1], d.H[i - 1], Ci, 1, s));
      const float* xprev = (const float*)(base + w.X[i - 1]) + (size_t)d.depth[i - 1] * Rp * Ci;
      const float* dst = (const float*)(base + w.ds_stats[i]);
      RC(vdk_layernorm_bwd(base + w.dhds, Ci, DT16, xprev, Ci, dst, dst + Rp, params + p.st[i].ds_nw, nullptr, 0, Rp, Ci, dxa, Ci, dxb, Ci,
                           grads + p.st[i].ds_nw, grads + p.st[i].ds_nb, lnws, w.lnws_bytes, s));
      if (on_ready) on_ready(user, p.st[i].ds_nw, p.st[i].blk[0].gamma - p.st[i].ds_nw);
    } else {
      // stem: LayerNorm2d backward (bf16 gradient of the conv output), then the conv's weight / bias gradient
      const float* st0 = (const float*)(base + w.stats0);
      RC(vdk_layernorm_bwd_deferred(dxa, C, VDK_F32, (const float*)(base + w.y0), C, st0, st0 + R, params + p.stem_nw, nullptr, 0, R, C, nullptr, 0, dh, C,
                                    grads + p.stem_nw, grads + p.stem_nb, lnws, w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf));
      RC(linear_wgrad(s, w, base, dh, (const bf16_t*)(base + w.patches), R, C, d.Kst, grads + p.stem_w, grads + p.stem_b));
      if (on_ready) on_ready(user, 0, p.st[0].blk[0].gamma);
    }
  }
  return vdk_check_launch("vdk_convnext_backward");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// PRECISE forward (evaluation / embedding extraction): fp32 activations, contractions on the fp32 MFMA (csrc/gemm_f32.hip), the
// depthwise conv and LayerNorm are fp32 already.  Reads the fp32 master weights (and the tap-major depthwise copy in `wx`).
namespace {
struct CnWsF32 { size_t total, big, xa, xb, t, h; };
void cn_plan_f32(const CnDims& d, CnWsF32* w) {
  size_t rc = 0, rm = 0, pk = (size_t)d.R[0] * d.Kst;
  for (int i = 0; i < 4; ++i) {
    const size_t r = (size_t)d.R[i] * d.C[i];
    if (r > rc) rc = r;
    if (4 * r > rm) rm = 4 * r;                      // fc1 output [R, 4C]; also covers the space-to-depth operand [R_i, 4 C_{i-1}]
  }
  if (pk > rm) rm = pk;
  size_t cur = 0;
  w->big = w_take(cur, rm * 4); w->xa = w_take(cur, rc * 4); w->xb = w_take(cur, rc * 4); w->t = w_take(cur, rc * 4); w->h = w_take(cur, rc * 4);
  w->total = cur;
}
int gemm32(hipStream_t s, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, const float* bias, const float* cscale,
           const float* res, int64_t ldr, int act) {
  VdkGemmF32Desc g = {};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.col_scale = cscale; g.residual = res; g.ldr = ldr;
  g.act = act; g.alpha = 1.0f;
  return vdk_gemm_f32_nt(&g, s);
}

// ---- fp32-class TRAINING path (feature mode): every activation f32, every contraction on the fp32 MFMA (vdk_gemm_f32_nt), library erff / expf -------------------------
// The reference runs the face / CBIR training loop WITHOUT autocast (engine/procedure/train.py:217-227): its arithmetic there is fp32.  This is the engine's mode with that
// arithmetic: forward as vdk_convnext_forward_f32 but keeping what the backward needs, backward with fp32 input- and weight-gradient GEMMs (k-major operands, the
// contraction over the rows split into slabs and summed in a fixed order).  About 6x slower than the bf16-operand step; parity with the fp32 oracle ~1e-5.
struct BlkT { size_t t, stats, h, u, g; };
struct TrainPlan32 {
  size_t total, patches, y0, stats0, X[4], ds_stats[4], ds_h[4], ds_A[4], head_stats;
  std::vector<BlkT> blk[4];
  size_t w2p, b2p, dxa, dxb, dt, du, dh, dA, dhds, dw2p, db2p, slabs, slabs_bytes, lnws, lnws_bytes, csws, csws_bytes, dwws, dwws_bytes;
};
int wgrad32_splits(int out, int in, long rows, long* kchunk) {
  const long tiles = (long)((out + 127) / 128) * ((in + 127) / 128);
  long S = (1024 + tiles - 1) / tiles;
  if (S > 64) S = 64;
  if (S > rows / 256) S = rows / 256;
  if (S < 1) S = 1;
  long kc = up((rows + S - 1) / S, 4);
  S = (rows + kc - 1) / kc;
  *kchunk = kc;
  return (int)S;
}
void cn_plan_train32(const CnDims& d, TrainPlan32* w) {
  size_t cur = 0, rc = 0, rm = 0, ra = 0, rh = 0, cm = 0, sl = 0, ln = 0, cs = 0, dww = 0;
  auto wg = [&](int out, int in, long rows) { long kc; const int S = wgrad32_splits(out, in, rows, &kc); const size_t b = (size_t)S * out * in * 4; if (b > sl) sl = b; };
  auto cw = [&](long rows, int n) { size_t c = 0; vdk_colsum_f32_workspace_bytes(rows, n, &c); if (c > cs) cs = c; };
  w->patches = w_take(cur, (size_t)d.R[0] * d.Kst * 4);
  w->y0 = w_take(cur, (size_t)d.R[0] * d.C[0] * 4);
  w->stats0 = w_take(cur, (size_t)d.R[0] * 2 * 4);
  wg(d.C[0], d.Kst, d.R[0]); cw(d.R[0], d.C[0]);
  { size_t l = 0; vdk_layernorm_bwd_workspace_bytes(d.R[0], d.C[0], &l); if (l > ln) ln = l; }
  for (int i = 0; i < 4; ++i) {
    const size_t R = d.R[i], C = d.C[i], M = 4 * C;
    w->ds_stats[i] = w->ds_h[i] = w->ds_A[i] = 0;
    if (i > 0) {
      const size_t Rp = d.R[i - 1], Ci = d.C[i - 1];
      w->ds_stats[i] = w_take(cur, Rp * 2 * 4); w->ds_h[i] = w_take(cur, Rp * Ci * 4); w->ds_A[i] = w_take(cur, R * 4 * Ci * 4);
      if (R * 4 * Ci > ra) ra = R * 4 * Ci;
      if (Rp * Ci > rh) rh = Rp * Ci;
      wg((int)C, (int)(4 * Ci), (long)R);
      size_t l = 0; vdk_layernorm_bwd_workspace_bytes((int)Rp, (int)Ci, &l); if (l > ln) ln = l;
    }
    w->X[i] = w_take(cur, (size_t)(d.depth[i] + 1) * R * C * 4);
    w->blk[i].resize(d.depth[i]);
    for (int j = 0; j < d.depth[i]; ++j) {
      BlkT& b = w->blk[i][j];
      b.t = w_take(cur, R * C * 4); b.stats = w_take(cur, R * 2 * 4); b.h = w_take(cur, R * C * 4); b.u = w_take(cur, R * M * 4); b.g = w_take(cur, R * M * 4);
    }
    if (R * C > rc) rc = R * C;
    if (R * M > rm) rm = R * M;
    if (C * M > cm) cm = C * M;
    wg((int)C, (int)M, (long)R); wg((int)M, (int)C, (long)R); cw((long)R, (int)M); cw((long)R, (int)C);
    size_t l = 0; vdk_layernorm_bwd_workspace_bytes((int)R, (int)C, &l); if (l > ln) ln = l;
    size_t q = 0; vdk_dwconv7_wgrad_workspace_bytes(d.B, d.H[i], d.H[i], (int)C, &q); if (q > dww) dww = q;
  }
  w->head_stats = w_take(cur, (size_t)d.R[3] * 2 * 4);
  { size_t l = 0; vdk_layernorm_bwd_workspace_bytes(d.R[3], d.C[3], &l); if (l > ln) ln = l; }
  w->w2p = w_take(cur, cm * 4); w->b2p = w_take(cur, 4096 * 4);
  w->dxa = w_take(cur, rc * 4); w->dxb = w_take(cur, rc * 4); w->dt = w_take(cur, rc * 4); w->du = w_take(cur, rm * 4); w->dh = w_take(cur, rc * 4);
  w->dA = w_take(cur, ra * 4 + 256); w->dhds = w_take(cur, rh * 4 + 256);
  w->dw2p = w_take(cur, cm * 4); w->db2p = w_take(cur, 4096 * 4);
  w->slabs_bytes = sl; w->slabs = w_take(cur, sl + 256);
  w->lnws_bytes = ln; w->lnws = w_take(cur, ln + 256);
  w->csws_bytes = cs; w->csws = w_take(cur, cs + 256);
  w->dwws_bytes = dww; w->dwws = w_take(cur, dww + 256);
  w->total = cur;
}
// dX[rows, in] = dY[rows, out] . W[out, in]   (W read as it lies: k-major B)
int dgrad32(hipStream_t s, const float* dY, int out, const float* W, int in, float* dX, int rows) {
  VdkGemmF32Desc g = {};
  g.A = dY; g.lda = out; g.B = W; g.ldb = in; g.b_kmajor = 1; g.C = dX; g.ldc = in; g.M = rows; g.N = in; g.K = out; g.alpha = 1.0f;
  return vdk_gemm_f32_nt(&g, s);
}
// dW[out, in] = dY[rows, out]^T . X[rows, in]   (both operands k-major; the contraction over the rows split into slabs, summed in slab order)
int wgrad32(hipStream_t s, const float* dY, int out, const float* X, int in, long rows, float* dW, float* slabs, size_t slabs_bytes) {
  long kc; const int S = wgrad32_splits(out, in, rows, &kc);
  if (S > 1 && (size_t)S * out * in * 4 > slabs_bytes) return vdk_fail(VDK_EWORKSPACE, "convnext fp32 training: weight-gradient slabs");
  VdkGemmF32Desc g = {};
  g.A = dY; g.lda = out; g.a_kmajor = 1; g.B = X; g.ldb = in; g.b_kmajor = 1; g.C = S > 1 ? slabs : dW; g.ldc = in; g.M = out; g.N = in; g.K = (int)kc; g.alpha = 1.0f;
  g.batch1 = S; g.batch2 = 1; g.sa1 = kc * out; g.sb1 = kc * in; g.sc1 = (int64_t)out * in; g.k_total = rows;
  RC(vdk_gemm_f32_nt(&g, s));
  if (S > 1) RC(vdk_reduce_rows_f32(slabs, (int64_t)out * in, S, (int64_t)out * in, dW, 1.0f, s));
  return VDK_OK;
}
}  // namespace

extern "C" {

int vdk_convnext_workspace_f32_bytes(const VdkConvNextConfig* cfg, size_t* bytes) {
  CnDims d; RC(cn_dims(cfg, &d));
  if (!bytes) return vdk_fail(VDK_EINVAL, "null");
  CnWsF32 w; cn_plan_f32(d, &w);
  *bytes = w.total;
  return VDK_OK;
}

int vdk_convnext_forward_f32(const VdkConvNextConfig* cfg, const float* x, const float* params, const void* wx, void* ws, size_t ws_bytes, float* out,
                             void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  CnDims d; RC(cn_dims(cfg, &d));
  PLayout p; cn_layout(d, &p);
  XLayout xl; cn_xlayout(d, &xl);
  CnWsF32 w; cn_plan_f32(d, &w);
  if (!x || !params || !wx || !ws || !out) return vdk_fail(VDK_EINVAL, "vdk_convnext_forward_f32: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_convnext_forward_f32: workspace too small");
  char* base = (char*)ws;
  const char* xb_ = (const char*)wx;
  float* big = (float*)(base + w.big); float* xa = (float*)(base + w.xa); float* xb = (float*)(base + w.xb); float* t = (float*)(base + w.t);
  float* h = (float*)(base + w.h);
  // stem
  RC(vdk_patchify_f32(x, d.B, d.Cin, d.img, d.img, 4, big, s));
  RC(gemm32(s, big, d.Kst, params + p.stem_w, d.Kst, t, d.C[0], d.R[0], d.C[0], d.Kst, params + p.stem_b, nullptr, nullptr, 0, VDK_ACT_NONE));
  RC(vdk_layernorm_fwd(t, d.C[0], d.R[0], d.C[0], params + p.stem_nw, params + p.stem_nb, d.eps, xa, d.C[0], VDK_F32, nullptr, nullptr, s));
  for (int i = 0; i < 4; ++i) {
    const int R = d.R[i], C = d.C[i], M = 4 * C, H = d.H[i];
    if (i > 0) {
      const int Rp = d.R[i - 1], Ci = d.C[i - 1];
      RC(vdk_layernorm_fwd(xa, Ci, Rp, Ci, params + p.st[i].ds_nw, params + p.st[i].ds_nb, d.eps, h, Ci, VDK_F32, nullptr, nullptr, s));
      RC(vdk_space_to_depth2_f32(h, big, d.B, d.H[i - 1], d.H[i - 1], Ci, s));
      RC(gemm32(s, big, 4 * Ci, params + p.st[i].ds_w, 4 * Ci, xa, C, R, C, 4 * Ci, params + p.st[i].ds_b, nullptr, nullptr, 0, VDK_ACT_NONE));
    }
    for (int j = 0; j < d.depth[i]; ++j) {
      const BlkP& b = p.st[i].blk[j]; const BlkX& bx = xl.blk[i][j];
      RC(vdk_dwconv7_fwd(xa, (const float*)(xb_ + bx.dwt), params + b.dw_b, nullptr, t, nullptr, d.B, H, H, C, 0, s));
      RC(vdk_layernorm_fwd(t, C, R, C, params + b.nw, params + b.nb, d.eps, h, C, VDK_F32, nullptr, nullptr, s));
      RC(gemm32(s, h, C, params + b.fc1_w, C, big, M, R, M, C, params + b.fc1_b, nullptr, nullptr, 0, VDK_ACT_GELU));
      RC(gemm32(s, big, M, params + b.fc2_w, M, xb, C, R, C, M, params + b.fc2_b, params + b.gamma, xa, C, VDK_ACT_NONE));
      float* sw = xa; xa = xb; xb = sw;
    }
  }
  RC(vdk_layernorm_fwd(xa, d.C[3], d.R[3], d.C[3], params + p.head_nw, params + p.head_nb, d.eps, out, d.C[3], VDK_F32, nullptr, nullptr, s));
  return vdk_check_launch("vdk_convnext_forward_f32");
}

int vdk_convnext_train_f32_workspace_bytes(const VdkConvNextConfig* cfg, size_t* bytes) {
  CnDims d; RC(cn_dims(cfg, &d));
  if (!bytes) return vdk_fail(VDK_EINVAL, "null");
  if (d.ncls > 0) return vdk_fail(VDK_EUNSUPPORTED, "vdk_convnext_train_f32: feature mode only (num_classes = 0)");
  TrainPlan32 w; cn_plan_train32(d, &w);
  *bytes = w.total;
  return VDK_OK;
}

// training forward in fp32-class arithmetic: out f32 [B * (img/32)^2, dims[3]]; ws keeps the activations for vdk_convnext_backward_train_f32
int vdk_convnext_forward_train_f32(const VdkConvNextConfig* cfg, const float* x, const float* params, const void* wx, void* ws, size_t ws_bytes, float* out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  CnDims d; RC(cn_dims(cfg, &d));
  if (d.ncls > 0) return vdk_fail(VDK_EUNSUPPORTED, "vdk_convnext_forward_train_f32: feature mode only");
  PLayout p; cn_layout(d, &p);
  XLayout xl; cn_xlayout(d, &xl);
  TrainPlan32 w; cn_plan_train32(d, &w);
  if (!x || !params || !wx || !ws || !out) return vdk_fail(VDK_EINVAL, "vdk_convnext_forward_train_f32: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_convnext_forward_train_f32: workspace too small");
  char* base = (char*)ws;
  const char* xb_ = (const char*)wx;
  float* patches = (float*)(base + w.patches); float* y0 = (float*)(base + w.y0); float* st0 = (float*)(base + w.stats0);
  float* w2p = (float*)(base + w.w2p); float* b2p = (float*)(base + w.b2p);
  RC(vdk_patchify_f32(x, d.B, d.Cin, d.img, d.img, 4, patches, s));
  RC(gemm32(s, patches, d.Kst, params + p.stem_w, d.Kst, y0, d.C[0], d.R[0], d.C[0], d.Kst, params + p.stem_b, nullptr, nullptr, 0, VDK_ACT_NONE));
  RC(vdk_layernorm_fwd(y0, d.C[0], d.R[0], d.C[0], params + p.stem_nw, params + p.stem_nb, d.eps, base + w.X[0], d.C[0], VDK_F32, st0, st0 + d.R[0], s));
  for (int i = 0; i < 4; ++i) {
    const int R = d.R[i], C = d.C[i], M = 4 * C, H = d.H[i];
    float* X = (float*)(base + w.X[i]);
    const size_t XS = (size_t)R * C;
    if (i > 0) {
      const int Rp = d.R[i - 1], Ci = d.C[i - 1];
      const float* xprev = (const float*)(base + w.X[i - 1]) + (size_t)d.depth[i - 1] * Rp * Ci;
      float* dst = (float*)(base + w.ds_stats[i]);
      RC(vdk_layernorm_fwd(xprev, Ci, Rp, Ci, params + p.st[i].ds_nw, params + p.st[i].ds_nb, d.eps, base + w.ds_h[i], Ci, VDK_F32, dst, dst + Rp, s));
      RC(vdk_space_to_depth2_f32((const float*)(base + w.ds_h[i]), (float*)(base + w.ds_A[i]), d.B, d.H[i - 1], d.H[i - 1], Ci, s));
      RC(gemm32(s, (const float*)(base + w.ds_A[i]), 4 * Ci, params + p.st[i].ds_w, 4 * Ci, X, C, R, C, 4 * Ci, params + p.st[i].ds_b, nullptr, nullptr, 0, VDK_ACT_NONE));
    }
    for (int j = 0; j < d.depth[i]; ++j) {
      const BlkP& b = p.st[i].blk[j]; const BlkX& bx = xl.blk[i][j]; const BlkT& bw = w.blk[i][j];
      float* xin = X + (size_t)j * XS; float* xout = xin + XS;
      float* t = (float*)(base + bw.t); float* st = (float*)(base + bw.stats); float* h = (float*)(base + bw.h); float* u = (float*)(base + bw.u); float* g = (float*)(base + bw.g);
      RC(vdk_dwconv7_fwd(xin, (const float*)(xb_ + bx.dwt), params + b.dw_b, nullptr, t, nullptr, d.B, H, H, C, 0, s));
      RC(vdk_layernorm_fwd(t, C, R, C, params + b.nw, params + b.nb, d.eps, h, C, VDK_F32, st, st + R, s));
      RC(gemm32(s, h, C, params + b.fc1_w, C, u, M, R, M, C, params + b.fc1_b, nullptr, nullptr, 0, VDK_ACT_NONE));
      RC(vdk_gelu_f32(u, g, (int64_t)R * M, s));
      // layer scale folded into fc2 (W2' = gamma (.) W2, b2' = gamma (.) b2), as in the bf16 engine: the backward then never divides by gamma
      RC(vdk_rowscale_f32(params + b.fc2_w, params + b.gamma, w2p, C, M, s));
      RC(vdk_rowscale_f32(params + b.fc2_b, params + b.gamma, b2p, C, 1, s));
      RC(gemm32(s, g, M, w2p, M, xout, C, R, C, M, b2p, nullptr, xin, C, VDK_ACT_NONE));
    }
  }
  const float* xlast = (const float*)(base + w.X[3]) + (size_t)d.depth[3] * d.R[3] * d.C[3];
  float* hs = (float*)(base + w.head_stats);
  RC(vdk_layernorm_fwd(xlast, d.C[3], d.R[3], d.C[3], params + p.head_nw, params + p.head_nb, d.eps, out, d.C[3], VDK_F32, hs, hs + d.R[3], s));
  return vdk_check_launch("vdk_convnext_forward_train_f32");
}

// dout f32 [B * (img/32)^2, dims[3]] -> grads (flat f32, param layout, fully overwritten).  on_ready: see vdk_vit_backward.
int vdk_convnext_backward_train_f32(const VdkConvNextConfig* cfg, const float* dout, const float* params, const void* wx, void* ws, size_t ws_bytes, float* grads,
                                    vdk_grad_ready_fn on_ready, void* user, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  CnDims d; RC(cn_dims(cfg, &d));
  if (d.ncls > 0) return vdk_fail(VDK_EUNSUPPORTED, "vdk_convnext_backward_train_f32: feature mode only");
  PLayout p; cn_layout(d, &p);
  XLayout xl; cn_xlayout(d, &xl);
  TrainPlan32 w; cn_plan_train32(d, &w);
  if (!dout || !params || !wx || !ws || !grads) return vdk_fail(VDK_EINVAL, "vdk_convnext_backward_train_f32: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_convnext_backward_train_f32: workspace too small");
  char* base = (char*)ws;
  const char* xb_ = (const char*)wx;
  float* dxa = (float*)(base + w.dxa); float* dxb = (float*)(base + w.dxb); float* dt = (float*)(base + w.dt); float* du = (float*)(base + w.du); float* dh = (float*)(base + w.dh);
  float* w2p = (float*)(base + w.w2p); float* dw2p = (float*)(base + w.dw2p); float* db2p = (float*)(base + w.db2p);
  float* slabs = (float*)(base + w.slabs); void* lnws = base + w.lnws; void* csws = base + w.csws;
  {
    const float* xlast = (const float*)(base + w.X[3]) + (size_t)d.depth[3] * d.R[3] * d.C[3];
    const float* hs = (const float*)(base + w.head_stats);
    RC(vdk_layernorm_bwd(dout, d.C[3], VDK_F32, xlast, d.C[3], hs, hs + d.R[3], params + p.head_nw, nullptr, 0, d.R[3], d.C[3], dxa, d.C[3], nullptr, 0,
                         grads + p.head_nw, grads + p.head_nb, lnws, w.lnws_bytes, s));
    if (on_ready) on_ready(user, p.head_nw, p.total - p.head_nw);
  }
  for (int i = 3; i >= 0; --i) {
    const int R = d.R[i], C = d.C[i], M = 4 * C, H = d.H[i];
    const float* X = (const float*)(base + w.X[i]);
    const size_t XS = (size_t)R * C;
    if (C > 4096) return vdk_fail(VDK_EUNSUPPORTED, "vdk_convnext_backward_train_f32: dims <= 4096");
    for (int j = d.depth[i] - 1; j >= 0; --j) {
      const BlkP& b = p.st[i].blk[j]; const BlkX& bx = xl.blk[i][j]; const BlkT& bw = w.blk[i][j];
      const float* xin = X + (size_t)j * XS;
      const float* st = (const float*)(base + bw.stats);
      const float* t = (const float*)(base + bw.t); const float* h = (const float*)(base + bw.h); const float* u = (const float*)(base + bw.u); const float* g = (const float*)(base + bw.g);
      // dxa = dL/d(block output).  MLP branch through the folded fc2
      RC(vdk_rowscale_f32(params + b.fc2_w, params + b.gamma, w2p, C, M, s));
      RC(vdk_colsum_f32(dxa, C, R, C, db2p, csws, w.csws_bytes, s));
      RC(dgrad32(s, dxa, C, w2p, M, du, R));                                             // dL/dg
      RC(wgrad32(s, dxa, C, g, M, R, dw2p, slabs, w.slabs_bytes));
      RC(vdk_layerscale_grad(dw2p, db2p, params + b.fc2_w, params + b.fc2_b, params + b.gamma, grads + b.fc2_w, grads + b.fc2_b, grads + b.gamma, C, M, s));
      RC(vdk_dgelu_f32(du, u, (int64_t)R * M, s));                                       // dL/du
      RC(vdk_colsum_f32(du, M, R, M, grads + b.fc1_b, csws, w.csws_bytes, s));
      RC(dgrad32(s, du, M, params + b.fc1_w, C, dh, R));                                 // dL/dh
      RC(wgrad32(s, du, M, h, C, R, grads + b.fc1_w, slabs, w.slabs_bytes));
      RC(vdk_layernorm_bwd(dh, C, VDK_F32, t, C, st, st + R, params + b.nw, nullptr, 0, R, C, dt, C, nullptr, 0, grads + b.nw, grads + b.nb, lnws, w.lnws_bytes, s));
      RC(vdk_dwconv7_wgrad(xin, dt, grads + b.dw_w, grads + b.dw_b, d.B, H, H, C, base + w.dwws, w.dwws_bytes, s));
      RC(vdk_dwconv7_fwd(dt, (const float*)(xb_ + bx.dwt), nullptr, dxa, dxb, nullptr, d.B, H, H, C, 1, s));   // + the shortcut's gradient
      { float* sw = dxa; dxa = dxb; dxb = sw; }
      if (on_ready) on_ready(user, b.gamma, (j + 1 < d.depth[i] ? p.st[i].blk[j + 1].gamma : (i < 3 ? p.st[i + 1].ds_nw : p.head_nw)) - b.gamma);
    }
    if (i > 0) {
      // downsample backward: dxa = dL/dX[i][0] [R, C]
      const int Rp = d.R[i - 1], Ci = d.C[i - 1];
      const float* xprev = (const float*)(base + w.X[i - 1]) + (size_t)d.depth[i - 1] * Rp * Ci;
      const float* dst = (const float*)(base + w.ds_stats[i]);
      float* dA = (float*)(base + w.dA); float* dhds = (float*)(base + w.dhds);
      RC(vdk_colsum_f32(dxa, C, R, C, grads + p.st[i].ds_b, csws, w.csws_bytes, s));
      RC(dgrad32(s, dxa, C, params + p.st[i].ds_w, 4 * Ci, dA, R));
      RC(wgrad32(s, dxa, C, (const float*)(base + w.ds_A[i]), 4 * Ci, R, grads + p.st[i].ds_w, slabs, w.slabs_bytes));
      RC(vdk_depth_to_space2_f32(dA, dhds, d.B, d.H[i - 1], d.H[i - 1], Ci, s));
      RC(vdk_layernorm_bwd(dhds, Ci, VDK_F32, xprev, Ci, dst, dst + Rp, params + p.st[i].ds_nw, nullptr, 0, Rp, Ci, dxb, Ci, nullptr, 0, grads + p.st[i].ds_nw,
                           grads + p.st[i].ds_nb, lnws, w.lnws_bytes, s));
      { float* sw = dxa; dxa = dxb; dxb = sw; }
      if (on_ready) on_ready(user, p.st[i].ds_nw, p.st[i].blk[0].gamma - p.st[i].ds_nw);
    } else {
      // stem: X[0][0] = LayerNorm(y0), y0 = patches . Wst^T + b
      const float* st0 = (const float*)(base + w.stats0);
      RC(vdk_layernorm_bwd(dxa, d.C[0], VDK_F32, (const float*)(base + w.y0), d.C[0], st0, st0 + d.R[0], params + p.stem_nw, nullptr, 0, d.R[0], d.C[0], dxb, d.C[0], nullptr, 0,
                           grads + p.stem_nw, grads + p.stem_nb, lnws, w.lnws_bytes, s));
      RC(vdk_colsum_f32(dxb, d.C[0], d.R[0], d.C[0], grads + p.stem_b, csws, w.csws_bytes, s));
      RC(wgrad32(s, dxb, d.C[0], (const float*)(base + w.patches), d.Kst, d.R[0], grads + p.stem_w, slabs, w.slabs_bytes));
      if (on_ready) on_ready(user, 0, p.st[0].blk[0].gamma);
    }
  }
  return vdk_check_launch("vdk_convnext_backward_train_f32");
}

}  // extern "C"

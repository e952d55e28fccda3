// swin_engine.hip — native orchestration of hot path A for timm's SwinTransformer (`swin_{tiny,small,base,large}_patch4_window7_224`), the DEFAULT backbone of both
// shipped configs of the reference (configs/classification/pet.yaml:25, configs/faceX/cbir.yaml:26; built by timm.create_model in models/classifier/classify_model.py:49-54
// and models/faceX/backbone/timm_wrapper.py:16-21, stepped by engine/procedure/train.py:177-215).  One C call runs the whole forward, one the whole backward, as a fixed
// sequence of the library's own kernels on ONE stream over caller-owned flat buffers, like vit_engine.hip: no autograd graph, no per-tensor launches, no allocation, no
// weight casts inside the step (round 3 ran this family as ~35 autograd nodes per block with torch SGD: 220 row-reduction launches, 100 weight casts and 100 weight
// transposes per step, ~1 600 at::native launches per 7 steps in the kernel trace).
//
// Architecture restated from timm 0.9.16 (oracle/swin_ref.py, pinned against transformers.SwinModel): PatchEmbed (4x4 / stride 4 convolution = a Linear over (c, ky, kx)
// patches, LayerNorm), four stages of pre-norm blocks x -> x + proj(W-MSA(norm1(x))) -> + fc2(gelu(fc1(norm2(.)))) with 7x7 windows, relative-position bias and a cyclic
// shift of 3 in every second block (none once the map is a single window), PatchMerging (2x2 neighbourhood -> LayerNorm(4C) -> Linear(4C -> 2C, no bias)) IN FRONT of
// stages 1..3, final LayerNorm, global average pool, Linear head.
//
// Data layout (everything in HBM):
//   params / grads / momentum / ema : flat fp32, identical offsets, timm state_dict order (vdk_swin_param_info); wb16 = same layout in bf16 (GEMM B operands, refreshed by
//                                     vdk_sgd_step), wt16 = per-Linear [in, out] bf16 copies for the input-gradient GEMMs.
//   token rows                      : [B * H * W, C] in IMAGE order in every stage; the (shifted) window partition is a row index the attention kernels follow
//                                     (window_attention.hip `rowidx`), built once per call into the workspace together with the shift masks.
//   residual stream                 : fp32, one buffer per block boundary; GEMM operands / saved activations bf16 (u = pre-GELU, g = post-GELU, qkv, o, h1, h2).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_gemm.h"

// window_attention.hip, in-library forms: bias tile prepared once per block and step from the table, kept for the backward; d(table) from the fragment-order d(bias)
size_t vdk_wa_bm_bytes(int32_t nW, int32_t H);
size_t vdk_wa_bwd_scratch_bytes(int64_t windows, int32_t H);
int vdk_wa_fwd_bm(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, const float* bm, int32_t nWm, int64_t windows, int32_t H, float scale, const int32_t* rowidx, int opf,
                  void* stream);
int vdk_wa_bwd_bm(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, const float* bm, int32_t nWm, int64_t windows, int32_t H, float scale,
                  const int32_t* rowidx, void* dqkv, int64_t ldd, void* scratch, size_t scratch_bytes, const int32_t* uses, int32_t R, int32_t U, float* dtable, int opf, void* stream);
extern "C" {
int vdk_layernorm_fwd(const float*, int64_t, int32_t, int32_t, const float*, const float*, float, void*, int64_t, int32_t, float*, float*, void*);
int vdk_layernorm_bwd_workspace_bytes(int32_t, int32_t, size_t*);
int vdk_colsum_bf16_workspace_bytes(int32_t, int32_t, size_t*);
int vdk_reduce_rows_f32(const float*, int64_t, int32_t, int64_t, float*, float, void*);
int vdk_avgpool_rows_f32_fwd(const float*, float*, int32_t, int32_t, int32_t, void*);
int vdk_avgpool_rows_f32_bwd(const float*, float*, void*, int32_t, int32_t, int32_t, void*);
int vdk_gemm_c_colsum_rows(int32_t, int32_t, int32_t);
}

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
static inline int64_t up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
#define SW_WS 7
#define SW_N 49
#define SW_NREL 169      /* (2 * 7 - 1)^2 relative offsets */

namespace {

// operand format of the running call (VdkSwinConfig.operand): bf16, or fp16 = the reference's autocast dtype (engine/procedure/train.py:118); set at every entry point, as in
// vit_engine.hip.  DT16 = the dtype code of the 16-bit tensors.
thread_local int t_opf = VDK_OPF_BF16;
#define DT16 (t_opf ? VDK_F16 : VDK_BF16)
// the MLP's saved tensor `u`: the pre-activation, or (default with fp16 operands, as in vit_engine.hip / convnext_engine.hip; VDK_SWIN_GELU_SAVED_GRAD=0 / 1 forces it) fp16
// GELU'(pre-activation) written by the fc1 epilogue, which turns the dfc2 epilogue into one multiplication
bool sw_saved_grad() {
  static const int v = [] { const char* e = getenv("VDK_SWIN_GELU_SAVED_GRAD"); return e ? atoi(e) : -1; }();
  return v < 0 ? t_opf != 0 : v != 0;
}

struct SwDims {
  int B, img, Cin, E, depth[4], heads[4], dim[4], res[4], nblk;
  int nst;                 // stages in use: the leading non-zero entries of depths (timm's family has 4; shallower members serve the tests)
  long T[4];               // token rows per stage = B * res^2
  int C, Cp, Bp, Kpe;      // classes (0: feature mode), padded to 8; batch padded to 64; K of the patch-embedding GEMM (in_chans * 16)
  float eps;
  int opf;                 // VDK_OPF_BF16 | VDK_OPF_F16
  const float* dp;         // VdkSwinConfig.drop_path: f32 [2 * nblk][B] per-sample branch factors (stochastic depth), or nullptr
};
int sw_dims(const VdkSwinConfig* c, SwDims* d) {
  if (!c) return vdk_fail(VDK_EINVAL, "swin: null config");
  if (c->batch <= 0 || c->img_size <= 0 || (c->img_size % 224) || c->in_chans <= 0 || c->embed_dim <= 0 || (c->embed_dim % 32) || c->num_classes < 0)
    return vdk_fail(VDK_EINVAL, "swin: bad config (img_size % 224 == 0: 7 x 7 windows on every stage's map; embed_dim % 32 == 0)");
  if ((c->in_chans * 16) & 7) return vdk_fail(VDK_EUNSUPPORTED, "swin: in_chans * 16 must be a multiple of 8");
  if (c->operand != VDK_BF16 && c->operand != VDK_F16) return vdk_fail(VDK_EINVAL, "swin: operand must be VDK_BF16 or VDK_F16");
  d->opf = c->operand == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16;
  d->dp = c->drop_path;
  d->B = c->batch; d->img = c->img_size; d->Cin = c->in_chans; d->E = c->embed_dim; d->eps = c->ln_eps; d->nblk = 0;
  d->C = c->num_classes; d->Cp = (int)up(c->num_classes, 8); d->Bp = (int)up(c->batch, 64); d->Kpe = c->in_chans * 16;
  int res = c->img_size / 4;
  d->nst = 0;
  while (d->nst < 4 && c->depths[d->nst] > 0) ++d->nst;
  for (int i = d->nst; i < 4; ++i) if (c->depths[i] != 0) return vdk_fail(VDK_EINVAL, "swin: depths must be a run of positive entries followed by zeros");
  if (d->nst == 0) return vdk_fail(VDK_EINVAL, "swin: depths[0] must be positive");
  for (int i = 0; i < 4; ++i) { d->depth[i] = 0; d->heads[i] = 0; d->dim[i] = 0; d->res[i] = 0; d->T[i] = 0; }
  for (int i = 0; i < d->nst; ++i) {
    if (c->depths[i] <= 0 || c->depths[i] > 64 || c->heads[i] <= 0) return vdk_fail(VDK_EINVAL, "swin: bad depths / heads");
    d->depth[i] = c->depths[i]; d->heads[i] = c->heads[i]; d->dim[i] = c->embed_dim << i;
    if (d->dim[i] != d->heads[i] * 32) return vdk_fail(VDK_EUNSUPPORTED, "swin: the window attention kernels are built for head dim 32 (dim == 32 * heads in every stage)");
    if (i > 0) res /= 2;
    if (res < SW_WS || res % SW_WS) return vdk_fail(VDK_EINVAL, "swin: every stage's map must be a multiple of the 7 x 7 window");
    d->res[i] = res;
    const int64_t t = (int64_t)c->batch * res * res;
    if (t * d->dim[i] * 4 > 0x7fffffffLL * 8) return vdk_fail(VDK_EINVAL, "swin: batch x resolution too large");
    d->T[i] = (long)t;
    d->nblk += c->depths[i];
  }
  return VDK_OK;
}

// ---------------------------------------------------------------------------------------------------------------- parameter layout
struct PEntry { char name[80]; int64_t off, numel; int64_t shape[4]; int ndim; };
struct BlkP { int64_t n1w, n1b, table, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b; int64_t tq, tp, t1, t2; };   // t*: offsets into the transposed-copy buffer
struct StageP { int64_t ds_nw, ds_nb, ds_w, ds_t; std::vector<BlkP> blk; };
struct PLayout {
  int64_t pe_w, pe_b, pe_nw, pe_nb, norm_w, norm_b, fc_w, fc_b, fc_t, total, totalT;
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
// timm state_dict order and names (0.9.16: `layers.i.downsample` sits in front of stage i's blocks)
void sw_layout(const SwDims& d, PLayout* p) {
  int64_t cur = 0, t = 0;
  char nm[80];
  p->entries.clear();
  p->pe_w = p_take(cur, (int64_t)d.E * d.Kpe); add_entry(p, "patch_embed.proj.weight", p->pe_w, 4, d.E, d.Cin, 4, 4);
  p->pe_b = p_take(cur, d.E); add_entry(p, "patch_embed.proj.bias", p->pe_b, 1, d.E);
  p->pe_nw = p_take(cur, d.E); add_entry(p, "patch_embed.norm.weight", p->pe_nw, 1, d.E);
  p->pe_nb = p_take(cur, d.E); add_entry(p, "patch_embed.norm.bias", p->pe_nb, 1, d.E);
  for (int i = 0; i < d.nst; ++i) {
    StageP& s = p->st[i];
    const int C = d.dim[i], M = 4 * C, H = d.heads[i];
    s.ds_nw = s.ds_nb = s.ds_w = s.ds_t = -1;
    if (i > 0) {
      const int Ci = d.dim[i - 1];
      s.ds_nw = p_take(cur, 4 * Ci); snprintf(nm, 80, "layers.%d.downsample.norm.weight", i); add_entry(p, nm, s.ds_nw, 1, 4 * Ci);
      s.ds_nb = p_take(cur, 4 * Ci); snprintf(nm, 80, "layers.%d.downsample.norm.bias", i); add_entry(p, nm, s.ds_nb, 1, 4 * Ci);
      s.ds_w = p_take(cur, (int64_t)C * 4 * Ci); snprintf(nm, 80, "layers.%d.downsample.reduction.weight", i); add_entry(p, nm, s.ds_w, 2, C, 4 * Ci);
      s.ds_t = p_take(t, (int64_t)4 * Ci * C);
    }
    s.blk.resize(d.depth[i]);
    for (int j = 0; j < d.depth[i]; ++j) {
      BlkP& b = s.blk[j];
#define SW_ENT(field, suffix, n, ...) b.field = p_take(cur, (n)); snprintf(nm, 80, "layers.%d.blocks.%d." suffix, i, j); add_entry(p, nm, b.field, __VA_ARGS__)
      SW_ENT(n1w, "norm1.weight", C, 1, C);
      SW_ENT(n1b, "norm1.bias", C, 1, C);
      SW_ENT(table, "attn.relative_position_bias_table", (int64_t)SW_NREL * H, 2, SW_NREL, H);
      SW_ENT(qkv_w, "attn.qkv.weight", (int64_t)3 * C * C, 2, 3 * C, C);
      SW_ENT(qkv_b, "attn.qkv.bias", 3 * C, 1, 3 * C);
      SW_ENT(proj_w, "attn.proj.weight", (int64_t)C * C, 2, C, C);
      SW_ENT(proj_b, "attn.proj.bias", C, 1, C);
      SW_ENT(n2w, "norm2.weight", C, 1, C);
      SW_ENT(n2b, "norm2.bias", C, 1, C);
      SW_ENT(fc1_w, "mlp.fc1.weight", (int64_t)M * C, 2, M, C);
      SW_ENT(fc1_b, "mlp.fc1.bias", M, 1, M);
      SW_ENT(fc2_w, "mlp.fc2.weight", (int64_t)C * M, 2, C, M);
      SW_ENT(fc2_b, "mlp.fc2.bias", C, 1, C);
#undef SW_ENT
      b.tq = p_take(t, (int64_t)C * 3 * C); b.tp = p_take(t, (int64_t)C * C); b.t1 = p_take(t, (int64_t)C * M); b.t2 = p_take(t, (int64_t)M * C);
    }
  }
  p->norm_w = p_take(cur, d.dim[d.nst - 1]); add_entry(p, "norm.weight", p->norm_w, 1, d.dim[d.nst - 1]);
  p->norm_b = p_take(cur, d.dim[d.nst - 1]); add_entry(p, "norm.bias", p->norm_b, 1, d.dim[d.nst - 1]);
  p->fc_w = p->fc_b = p->fc_t = 0;
  if (d.C > 0) {          // rows >= C of the stored [Cp, D] weight are padding (zero gradient, untouched by state_dict I/O)
    p->fc_w = p_take(cur, (int64_t)d.Cp * d.dim[d.nst - 1]); add_entry(p, "head.fc.weight", p->fc_w, 2, d.C, d.dim[d.nst - 1]);
    p->fc_b = p_take(cur, d.Cp); add_entry(p, "head.fc.bias", p->fc_b, 1, d.C);
    p->fc_t = p_take(t, (int64_t)d.dim[d.nst - 1] * d.Cp);
  }
  p->total = cur; p->totalT = t;
}

// ---------------------------------------------------------------------------------------------------------------- workspace plan
size_t w_take(size_t& cur, size_t n) { size_t o = cur; cur = (cur + n + 255) & ~(size_t)255; return o; }
struct BlkW { size_t x, xmid, stats, h1, qkv, lse, o, h2, u, g, bias; };
struct StageW { size_t rowidx0, rowidx3, mask, mg, mstats, mh, xout; std::vector<BlkW> blk; };
struct WsPlan {
  size_t total;
  size_t uses;                            // int32 [169][49]: where each relative-position table entry is used (for the table gradient)
  size_t patches, petmp, pestats;         // bf16 [T0, Kpe]; f32 [T0, E] (the patch projection before its LayerNorm); f32 2 x [T0]
  StageW st[4];
  size_t fmap, fstats, pooled, hf;        // f32 [T3, D]; 2 x [T3]; f32 [B, D]; bf16 [Bp, D]
  // backward scratch (sized for the largest stage)
  size_t dxa, dxm, dxab, dxmb, dsm, du, dqkv, dpool, dhf;
  size_t tA, tB, slabs, slabs_bytes, lnws, lnws_bytes, csws, csws_bytes, waws, waws_bytes;
  long tcols; size_t trows;
};
int wgrad_splitk_tn(int M, int N, int K) {      // (vit_engine.hip's rule: tiles x splits ~ one round of 256 CUs, >= 4 k-tiles per split)
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  int s = 256 / tiles;
  if (s < 1) s = 1;
  if (tiles >= 4 && s > 32) s = 32;       // (tools/bench_gemm_swin.py: 512 x 512 over 25 088 rows 41 -> 38 us, 1024 x 256 over 100 352 rows 93 -> 82 us: the slabs' write + re-read)
  const int kt = K / 64;
  if (s > kt / 4) s = kt / 4;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}
// ... and which of them run on the 256x128 two-workgroup kernel with the split count of ITS tiles (tools/bench_gemm_swin.py, swin_base at batch 128; us, four-wave at its
// rule / half-tile form): outputs of at most one whole tile's worth of elements, where a 256x256 tile is mostly padding (128 x 128 over 401 408 rows 117 / 76, 384 x 128
// 140 / 125, 256 x 256 over 100 352 rows 52 / 45), and outputs so large that the slabs' write + re-read weighs more than the parallelism of many splits (2048 x 512 over
// 25 088 rows 66 / 59, 4096 x 1024 over 6 272 rows 67 / 59, 3072 x 1024 56 / 49); in between the four-wave kernel keeps them (1536 x 512 56 / 57, 1024 x 256 81 / 110)
bool wgrad_tn_half(int M, int N) {
  const long mn = (long)M * N;
  return mn <= 65536 || mn >= (1L << 20);
}
int wgrad_splitk_tn_half(int M, int N, int K) {
  const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
  int s = 256 / tiles;
  const int kt = K / 64;
  if (s > kt / 4) s = kt / 4;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}
int wgrad_tn_splits(int M, int N, int K) { return wgrad_tn_half(M, N) ? wgrad_splitk_tn_half(M, N, K) : wgrad_splitk_tn(M, N, K); }
int wgrad_splitk(int M, int N, int K) {
  if (K < 4096) return 1;
  int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int s = (1024 + tiles - 1) / tiles;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return s;
}
void sw_plan(const SwDims& d, WsPlan* w) {
  size_t cur = 0;
  w->uses = w_take(cur, (size_t)SW_NREL * SW_N * 4);
  w->patches = w_take(cur, (size_t)d.T[0] * d.Kpe * 2);
  w->petmp = w_take(cur, (size_t)d.T[0] * d.E * 4);
  w->pestats = w_take(cur, (size_t)d.T[0] * 2 * 4);
  size_t maxTD = 0, maxTM = 0, maxT3 = 0, sl = 0, trows = 0;
  long tcols = d.Bp;
  size_t lnmax = 0, csmax = 0, wamax = 0;
  for (int i = 0; i < d.nst; ++i) {
    StageW& s = w->st[i];
    const size_t T = (size_t)d.T[i], C = d.dim[i], M = 4 * C, H = d.heads[i];
    const int nW = (d.res[i] / SW_WS) * (d.res[i] / SW_WS);
    s.rowidx0 = w_take(cur, T * 4);
    s.rowidx3 = s.mask = 0;
    if (d.res[i] > SW_WS) { s.rowidx3 = w_take(cur, T * 4); s.mask = w_take(cur, (size_t)nW * SW_N * SW_N * 4); }
    s.mg = s.mstats = s.mh = 0;
    if (i > 0) {
      const size_t C4 = 4 * (size_t)d.dim[i - 1];
      s.mg = w_take(cur, T * C4 * 4); s.mstats = w_take(cur, T * 2 * 4); s.mh = w_take(cur, T * C4 * 2);
      if (T * C4 > maxTD) maxTD = T * C4;
      size_t ln = 0; vdk_layernorm_bwd_workspace_bytes((int)T, (int)C4, &ln); if (ln > lnmax) lnmax = ln;
      if (C4 > trows) trows = C4;
      { int k1 = wgrad_splitk((int)C, (int)C4, (int)up(T, 64)), k2 = wgrad_tn_splits((int)C, (int)C4, (int)T); size_t b = (size_t)(k1 > k2 ? k1 : k2) * C * C4 * 4; if (b > sl) sl = b; }
    }
    s.blk.resize(d.depth[i]);
    for (int j = 0; j < d.depth[i]; ++j) {
      BlkW& b = s.blk[j];
      b.x = w_take(cur, T * C * 4); b.xmid = w_take(cur, T * C * 4); b.stats = w_take(cur, T * 4 * 4);
      b.h1 = w_take(cur, T * C * 2); b.qkv = w_take(cur, T * 3 * C * 2); b.lse = w_take(cur, T * H * 4); b.o = w_take(cur, T * C * 2);
      b.h2 = w_take(cur, T * C * 2); b.u = w_take(cur, T * M * 2); b.g = w_take(cur, T * M * 2);
      b.bias = w_take(cur, vdk_wa_bm_bytes(((j & 1) && d.res[i] > SW_WS) ? (int)nW : 0, (int)H));       // bias (+ shift mask) in the attention kernels' fragment order
    }
    s.xout = w_take(cur, T * C * 4);
    if (T * C > maxTD) maxTD = T * C;
    if (T * M > maxTM) maxTM = T * M;
    if (T * 3 * C > maxT3) maxT3 = T * 3 * C;
    if ((long)up(T, 64) > tcols) tcols = (long)up(T, 64);
    if (M > trows) trows = M;
    const int sh[4][2] = {{(int)M, (int)C}, {(int)C, (int)M}, {3 * (int)C, (int)C}, {(int)C, (int)C}};
    for (auto& q : sh) { int k1 = wgrad_splitk(q[0], q[1], (int)up(T, 64)), k2 = wgrad_tn_splits(q[0], q[1], (int)T); size_t b = (size_t)(k1 > k2 ? k1 : k2) * q[0] * q[1] * 4; if (b > sl) sl = b; }
    size_t ln = 0; vdk_layernorm_bwd_workspace_bytes((int)T, (int)C, &ln); if (ln > lnmax) lnmax = ln;
    size_t cs = 0; vdk_colsum_bf16_workspace_bytes((int)T, (int)M, &cs); if (cs > csmax) csmax = cs;
    { size_t c3 = (size_t)2 * ((T + 255) / 256) * M * 4; if (c3 > csmax) csmax = c3; }
    const size_t wa = vdk_wa_bwd_scratch_bytes((int64_t)(T / SW_N), (int)H); if (wa > wamax) wamax = wa;
  }
  const size_t T3 = (size_t)d.T[d.nst - 1], D = d.dim[d.nst - 1];
  w->fmap = w_take(cur, T3 * D * 4); w->fstats = w_take(cur, T3 * 2 * 4);
  w->pooled = w_take(cur, (size_t)d.B * D * 4); w->hf = w_take(cur, (size_t)d.Bp * D * 2);
  // the patch embedding's LayerNorm backward and weight gradient
  { size_t ln = 0; vdk_layernorm_bwd_workspace_bytes((int)d.T[0], d.E, &ln); if (ln > lnmax) lnmax = ln;
    int k1 = wgrad_splitk(d.E, d.Kpe, (int)up(d.T[0], 64)), k2 = wgrad_tn_splits(d.E, d.Kpe, (int)d.T[0]); size_t b = (size_t)(k1 > k2 ? k1 : k2) * d.E * d.Kpe * 4; if (b > sl) sl = b;
    size_t cs = 0; vdk_colsum_bf16_workspace_bytes((int)d.T[0], d.E, &cs); if (cs > csmax) csmax = cs;
    if ((size_t)d.Kpe > trows) trows = d.Kpe; }
  if (d.C > 0) {
    if ((size_t)d.Cp > trows) trows = d.Cp;
    { int k1 = wgrad_splitk(d.Cp, (int)D, d.Bp), k2 = wgrad_tn_splits(d.Cp, (int)D, d.B); size_t b = (size_t)(k1 > k2 ? k1 : k2) * d.Cp * D * 4; if (b > sl) sl = b; }      // (batch % 64 == 0: linear_wgrad takes the TN form with its own split count)
    size_t cs = (size_t)((d.Bp + 63) / 64) * d.Cp * 4; if (cs > csmax) csmax = cs;
  }
  if (D > trows) trows = D;
  w->dxa = w_take(cur, maxTD * 4); w->dxm = w_take(cur, maxTD * 4);
  w->dxab = w_take(cur, maxTD * 2); w->dxmb = w_take(cur, maxTD * 2); w->dsm = w_take(cur, maxTD * 2);
  w->du = w_take(cur, maxTM * 2); w->dqkv = w_take(cur, maxT3 * 2);
  w->dpool = w_take(cur, (size_t)d.B * D * 4); w->dhf = w_take(cur, (size_t)d.Bp * D * 2);
  w->trows = trows; w->tcols = tcols;
  w->tA = w_take(cur, trows * (size_t)tcols * 2); w->tB = w_take(cur, trows * (size_t)tcols * 2);
  { size_t cs2 = (size_t)((tcols + 63) / 64) * trows * 4; if (cs2 > csmax) csmax = cs2; }
  w->slabs_bytes = sl; w->slabs = w_take(cur, sl);
  w->lnws_bytes = (lnmax + 255) & ~(size_t)255; w->lnws = w_take(cur, 2 * w->lnws_bytes);
  w->csws_bytes = (csmax + 255) & ~(size_t)255; w->csws = w_take(cur, 6 * w->csws_bytes);
  w->waws_bytes = wamax; w->waws = w_take(cur, wamax);
  w->total = cur;
}

// ---------------------------------------------------------------------------------------------------------------- small kernels
// The index tables of a batch size, ALL in one launch (round 5: they were 12 launches per forward, one of them a single workgroup walking 169 x 2401 pairs for 250 us).
// A segment per table; a thread finds its segment by a linear scan of <= 13 entries travelling in the kernel arguments.
//   kind 0  rowidx: token (window w = (b, wy, wx), j = (ty, tx)) of the window partition of the map rolled by -shift lives in image-order row
//                   b res^2 + ((7 wy + ty + shift) mod res) res + (7 wx + tx + shift) mod res
//   kind 1  timm's attn_mask of a shifted block: region ids of the rolled frame (rows / columns < res - 7, < res - shift, rest), 0 inside a region pair, -100 across
//   kind 2  uses[r][0 .. 49): the positions q * 49 + k that read table entry r, ascending, -1 padded
__device__ __forceinline__ int swin_rel(int qi, int kj) {      // relative_position_index[q][k] = (yq - yk + 6) * 13 + (xq - xk + 6)
  return (qi / SW_WS - kj / SW_WS + SW_WS - 1) * (2 * SW_WS - 1) + (qi % SW_WS - kj % SW_WS + SW_WS - 1);
}
struct SwTableSeg { long first; void* out; int kind, B, res, shift; };
struct SwTables { SwTableSeg seg[13]; int n; long total; };
__global__ __launch_bounds__(256) void swin_tables_kernel(SwTables t) {
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= t.total) return;
  int k = 0;
  while (k + 1 < t.n && gi >= t.seg[k + 1].first) ++k;
  const SwTableSeg sg = t.seg[k];
  const long i = gi - sg.first;
  const int res = sg.res, shift = sg.shift, nw = res / SW_WS;
  if (sg.kind == 0) {
    const int j = (int)(i % SW_N);
    const long wi = i / SW_N;
    const int wx = (int)(wi % nw), wy = (int)((wi / nw) % nw), b = (int)(wi / ((long)nw * nw));
    const int y = (SW_WS * wy + j / SW_WS + shift) % res, x = (SW_WS * wx + j % SW_WS + shift) % res;
    ((int*)sg.out)[i] = (int)((long)b * res * res + (long)y * res + x);
  } else if (sg.kind == 1) {
    const int kj = (int)(i % SW_N), qi = (int)((i / SW_N) % SW_N);
    const int wi = (int)(i / (SW_N * SW_N)), wx = wi % nw, wy = wi / nw;
    auto reg = [&](int tk) {
      const int y = SW_WS * wy + tk / SW_WS, x = SW_WS * wx + tk % SW_WS;
      const int ry = y < res - SW_WS ? 0 : (y < res - shift ? 1 : 2), rx = x < res - SW_WS ? 0 : (x < res - shift ? 1 : 2);
      return ry * 3 + rx;
    };
    ((float*)sg.out)[i] = reg(qi) == reg(kj) ? 0.0f : -100.0f;
  } else {
    // entry r = (dy + 6) * 13 + (dx + 6) is read by the pairs (q, k) with yq - yk = dy, xq - xk = dx: at most one k per q, so the ascending positions are the valid q in
    // order; thread (r, q) places its own pair at its rank among them, and pads slot q when q >= their number (7 - |dy|) (7 - |dx|).
    const int r = (int)(i / SW_N), q = (int)(i % SW_N);
    const int dy = r / (2 * SW_WS - 1) - (SW_WS - 1), dx = r % (2 * SW_WS - 1) - (SW_WS - 1);
    auto key_of = [&](int qq) { const int yk = qq / SW_WS - dy, xk = qq % SW_WS - dx; return (yk >= 0 && yk < SW_WS && xk >= 0 && xk < SW_WS) ? yk * SW_WS + xk : -1; };
    int* const uses = (int*)sg.out;
    const int kq = key_of(q);
    if (kq >= 0) {
      int n = 0;
      for (int qq = 0; qq < q; ++qq) n += key_of(qq) >= 0;
      uses[r * SW_N + n] = q * SW_N + kq;
    }
    const int cnt = (SW_WS - (dy < 0 ? -dy : dy)) * (SW_WS - (dx < 0 ? -dx : dx));
    if (q >= cnt) uses[r * SW_N + q] = -1;
  }
}
// PatchMerging's gather: out[(b, y2, x2)][(xp * 2 + yp) * C + c] = in[(b, 2 y2 + yp, 2 x2 + xp)][c]   (timm: reshape(B, H/2, 2, W/2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3))
// INVERSE = the scatter of the gradient (same index map, roles swapped)
template <bool INVERSE>
__global__ __launch_bounds__(256) void swin_merge_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int res, int C) {      // res: the INPUT map's side
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
  const int c4n = C / 4;
  const long n4 = (long)B * res * res * c4n;
  if (i4 >= n4) return;
  const int c = (int)(i4 % c4n) * 4;
  const long pix = i4 / c4n;
  const int x = (int)(pix % res), y = (int)((pix / res) % res), b = (int)(pix / ((long)res * res));
  const int r2 = res / 2;
  const long mrow = (long)b * r2 * r2 + (long)(y >> 1) * r2 + (x >> 1);
  const long moff = mrow * 4 * C + (long)((x & 1) * 2 + (y & 1)) * C + c;
  const long ioff = pix * C + c;
  if (INVERSE) *(f32x4*)(out + ioff) = *(const f32x4*)(in + moff);
  else *(f32x4*)(out + moff) = *(const f32x4*)(in + ioff);
}

// ---------------------------------------------------------------------------------------------------------------- GEMM helpers (as in vit_engine.hip)
int gemm(hipStream_t s, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, int cdt, const float* bias, const float* res, int64_t ldr,
         int act, void* aux, int64_t ldaux, int splitk, void* ws, size_t wsb, float* c_colsum = nullptr, const float* row_scale = nullptr, int rows_per_scale = 0) {
  VdkGemmDesc g = {};
  g.row_scale = row_scale; g.rows_per_scale = rows_per_scale;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.c_dtype = cdt == VDK_F32 ? VDK_F32 : DT16; g.bias = bias; g.residual = res; g.ldr = ldr;
  g.act = act; g.aux = aux; g.ldaux = ldaux; g.alpha = 1.0f; g.splitk = splitk; g.c_colsum = c_colsum; g.ab_dtype = DT16;
  return vdk_gemm_bf16_nt(&g, ws, wsb, s);
}
// dW[out, in] = dY^T X (+ db = colsum(dY) when db != nullptr): TN GEMM straight from the row-major tensors when rows % 64 == 0, else through zero-padded transposes
int linear_wgrad(hipStream_t s, const WsPlan& w, char* base, const bf16_t* dY, int64_t lddy, const bf16_t* Xa, int64_t ldx, int rows, int out, int in, float* dW, float* db) {
  if ((rows % 64) == 0 && (out % 8) == 0 && (in % 8) == 0 && out >= 8 && in >= 8) {
    VdkGemmDesc g = {};
    g.A = dY; g.lda = lddy; g.B = Xa; g.ldb = ldx; g.C = dW; g.ldc = in; g.M = out; g.N = in; g.K = rows; g.c_dtype = VDK_F32; g.alpha = 1.0f; g.ab_dtype = DT16;
    const bool half = wgrad_tn_half(out, in);
    g.splitk = wgrad_tn_splits(out, in, rows); g.trans = 1;
    vdk_gemm_tn_prefer_half(half ? 1 : 0);
    const int rc = vdk_gemm_bf16_nt(&g, base + w.slabs, w.slabs_bytes, s);
    vdk_gemm_tn_prefer_half(0);
    RC(rc);
    if (db) RC(vdk_colsum_16(dY, lddy, rows, out, db, base + w.csws + 5 * w.csws_bytes, w.csws_bytes, t_opf, s));
    return VDK_OK;
  }
  const int rows_pad = (int)up(rows, 64);
  bf16_t* tA = (bf16_t*)(base + w.tA); bf16_t* tB = (bf16_t*)(base + w.tB);
  float* csp = db ? (float*)(base + w.csws + 5 * w.csws_bytes) : nullptr;
  RC(vdk_transpose_16(dY, lddy, rows, out, tA, rows_pad, rows_pad, 0, csp, t_opf, s));
  RC(vdk_transpose_16(Xa, ldx, rows, in, tB, rows_pad, rows_pad, 0, nullptr, t_opf, s));
  RC(gemm(s, tA, rows_pad, tB, rows_pad, dW, in, out, in, rows_pad, VDK_F32, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, wgrad_splitk(out, in, rows_pad), base + w.slabs, w.slabs_bytes));
  if (csp) RC(vdk_reduce_rows_f32(csp, out, (rows_pad + 63) / 64, out, db, 1.0f, s));
  return VDK_OK;
}

}  // namespace

extern "C" {

int vdk_swin_param_count(const VdkSwinConfig* cfg, int64_t* n_floats, int32_t* n_tensors, int64_t* n_transposed) {
  SwDims d; RC(sw_dims(cfg, &d));
  t_opf = d.opf;
  PLayout p; sw_layout(d, &p);
  if (n_floats) *n_floats = p.total;
  if (n_tensors) *n_tensors = (int32_t)p.entries.size();
  if (n_transposed) *n_transposed = p.totalT;
  return VDK_OK;
}

int vdk_swin_param_info(const VdkSwinConfig* cfg, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int64_t* shape4, int32_t* ndim) {
  SwDims d; RC(sw_dims(cfg, &d));
  t_opf = d.opf;
  PLayout p; sw_layout(d, &p);
  if (index < 0 || index >= (int)p.entries.size()) return vdk_fail(VDK_EINVAL, "vdk_swin_param_info: index out of range");
  const PEntry& e = p.entries[index];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", e.name);
  if (offset) *offset = e.off;
  if (numel) *numel = e.numel;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = e.shape[i];
  if (ndim) *ndim = e.ndim;
  return VDK_OK;
}

int vdk_swin_workspace_bytes(const VdkSwinConfig* cfg, size_t* bytes) {
  SwDims d; RC(sw_dims(cfg, &d));
  t_opf = d.opf;
  WsPlan w; sw_plan(d, &w);
  if (!bytes) return vdk_fail(VDK_EINVAL, "null");
  *bytes = w.total;
  return VDK_OK;
}

// (Re)build the bf16 operand copies from the fp32 master weights: wb16 (same layout) unless skip_wb16 (vdk_sgd_step already refreshed it), and the [in, out]
// transposes the input-gradient GEMMs read -- one batched launch for all of them
int vdk_swin_refresh_weights(const VdkSwinConfig* cfg, const float* params, void* wb16, void* wt16, int32_t skip_wb16, void* stream) {
  SwDims d; RC(sw_dims(cfg, &d));
  t_opf = d.opf;
  PLayout p; sw_layout(d, &p);
  if (!params || !wb16 || !wt16) return vdk_fail(VDK_EINVAL, "vdk_swin_refresh_weights: null pointer");
  if (!skip_wb16) RC(vdk_cast_f32_16(params, wb16, p.total, t_opf, stream));
  bf16_t* wt = (bf16_t*)wt16;
  std::vector<VdkTcItem> jobs;
  auto add = [&](int64_t off, int R, int Cc, int64_t toff) { jobs.push_back(VdkTcItem{params + off, wt + toff, Cc, R, Cc, R, R}); };
  for (int i = 0; i < d.nst; ++i) {
    const int C = d.dim[i], M = 4 * C;
    if (i > 0) add(p.st[i].ds_w, C, 4 * d.dim[i - 1], p.st[i].ds_t);
    for (const BlkP& b : p.st[i].blk) { add(b.qkv_w, 3 * C, C, b.tq); add(b.proj_w, C, C, b.tp); add(b.fc1_w, M, C, b.t1); add(b.fc2_w, C, M, b.t2); }
  }
  if (d.C > 0) add(p.fc_w, d.Cp, d.dim[d.nst - 1], p.fc_t);
  return vdk_transpose_cast_batch(jobs.data(), (int)jobs.size(), stream, t_opf);
}

// x: f32 [B, Cin, img, img] -> out f32: logits [B, Cp] (columns C..Cp-1 padding) or, in feature mode (num_classes = 0), the normed NHWC map [B * 49 * (img / 224)^2, D]
// (timm's forward_features of this family).  Saves activations in ws.
int vdk_swin_forward(const VdkSwinConfig* cfg, const float* x, const float* params, const void* wb16, void* ws, size_t ws_bytes, float* out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  SwDims d; RC(sw_dims(cfg, &d));
  t_opf = d.opf;
  PLayout p; sw_layout(d, &p);
  WsPlan w; sw_plan(d, &w);
  if (!x || !params || !wb16 || !ws || !out) return vdk_fail(VDK_EINVAL, "vdk_swin_forward: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_swin_forward: workspace too small");
  char* base = (char*)ws;
  const bf16_t* wb = (const bf16_t*)wb16;
  // index tables of this batch size (a few tiny launches; they stay in the workspace for the backward)
  {
    SwTables t; t.n = 0; t.total = 0;
    auto seg = [&](void* out, long count, int kind, int res, int shift) { t.seg[t.n++] = SwTableSeg{t.total, out, kind, d.B, res, shift}; t.total += count; };
    seg(base + w.uses, (long)SW_NREL * SW_N, 2, SW_WS, 0);
    for (int i = 0; i < d.nst; ++i) {
      seg(base + w.st[i].rowidx0, d.T[i], 0, d.res[i], 0);
      if (d.res[i] > SW_WS) {
        seg(base + w.st[i].rowidx3, d.T[i], 0, d.res[i], SW_WS / 2);
        seg(base + w.st[i].mask, (long)(d.res[i] / SW_WS) * (d.res[i] / SW_WS) * SW_N * SW_N, 1, d.res[i], SW_WS / 2);
      }
    }
    hipLaunchKernelGGL(swin_tables_kernel, dim3((unsigned)((t.total + 255) / 256)), dim3(256), 0, s, t);
  }
  // patch embedding: patch operand (an index permutation of the image) x Linear, then its LayerNorm -> the residual stream of stage 0
  bf16_t* patches = (bf16_t*)(base + w.patches);
  RC(vdk_patchify_16(x, d.B, d.Cin, d.img, d.img, 4, patches, d.Kpe, t_opf, s));
  float* petmp = (float*)(base + w.petmp); float* pest = (float*)(base + w.pestats);
  RC(gemm(s, patches, d.Kpe, wb + p.pe_w, d.Kpe, petmp, d.E, (int)d.T[0], d.E, d.Kpe, VDK_F32, params + p.pe_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));
  RC(vdk_layernorm_fwd(petmp, d.E, (int)d.T[0], d.E, params + p.pe_nw, params + p.pe_nb, d.eps, base + w.st[0].blk[0].x, d.E, VDK_F32, pest, pest + d.T[0], s));
  {      // every block's additive attention tile (relative-position bias + shift mask) from the current tables: one launch; the backward reads them again
    std::vector<VdkWaPrepJob> jobs;
    for (int i = 0; i < d.nst; ++i)
      for (int j = 0; j < d.depth[i]; ++j) {
        const bool shifted = (j & 1) && d.res[i] > SW_WS;
        jobs.push_back(VdkWaPrepJob{params + p.st[i].blk[j].table, shifted ? (const float*)(base + w.st[i].mask) : nullptr, (float*)(base + w.st[i].blk[j].bias),
                                    shifted ? (d.res[i] / SW_WS) * (d.res[i] / SW_WS) : 0, d.heads[i]});
      }
    RC(vdk_wa_prep_table_batch(jobs.data(), (int)jobs.size(), s));
  }
  const float* xprev = nullptr;      // output of the previous stage
  int kblk = 0;                      // running block index: rows 2 k / 2 k + 1 of the drop-path factors
  for (int i = 0; i < d.nst; ++i) {
    const StageW& sw_ = w.st[i]; const StageP& sp = p.st[i];
    const int T = (int)d.T[i], C = d.dim[i], M = 4 * C, H = d.heads[i];
    const int nW = (d.res[i] / SW_WS) * (d.res[i] / SW_WS);
    const int tpi = T / d.B;         // tokens per image: rows that share a drop-path factor
    if (i > 0) {      // PatchMerging in front of the stage
      const int C4 = 4 * d.dim[i - 1];
      float* mg = (float*)(base + sw_.mg); float* mst = (float*)(base + sw_.mstats); bf16_t* mh = (bf16_t*)(base + sw_.mh);
      const long n4 = d.T[i - 1] * (d.dim[i - 1] / 4);
      hipLaunchKernelGGL(swin_merge_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, xprev, mg, d.B, d.res[i - 1], d.dim[i - 1]);
      RC(vdk_layernorm_fwd(mg, C4, T, C4, params + sp.ds_nw, params + sp.ds_nb, d.eps, mh, C4, DT16, mst, mst + T, s));
      RC(gemm(s, mh, C4, wb + sp.ds_w, C4, base + sw_.blk[0].x, C, T, C, C4, VDK_F32, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));
    }
    for (int j = 0; j < d.depth[i]; ++j, ++kblk) {
      const BlkW& bw = sw_.blk[j]; const BlkP& b = sp.blk[j];
      const bool shifted = (j & 1) && d.res[i] > SW_WS;
      const float* dp1 = d.dp ? d.dp + (size_t)(2 * kblk) * d.B : nullptr; const float* dp2 = d.dp ? dp1 + d.B : nullptr;      // stochastic depth: x + factor_b * branch(x)
      float* xin = (float*)(base + bw.x); float* xmid = (float*)(base + bw.xmid);
      float* xout = (float*)(base + (j + 1 < d.depth[i] ? sw_.blk[j + 1].x : sw_.xout));
      float* st = (float*)(base + bw.stats);
      bf16_t* h1 = (bf16_t*)(base + bw.h1); bf16_t* qkv = (bf16_t*)(base + bw.qkv); bf16_t* o = (bf16_t*)(base + bw.o); bf16_t* h2 = (bf16_t*)(base + bw.h2);
      bf16_t* u = (bf16_t*)(base + bw.u); bf16_t* g = (bf16_t*)(base + bw.g);
      float* bias = (float*)(base + bw.bias);
      // x = x + proj(W-MSA(norm1(x)))
      RC(vdk_layernorm_fwd(xin, C, T, C, params + b.n1w, params + b.n1b, d.eps, h1, C, DT16, st, st + T, s));
      RC(gemm(s, h1, C, wb + b.qkv_w, C, qkv, 3 * C, T, 3 * C, C, DT16, params + b.qkv_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));
      RC(vdk_wa_fwd_bm(qkv, 3 * C, o, C, (float*)(base + bw.lse), bias, shifted ? nW : 1, (int64_t)(T / SW_N), H, 0.17677669529663687f /* 32^-0.5 */,
                       (const int32_t*)(base + (shifted ? sw_.rowidx3 : sw_.rowidx0)), t_opf, s));
      RC(gemm(s, o, C, wb + b.proj_w, C, xmid, C, T, C, C, VDK_F32, params + b.proj_b, xin, C, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0, nullptr, dp1, tpi));
      // x = x + fc2(gelu(fc1(norm2(x))))
      RC(vdk_layernorm_fwd(xmid, C, T, C, params + b.n2w, params + b.n2b, d.eps, h2, C, DT16, st + 2 * (size_t)T, st + 3 * (size_t)T, s));
      RC(gemm(s, h2, C, wb + b.fc1_w, C, g, M, T, M, C, DT16, params + b.fc1_b, nullptr, 0, sw_saved_grad() ? VDK_ACT_GELU_SAVE_GRAD : VDK_ACT_GELU, u, M, 1, nullptr, 0));
      RC(gemm(s, g, M, wb + b.fc2_w, M, xout, C, T, C, M, VDK_F32, params + b.fc2_b, xmid, C, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0, nullptr, dp2, tpi));
    }
    xprev = (const float*)(base + sw_.xout);
  }
  const int T3 = (int)d.T[d.nst - 1], D = d.dim[d.nst - 1];
  float* fst = (float*)(base + w.fstats);
  if (d.C == 0) {
    RC(vdk_layernorm_fwd(xprev, D, T3, D, params + p.norm_w, params + p.norm_b, d.eps, out, D, VDK_F32, fst, fst + T3, s));
    return vdk_check_launch("vdk_swin_forward");
  }
  float* fmap = (float*)(base + w.fmap); float* pooled = (float*)(base + w.pooled); bf16_t* hf = (bf16_t*)(base + w.hf);
  RC(vdk_layernorm_fwd(xprev, D, T3, D, params + p.norm_w, params + p.norm_b, d.eps, fmap, D, VDK_F32, fst, fst + T3, s));
  RC(vdk_avgpool_rows_f32_fwd(fmap, pooled, d.B, T3 / d.B, D, s));
  if (d.Bp != d.B && hipMemsetAsync(hf + (size_t)d.B * D, 0, (size_t)(d.Bp - d.B) * D * 2, s) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_swin_forward: memset failed");
  RC(vdk_cast_f32_16(pooled, hf, (int64_t)d.B * D, t_opf, s));
  RC(gemm(s, hf, D, wb + p.fc_w, D, out, d.Cp, d.B, d.Cp, D, VDK_F32, params + p.fc_b, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));
  return vdk_check_launch("vdk_swin_forward");
}

// dout: dlogits bf16 [B, Cp] (what vdk_softmax_ce writes, padding columns zero); feature mode: f32 [T3, D] = dL/d(normed map).  grads: flat fp32, param layout, fully
// overwritten.  on_ready(user, offset, numel): called on the host right after the kernels producing grads[offset, offset + numel) have been enqueued (descending, contiguous
// ranges: head + final norm, then block by block from the last to the first with each stage's downsample, then the patch embedding) -- the data-parallel bucket hook.
int vdk_swin_backward(const VdkSwinConfig* cfg, const void* dout, const float* params, const void* wb16, const void* wt16, void* ws, size_t ws_bytes, float* grads,
                      vdk_grad_ready_fn on_ready, void* user, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  SwDims d; RC(sw_dims(cfg, &d));
  t_opf = d.opf;
  PLayout p; sw_layout(d, &p);
  WsPlan w; sw_plan(d, &w);
  if (!dout || !params || !wb16 || !wt16 || !ws || !grads) return vdk_fail(VDK_EINVAL, "vdk_swin_backward: null pointer");
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_swin_backward: workspace too small");
  char* base = (char*)ws;
  const bf16_t* wt = (const bf16_t*)wt16;
  float* dxa = (float*)(base + w.dxa); float* dxm = (float*)(base + w.dxm);
  bf16_t* dxab = (bf16_t*)(base + w.dxab); bf16_t* dxmb = (bf16_t*)(base + w.dxmb); bf16_t* dsm = (bf16_t*)(base + w.dsm);
  bf16_t* du = (bf16_t*)(base + w.du); bf16_t* dqkv = (bf16_t*)(base + w.dqkv);
  char* lnws0 = base + w.lnws; char* lnws1 = lnws0 + w.lnws_bytes;
  auto csws = [&](int slot) { return base + w.csws + (size_t)slot * w.csws_bytes; };
  const int T3 = (int)d.T[d.nst - 1], D = d.dim[d.nst - 1];
  const float* xlast = (const float*)(base + w.st[d.nst - 1].xout);
  float* fst = (float*)(base + w.fstats);
  // Stochastic depth in the backward: a branch's input gradient is factor_b * dL/dx_out, and every GEMM of the branch reads the 16-BIT copy of that gradient -- so the kernel that
  // stores a copy (the LayerNorm backward in front of it, or the stage-boundary cast) multiplies the factor of the branch that will read it into the copy only; the fp32
  // stream (the shortcut's gradient) stays.  dpf(k, which): block k's attention (0) / MLP (1) factors.
  int nblk_total = 0;
  for (int i = 0; i < d.nst; ++i) nblk_total += d.depth[i];
  auto dpf = [&](int k, int which) -> const float* { return d.dp ? d.dp + (size_t)(2 * k + which) * d.B : nullptr; };
  int kblk = nblk_total - 1;
  const int tpi3 = T3 / d.B;
  // ---- head + final norm: dxa / dxab = dL/d(stage 3 output) ----------------------------------------------------------------------------
  if (d.C == 0) {
    RC(vdk_layernorm_bwd_deferred(dout, D, VDK_F32, xlast, D, fst, fst + T3, params + p.norm_w, nullptr, 0, T3, D, dxa, D, dxab, D, grads + p.norm_w, grads + p.norm_b, lnws0,
                                  w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf, nullptr, dpf(kblk, 1), tpi3));
    if (on_ready) on_ready(user, p.norm_w, p.total - p.norm_w);
  } else {
    const bf16_t* dl = (const bf16_t*)dout;
    bf16_t* hf = (bf16_t*)(base + w.hf);
    float* dpool = (float*)(base + w.dpool);
    RC(linear_wgrad(s, w, base, dl, d.Cp, hf, D, d.B, d.Cp, D, grads + p.fc_w, grads + p.fc_b));
    RC(gemm(s, dl, d.Cp, wt + p.fc_t, d.Cp, dpool, D, d.B, D, d.Cp, VDK_F32, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));
    float* dmap = dxm;      // (scratch: d(normed map) f32 [T3, D])
    RC(vdk_avgpool_rows_f32_bwd(dpool, dmap, nullptr, d.B, T3 / d.B, D, s));
    RC(vdk_layernorm_bwd_deferred(dmap, D, VDK_F32, xlast, D, fst, fst + T3, params + p.norm_w, nullptr, 0, T3, D, dxa, D, dxab, D, grads + p.norm_w, grads + p.norm_b, lnws0,
                                  w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf, nullptr, dpf(kblk, 1), tpi3));
    if (on_ready) on_ready(user, p.norm_w, p.total - p.norm_w);
  }
  // ---- stages, last to first -------------------------------------------------------------------------------------------------------------
  for (int i = d.nst - 1; i >= 0; --i) {
    const StageW& sw_ = w.st[i]; const StageP& sp = p.st[i];
    const int T = (int)d.T[i], C = d.dim[i], M = 4 * C, H = d.heads[i];
    const int nW = (d.res[i] / SW_WS) * (d.res[i] / SW_WS);
    const int tpi = T / d.B;
    for (int j = d.depth[i] - 1; j >= 0; --j, --kblk) {
      const BlkW& bw = sw_.blk[j]; const BlkP& b = sp.blk[j];
      const bool shifted = (j & 1) && d.res[i] > SW_WS;
      const float* xin = (const float*)(base + bw.x); const float* xmid = (const float*)(base + bw.xmid);
      const float* st = (const float*)(base + bw.stats);
      const bf16_t* h1 = (const bf16_t*)(base + bw.h1); const bf16_t* qkv = (const bf16_t*)(base + bw.qkv); const bf16_t* o = (const bf16_t*)(base + bw.o);
      const bf16_t* h2 = (const bf16_t*)(base + bw.h2); bf16_t* u = (bf16_t*)(base + bw.u); const bf16_t* g = (const bf16_t*)(base + bw.g);
      VdkReduceJob jobs[8]; int nj = 0;          // this block's small reductions: one batched launch at its end
      // MLP branch: dxa (f32) / dxab (bf16) hold dL/dx_out.  fc1.bias = column sums of du, accumulated by the dGELU epilogue that stores it where that form serves the shape
      const int xrow = vdk_gemm_c_colsum_rows(T, M, C);
      const bool fo = xrow > 0 && (size_t)xrow * M * 4 <= w.csws_bytes;
      float* part1 = (float*)csws(0);
      RC(gemm(s, dxab, C, wt + b.t2, C, du, M, T, M, C, DT16, nullptr, nullptr, 0, sw_saved_grad() ? VDK_ACT_MUL_AUX : VDK_ACT_DGELU, u, M, 1, nullptr, 0, fo ? part1 : nullptr));   // du
      if (fo) jobs[nj++] = VdkReduceJob{part1, (long)M, xrow, (long)M, grads + b.fc1_b, 1.0f};
      // fc2.bias = column sums of dxab: a by-product of the LayerNorm backward that stored it (norm1 of the block after this one, below), except for a stage's last block
      const bool fc2b_done = C <= 1024 && j + 1 < d.depth[i];
      RC(linear_wgrad(s, w, base, dxab, C, g, M, T, C, M, grads + b.fc2_w, fc2b_done ? nullptr : grads + b.fc2_b));
      RC(gemm(s, du, M, wt + b.t1, M, dsm, C, T, C, M, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));   // dh2
      RC(linear_wgrad(s, w, base, du, M, h2, C, T, M, C, grads + b.fc1_w, fo ? nullptr : grads + b.fc1_b));
      // norm2 backward + the shortcut: dxm / dxmb = dL/dx_mid; proj.bias = column sums of the bf16 copy it stores
      const bool ocs = C <= 1024;
      RC(vdk_layernorm_bwd_deferred(dsm, C, DT16, xmid, C, st + 2 * (size_t)T, st + 3 * (size_t)T, params + b.n2w, dxa, C, T, C, dxm, C, dxmb, C, grads + b.n2w, grads + b.n2b,
                                    lnws0, w.lnws_bytes, s, &jobs[nj], ocs ? grads + b.proj_b : nullptr, ocs ? &jobs[nj + 1] : nullptr, nullptr, 0, nullptr, dpf(kblk, 0), tpi));
      nj += ocs ? 2 : 1;
      // attention branch
      RC(gemm(s, dxmb, C, wt + b.tp, C, dsm, C, T, C, C, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));   // do
      RC(linear_wgrad(s, w, base, dxmb, C, o, C, T, C, C, grads + b.proj_w, ocs ? nullptr : grads + b.proj_b));
      RC(vdk_wa_bwd_bm(qkv, 3 * C, o, dsm, C, (const float*)(base + bw.lse), (const float*)(base + bw.bias), shifted ? nW : 1, (int64_t)(T / SW_N), H, 0.17677669529663687f,
                       (const int32_t*)(base + (shifted ? sw_.rowidx3 : sw_.rowidx0)), dqkv, 3 * C, base + w.waws, w.waws_bytes, (const int32_t*)(base + w.uses), SW_NREL, SW_N,
                       grads + b.table, t_opf, s));
      RC(vdk_colsum_bf16_deferred(dqkv, 3 * C, T, 3 * C, grads + b.qkv_b, csws(2), w.csws_bytes, s, &jobs[nj], nullptr, t_opf)); ++nj;
      RC(gemm(s, dqkv, 3 * C, wt + b.tq, 3 * C, dsm, C, T, C, 3 * C, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));   // dh1
      RC(linear_wgrad(s, w, base, dqkv, 3 * C, h1, C, T, 3 * C, C, grads + b.qkv_w, nullptr));
      // norm1 backward + the shortcut: dxa / dxab = dL/dx_in (= dL/dx_out of the block before)
      const bool ocs1 = ocs && j > 0;      // dxab = dL/dx_out of block j - 1: its column sums are that block's fc2.bias gradient
      RC(vdk_layernorm_bwd_deferred(dsm, C, DT16, xin, C, st, st + T, params + b.n1w, dxm, C, T, C, dxa, C, dxab, C, grads + b.n1w, grads + b.n1b, lnws1, w.lnws_bytes, s,
                                    &jobs[nj], ocs1 ? grads + sp.blk[j - 1].fc2_b : nullptr, ocs1 ? &jobs[nj + 1] : nullptr, nullptr, 0, nullptr,
                                    j > 0 ? dpf(kblk - 1, 1) : nullptr, tpi));      // (j == 0: the copy goes to the PatchMerging / patch-embedding backward, not to a branch)
      nj += ocs1 ? 2 : 1;
      RC(vdk_reduce_rows_batch(jobs, nj, s));
      if (on_ready) {
        const int64_t lo = b.n1w;
        const int64_t hi = (j + 1 < d.depth[i]) ? sp.blk[j + 1].n1w : (i < d.nst - 1 ? p.st[i + 1].ds_nw : p.norm_w);
        on_ready(user, lo, hi - lo);
      }
    }
    if (i > 0) {      // PatchMerging backward: dxa (f32 [T, C]) -> dL/d(previous stage's output) in dxa / dxab
      const int C4 = 4 * d.dim[i - 1], Cp_ = d.dim[i - 1];
      const float* mg = (const float*)(base + sw_.mg); const float* mst = (const float*)(base + sw_.mstats); const bf16_t* mh = (const bf16_t*)(base + sw_.mh);
      RC(linear_wgrad(s, w, base, dxab, C, mh, C4, T, C, C4, grads + sp.ds_w, nullptr));
      RC(gemm(s, dxab, C, wt + sp.ds_t, C, dsm, C4, T, C4, C, DT16, nullptr, nullptr, 0, VDK_ACT_NONE, nullptr, 0, 1, nullptr, 0));   // d(norm output) bf16 [T, 4 C_prev]
      RC(vdk_layernorm_bwd_deferred(dsm, C4, DT16, mg, C4, mst, mst + T, params + sp.ds_nw, nullptr, 0, T, C4, dxm, C4, nullptr, 0, grads + sp.ds_nw, grads + sp.ds_nb, lnws0,
                                    w.lnws_bytes, s, nullptr));
      const long n4 = d.T[i - 1] * (Cp_ / 4);
      hipLaunchKernelGGL(swin_merge_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)dxm, dxa, d.B, d.res[i - 1], Cp_);
      RC(vdk_cast_f32_16(dxa, dxab, d.T[i - 1] * Cp_, t_opf, s));
      if (d.dp) RC(vdk_rowscale_16(dxab, d.T[i - 1], Cp_, dpf(kblk, 1), (int)(d.T[i - 1] / d.B), t_opf, s));      // (kblk: already the previous stage's last block) its MLP branch reads this copy
      if (on_ready) on_ready(user, sp.ds_nw, sp.blk[0].n1w - sp.ds_nw);
    }
  }
  // ---- patch embedding: dxa = dL/d(its LayerNorm's output) ----------------------------------------------------------------------------------
  {
    const float* petmp = (const float*)(base + w.petmp); const float* pest = (const float*)(base + w.pestats);
    const int T0 = (int)d.T[0];
    RC(vdk_layernorm_bwd_deferred(dxa, d.E, VDK_F32, petmp, d.E, pest, pest + T0, params + p.pe_nw, nullptr, 0, T0, d.E, nullptr, 0, dxmb, d.E, grads + p.pe_nw, grads + p.pe_nb, lnws0,
                                  w.lnws_bytes, s, nullptr, nullptr, nullptr, nullptr, t_opf));
    RC(linear_wgrad(s, w, base, dxmb, d.E, (const bf16_t*)(base + w.patches), d.Kpe, T0, d.E, d.Kpe, grads + p.pe_w, grads + p.pe_b));
    if (on_ready) on_ready(user, 0, p.st[0].blk[0].n1w);
  }
  return vdk_check_launch("vdk_swin_backward");
}

}  // extern "C"

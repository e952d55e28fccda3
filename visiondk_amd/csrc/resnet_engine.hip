// resnet_engine.hip — native forward/backward of timm's BasicBlock ResNets (`timm-resnet18` is BASELINE.json configs[0]; the reference builds it with
// timm.create_model(name, num_classes=C) at models/classifier/classify_model.py:49-54).  Semantics restated from timm 0.9.16 (un-vendored; see
// oracle/resnet_ref.py, pinned against transformers.ResNetModel): conv7x7/2 + BN + ReLU -> maxpool 3x3/2 -> 4 stages of BasicBlocks
// (conv3x3-BN-ReLU-conv3x3-BN (+ 1x1/2 conv + BN shortcut) + add + ReLU) -> global average pool -> fc.
//
// MI355X design: activations are NHWC (bf16 conv operands, f32 conv outputs kept for BatchNorm's backward); every convolution is an
// implicit GEMM on the MFMA kernel (VdkGemmDesc.conv: the A tile is gathered from the NHWC tensor, no im2col buffer) in forward AND in the input gradient
// (transposed gather of dY) AND in the weight gradient (TN GEMM dY^T . col whose col tiles the four-wave kernel gathers from the NHWC input: VdkConvGeom.rows).
// BatchNorm + shortcut + ReLU is one fused pass (csrc/resnet_ops.hip).  One C call per forward, one per backward; no allocation, no sync.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "vdk_device.h"
#include "vdk_host.h"

extern "C" {
int vdk_gemm_bf16_nt(const VdkGemmDesc*, void*, size_t, void*);
int vdk_reduce_rows_f32(const float*, int64_t, int32_t, int64_t, float*, float, void*);
int vdk_colsum_bf16_workspace_bytes(int32_t, int32_t, size_t*);
int vdk_colsum_bf16(const void*, int64_t, int32_t, int32_t, float*, void*, size_t, void*);
int vdk_cast_f32_bf16(const float*, void*, int64_t, void*);
int vdk_cast_f32_f16(const float*, void*, int64_t, void*);
int vdk_resnet_ops_format(int32_t);
int vdk_transpose_cast_f32_bf16(const float*, int64_t, int32_t, int32_t, void*, int64_t, int32_t, void*);
int vdk_transpose_bf16(const void*, int64_t, int32_t, int32_t, void*, int64_t, int32_t, int32_t, float*, void*);
int vdk_conv_weight_prep(const float*, void*, void*, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_conv_wgrad_unpermute(const float*, float*, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_nchw_to_nhwc_bf16(const float*, void*, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_im2col_bf16(const void*, void*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_bn_rows_workspace_bytes(int64_t, int32_t, size_t*);
int vdk_bn_act_fwd(const float*, int64_t, int32_t, const float*, const float*, float, float, int32_t, float*, float*, const float*, const void*, int32_t, void*, float*,
                   float*, float*, void*, size_t, vdk_stat_sync_fn, void*, void*);
int vdk_bn_act_bwd(const float*, const float*, const void*, int64_t, int32_t, const float*, const float*, const float*, void*, float*, float*, float*, void*, size_t,
                   vdk_stat_sync_fn, void*, void*);
int vdk_maxpool3s2_fwd(const void*, void*, uint8_t*, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_maxpool3s2_bwd(const void*, const uint8_t*, const float*, float*, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_avgpool_fwd(const void*, void*, int32_t, int32_t, int32_t, int32_t, void*);
int vdk_avgpool_bwd(const void*, int64_t, float*, int32_t, int32_t, int32_t, void*);
}

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
int vdk_colsum_16(const void* in, int64_t ld, int32_t T, int32_t N, float* out, void* ws, size_t ws_bytes, int opf, void* stream);      // (C++ linkage: norm_loss.hip, gemm.hip)
int vdk_transpose_16(const void* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, int32_t Rpad, int32_t in_row_group, float* colsum_partial, int opf, void* stream);

namespace {
inline int64_t up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
// operand format of the running engine call (VdkResNetConfig.operand_dtype): every GEMM descriptor and 16-bit helper below reads it
thread_local int t_dt16 = VDK_BF16;
struct FmtGuard {      // sets the format for csrc/resnet_ops.hip's functions and for this file, restores bf16 when the entry point returns
  explicit FmtGuard(int dt) { t_dt16 = dt == VDK_F16 ? VDK_F16 : VDK_BF16; vdk_resnet_ops_format(t_dt16); }
  ~FmtGuard() { t_dt16 = VDK_BF16; vdk_resnet_ops_format(VDK_BF16); }
};

struct Conv { int ci, cip, co, k, s, p, hin, hout; int64_t w; size_t wf, wd; };         // w: param offset; wf / wd: byte offsets in wx
struct Bn { int c; int64_t g, b, rm, rv; };                                                // param offsets (g, b), buffer offsets (rm, rv)
struct Blk { Conv c1, c2, c3, cd; Bn b1, b2, b3, bd; bool ds, bott; int R, Ra; };   // bott: Bottleneck (1x1, 3x3 / stride, 1x1); Ra = rows after c1, R = rows of the block output
struct RnDims {
  int B, Bp, img, Cin, Cinp, C, Cp, H0, Hp, R0, Rp;
  float eps, mom;
  Conv stem; Bn stem_bn;
  std::vector<Blk> blk;
  int64_t fc_w, fc_b, ptotal, btotal;
  size_t fct, xtotal;                          // transposed fc weight in wx
  int wlast;
};
struct PEntry { char name[64]; int64_t off, numel; int64_t shape[4]; int ndim; };
int64_t p_take(int64_t& cur, int64_t n) { int64_t o = cur; cur = up(cur + n, 64); return o; }
size_t w_take(size_t& cur, size_t n) { size_t o = cur; cur = (cur + n + 255) & ~(size_t)255; return o; }
void add_entry(std::vector<PEntry>* v, const char* name, int64_t off, int ndim, int64_t s0, int64_t s1 = 1, int64_t s2 = 1, int64_t s3 = 1) {
  if (!v) return;
  PEntry e; memset(&e, 0, sizeof(e));
  snprintf(e.name, sizeof(e.name), "%s", name);
  e.off = off; e.ndim = ndim; e.shape[0] = s0; e.shape[1] = s1; e.shape[2] = s2; e.shape[3] = s3; e.numel = s0 * s1 * s2 * s3;
  v->push_back(e);
}
void mk_conv(Conv* c, int ci, int co, int k, int s, int p, int hin, int64_t& pcur, size_t& xcur, bool need_wd, const char* name, std::vector<PEntry>* pe) {
  c->ci = ci; c->cip = (int)up(ci, 8); c->co = co; c->k = k; c->s = s; c->p = p; c->hin = hin; c->hout = (hin + 2 * p - k) / s + 1;
  c->w = p_take(pcur, (int64_t)co * ci * k * k);
  c->wf = w_take(xcur, (size_t)co * k * k * c->cip * 2);
  c->wd = need_wd ? w_take(xcur, (size_t)c->cip * k * k * co * 2) : 0;
  add_entry(pe, name, c->w, 4, co, ci, k, k);
}
void mk_bn(Bn* b, int c, int64_t& pcur, int64_t& bcur, const char* prefix, std::vector<PEntry>* pe, std::vector<PEntry>* be) {
  char nm[64];
  b->c = c;
  b->g = p_take(pcur, c); snprintf(nm, 64, "%s.weight", prefix); add_entry(pe, nm, b->g, 1, c);
  b->b = p_take(pcur, c); snprintf(nm, 64, "%s.bias", prefix); add_entry(pe, nm, b->b, 1, c);
  b->rm = p_take(bcur, c); snprintf(nm, 64, "%s.running_mean", prefix); add_entry(be, nm, b->rm, 1, c);
  b->rv = p_take(bcur, c); snprintf(nm, 64, "%s.running_var", prefix); add_entry(be, nm, b->rv, 1, c);
}
int rn_dims(const VdkResNetConfig* c, RnDims* d, std::vector<PEntry>* pe = nullptr, std::vector<PEntry>* be = nullptr) {
  if (!c) return vdk_fail(VDK_EINVAL, "resnet: null config");
  if (c->batch <= 0 || c->img_size <= 0 || (c->img_size % 32) || c->in_chans <= 0 || c->in_chans > 8 || c->num_classes <= 0)
    return vdk_fail(VDK_EINVAL, "resnet: bad config (img_size % 32 == 0, in_chans <= 8, num_classes > 0)");
  d->B = c->batch; d->Bp = (int)up(c->batch, 64); d->img = c->img_size; d->Cin = c->in_chans; d->Cinp = 8; d->C = c->num_classes; d->Cp = (int)up(c->num_classes, 8);
  d->eps = c->bn_eps; d->mom = c->bn_momentum;
  int64_t pcur = 0, bcur = 0; size_t xcur = 0;
  for (int i = 0; i < 4; ++i)
    if (c->widths[i] <= 0 || (c->widths[i] & 7) || c->depths[i] <= 0) return vdk_fail(VDK_EINVAL, "resnet: bad config (widths % 8 == 0, depths > 0)");
  const bool bott = c->mid[0] > 0;
  for (int i = 0; i < 4; ++i)
    if (bott && (c->mid[i] <= 0 || (c->mid[i] & 7))) return vdk_fail(VDK_EINVAL, "resnet: bad config (mid % 8 == 0 for every stage of a bottleneck network)");
  const int stem_w = c->stem_width > 0 ? c->stem_width : (bott ? 64 : c->widths[0]);
  if (stem_w & 7) return vdk_fail(VDK_EINVAL, "resnet: bad config (stem_width % 8 == 0)");
  mk_conv(&d->stem, d->Cin, stem_w, 7, 2, 3, d->img, pcur, xcur, false, "conv1.weight", pe);
  mk_bn(&d->stem_bn, stem_w, pcur, bcur, "bn1", pe, be);
  d->H0 = d->stem.hout; d->Hp = (d->H0 - 1) / 2 + 1;
  d->R0 = d->B * d->H0 * d->H0; d->Rp = d->B * d->Hp * d->Hp;
  d->blk.clear();
  int cin = stem_w, h = d->Hp;
  char nm[64], pf[64];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < c->depths[i]; ++j) {
      Blk b;
      b.bott = bott;
      const int co = c->widths[i], s = (j == 0 && i > 0) ? 2 : 1;
      if (!bott) {
        snprintf(nm, 64, "layer%d.%d.conv1.weight", i + 1, j); mk_conv(&b.c1, cin, co, 3, s, 1, h, pcur, xcur, true, nm, pe);
        snprintf(pf, 64, "layer%d.%d.bn1", i + 1, j); mk_bn(&b.b1, co, pcur, bcur, pf, pe, be);
        snprintf(nm, 64, "layer%d.%d.conv2.weight", i + 1, j); mk_conv(&b.c2, co, co, 3, 1, 1, b.c1.hout, pcur, xcur, true, nm, pe);
        snprintf(pf, 64, "layer%d.%d.bn2", i + 1, j); mk_bn(&b.b2, co, pcur, bcur, pf, pe, be);
      } else {          // timm / torchvision v1.5 Bottleneck: the stride sits on the 3x3
        const int md = c->mid[i];
        snprintf(nm, 64, "layer%d.%d.conv1.weight", i + 1, j); mk_conv(&b.c1, cin, md, 1, 1, 0, h, pcur, xcur, true, nm, pe);
        snprintf(pf, 64, "layer%d.%d.bn1", i + 1, j); mk_bn(&b.b1, md, pcur, bcur, pf, pe, be);
        snprintf(nm, 64, "layer%d.%d.conv2.weight", i + 1, j); mk_conv(&b.c2, md, md, 3, s, 1, h, pcur, xcur, true, nm, pe);
        snprintf(pf, 64, "layer%d.%d.bn2", i + 1, j); mk_bn(&b.b2, md, pcur, bcur, pf, pe, be);
        snprintf(nm, 64, "layer%d.%d.conv3.weight", i + 1, j); mk_conv(&b.c3, md, co, 1, 1, 0, b.c2.hout, pcur, xcur, true, nm, pe);
        snprintf(pf, 64, "layer%d.%d.bn3", i + 1, j); mk_bn(&b.b3, co, pcur, bcur, pf, pe, be);
      }
      const int hout = b.c2.hout;
      b.ds = (s != 1 || cin != co);
      if (b.ds) {
        snprintf(nm, 64, "layer%d.%d.downsample.0.weight", i + 1, j); mk_conv(&b.cd, cin, co, 1, s, 0, h, pcur, xcur, true, nm, pe);
        snprintf(pf, 64, "layer%d.%d.downsample.1", i + 1, j); mk_bn(&b.bd, co, pcur, bcur, pf, pe, be);
      }
      b.Ra = d->B * b.c1.hout * b.c1.hout;
      b.R = d->B * hout * hout;
      d->blk.push_back(b);
      cin = co; h = hout;
    }
  d->wlast = cin;
  d->fc_w = p_take(pcur, (int64_t)d->Cp * cin); add_entry(pe, "fc.weight", d->fc_w, 2, d->C, cin);
  d->fc_b = p_take(pcur, d->Cp); add_entry(pe, "fc.bias", d->fc_b, 1, d->C);
  d->fct = w_take(xcur, (size_t)cin * d->Cp * 2);
  d->ptotal = pcur; d->btotal = bcur; d->xtotal = xcur;
  return VDK_OK;
}

struct BlkW { size_t y1, a1, y2, a2, y3, out, st1, st2, st3, yd, idf, std_; };
struct WsPlan {
  size_t total, img, y0, a0, st0, ap, apk, feat;
  std::vector<BlkW> blk;
  size_t da, db, dyb, dres, tmp, col, dwp, slabs, slabs_bytes, bnws, bnws_bytes, csws, csws_bytes, tA, tB, dfeat, dbn;
};
int wgrad_splitk(int M, int N, int K) {
  if (K < 4096) return 1;
  int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int s = (1024 + tiles - 1) / tiles;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}
int wgrad_splitk_tn(int M, int N, int K) {
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  int s = 256 / tiles;
  if (s < 1) s = 1;
  const int kt = K / 64;
  if (s > kt / 4) s = kt / 4;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}
void rn_plan(const RnDims& d, WsPlan* w) {
  size_t cur = 0;
  w->img = w_take(cur, (size_t)d.B * d.img * d.img * d.Cinp * 2);
  w->y0 = w_take(cur, (size_t)d.R0 * d.stem.co * 4); w->a0 = w_take(cur, (size_t)d.R0 * d.stem.co * 2); w->st0 = w_take(cur, ((size_t)d.stem.co * 2 + 4) * 4);
  w->ap = w_take(cur, (size_t)d.Rp * d.stem.co * 2); w->apk = w_take(cur, (size_t)d.Rp * d.stem.co);   // pooled map and its argmax bytes
  size_t rc = (size_t)d.R0 * d.stem.co, colmax = (size_t)d.R0 * 49 * d.Cinp, dwpmax = (size_t)d.stem.co * 49 * d.Cinp, sl = 0, bn = 0, cs = 0, tr = 0;
  auto wg = [&](int out, int in, int rows) {
    const int k1 = wgrad_splitk(out, in, (int)up(rows, 64)), k2 = wgrad_splitk_tn(out, in, rows), k3 = wgrad_splitk_tn(out, in, (int)up(rows, 128));
    const size_t b = (size_t)(k1 > k2 ? (k1 > k3 ? k1 : k3) : (k2 > k3 ? k2 : k3)) * out * in * 4; if (b > sl) sl = b;
    size_t c2 = 0; vdk_colsum_bf16_workspace_bytes(rows, out, &c2);
    const size_t c1 = (size_t)((up(rows, 64) + 63) / 64) * out * 4;
    if (c2 > cs) cs = c2;
    if (c1 > cs) cs = c1;
    if (rows % 64) { const size_t t = (size_t)(out > in ? out : in) * up(rows, 64) * 2; if (t > tr) tr = t; }
  };
  auto bnw = [&](long R, int C) { size_t q = 0; vdk_bn_rows_workspace_bytes(R, C, &q); if (q > bn) bn = q; };
  wg(d.stem.co, 49 * d.Cinp, d.R0); bnw(d.R0, d.stem.co);
  w->blk.resize(d.blk.size());
  for (size_t i = 0; i < d.blk.size(); ++i) {
    const Blk& b = d.blk[i]; BlkW& bw = w->blk[i];
    const int cout = b.bott ? b.c3.co : b.c2.co;
    const size_t n1 = (size_t)b.Ra * b.c1.co, n2 = (size_t)b.R * b.c2.co, n = (size_t)b.R * cout;
    bw.y1 = w_take(cur, n1 * 4); bw.a1 = w_take(cur, n1 * 2); bw.y2 = w_take(cur, n2 * 4);
    bw.a2 = bw.y3 = bw.st3 = 0;
    if (b.bott) { bw.a2 = w_take(cur, n2 * 2); bw.y3 = w_take(cur, n * 4); bw.st3 = w_take(cur, ((size_t)cout * 2 + 4) * 4); }
    bw.out = w_take(cur, n * 2);
    bw.st1 = w_take(cur, ((size_t)b.c1.co * 2 + 4) * 4); bw.st2 = w_take(cur, ((size_t)b.c2.co * 2 + 4) * 4);
    bw.yd = bw.idf = bw.std_ = 0;
    if (b.ds) { bw.yd = w_take(cur, n * 4); bw.idf = w_take(cur, n * 4); bw.std_ = w_take(cur, ((size_t)cout * 2 + 4) * 4); }
    const size_t nin = (size_t)d.B * b.c1.hin * b.c1.hin * b.c1.cip;
    for (size_t q : {n, n1, n2, nin}) if (q > rc) rc = q;
    auto conv_sizes = [&](const Conv& c, int rows) {
      const size_t cl = (size_t)rows * c.k * c.k * c.cip, dw = (size_t)c.co * c.k * c.k * c.cip;
      if (cl > colmax) colmax = cl;
      if (dw > dwpmax) dwpmax = dw;
      wg(c.co, c.k * c.k * c.cip, rows);
    };
    conv_sizes(b.c1, b.Ra); conv_sizes(b.c2, b.R); bnw(b.Ra, b.c1.co); bnw(b.R, b.c2.co);
    if (b.bott) { conv_sizes(b.c3, b.R); bnw(b.R, b.c3.co); }
    if (b.ds) { conv_sizes(b.cd, b.R); bnw(b.R, b.cd.co); }
  }
  wg(d.Cp, d.wlast, d.B);
  w->feat = w_take(cur, (size_t)d.Bp * d.wlast * 2);
  w->da = w_take(cur, rc * 4); w->db = w_take(cur, rc * 4); w->dyb = w_take(cur, rc * 2); w->dres = w_take(cur, rc * 4); w->tmp = w_take(cur, rc * 4);
  w->col = w_take(cur, colmax * 2); w->dwp = w_take(cur, dwpmax * 4);
  w->slabs_bytes = sl; w->slabs = w_take(cur, sl + 256);
  w->bnws_bytes = bn; w->bnws = w_take(cur, bn + 256);
  w->csws_bytes = cs; w->csws = w_take(cur, cs + 256);
  w->tA = w_take(cur, tr + 256); w->tB = w_take(cur, tr + 256);
  w->dfeat = w_take(cur, (size_t)d.Bp * d.wlast * 2);
  w->dbn = w_take(cur, 4096 * 4);
  w->total = cur;
}

// implicit-GEMM convolution: forward (rows = output pixels) or input gradient (transposed: rows = input pixels, A = dY)
int conv_gemm(hipStream_t s, const Conv& c, int B, bool transposed, const void* A, const void* W, void* Cout, int cdt, const float* res) {
  VdkConvGeom g;
  VdkGemmDesc d = {};
  if (!transposed) {
    g = {c.cip, c.hin, c.hin, c.hout, c.hout, c.k, c.k, c.s, c.p, 0};
    d.M = B * c.hout * c.hout; d.N = c.co; d.K = c.k * c.k * c.cip;
  } else {
    g = {c.co, c.hout, c.hout, c.hin, c.hin, c.k, c.k, c.s, c.p, 1};
    d.M = B * c.hin * c.hin; d.N = c.cip; d.K = c.k * c.k * c.co;
  }
  d.A = A; d.B = W; d.ldb = d.K; d.C = Cout; d.ldc = d.N; d.c_dtype = cdt; d.residual = res; d.ldr = d.N; d.alpha = 1.0f; d.splitk = 1; d.conv = &g; d.ab_dtype = t_dt16;
  return vdk_gemm_bf16_nt(&d, nullptr, 0, s);
}
// dW[out, in] = dY^T X (dY bf16 [rows, out], X bf16 [rows, in]); db optional
int linear_wgrad(hipStream_t s, const WsPlan& w, char* base, const bf16_t* dY, const bf16_t* X, int rows, int out, int in, float* dW, float* db) {
  if ((rows % 64) == 0) {
    VdkGemmDesc g = {};
    g.A = dY; g.lda = out; g.B = X; g.ldb = in; g.C = dW; g.ldc = in; g.M = out; g.N = in; g.K = rows; g.c_dtype = VDK_F32;
    g.alpha = 1.0f; g.splitk = wgrad_splitk_tn(out, in, rows); g.trans = 1; g.ab_dtype = t_dt16;
    RC(vdk_gemm_bf16_nt(&g, base + w.slabs, w.slabs_bytes, s));
    if (db) RC(vdk_colsum_16(dY, out, rows, out, db, base + w.csws, w.csws_bytes, t_dt16 == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16, s));
    return VDK_OK;
  }
  const int rp = (int)up(rows, 64);
  bf16_t* tA = (bf16_t*)(base + w.tA); bf16_t* tB = (bf16_t*)(base + w.tB);
  float* csp = db ? (float*)(base + w.csws) : nullptr;
  RC(vdk_transpose_16(dY, out, rows, out, tA, rp, rp, 0, csp, t_dt16 == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16, s));
  RC(vdk_transpose_16(X, in, rows, in, tB, rp, rp, 0, nullptr, t_dt16 == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16, s));
  VdkGemmDesc g = {};
  g.A = tA; g.lda = rp; g.B = tB; g.ldb = rp; g.C = dW; g.ldc = in; g.M = out; g.N = in; g.K = rp; g.c_dtype = VDK_F32; g.alpha = 1.0f; g.ab_dtype = t_dt16;
  g.splitk = wgrad_splitk(out, in, rp);
  RC(vdk_gemm_bf16_nt(&g, base + w.slabs, w.slabs_bytes, s));
  if (db) RC(vdk_reduce_rows_f32(csp, out, (rp + 63) / 64, out, db, 1.0f, s));
  return VDK_OK;
}
// weight gradient of one convolution, dW'[Co, KH*KW*Cip] = dY^T . im2col(input).  The im2col operand is never written: the four-wave TN kernel gathers its B tiles from the
// NHWC input (VdkConvGeom.rows); VDK_RN_WGRAD_IM2COL=1 (A/B switch) or a geometry that kernel refuses takes the explicit im2col + TN GEMM instead.  Then back to [Co][Ci][k][k].
bool rn_wgrad_implicit() {
  static const bool on = [] { const char* e = getenv("VDK_RN_WGRAD_IM2COL"); return !(e && e[0] == '1'); }();
  return on;
}
int conv_wgrad(hipStream_t s, const WsPlan& w, char* base, const Conv& c, int B, const bf16_t* dY, const void* in_nhwc, float* dW) {
  float* dwp = (float*)(base + w.dwp);
  const int rows = B * c.hout * c.hout, N = c.k * c.k * c.cip;
  if (rn_wgrad_implicit() && rows < (1 << 24) && !(c.co & 7)) {
    VdkConvGeom g = {c.cip, c.hin, c.hin, c.hout, c.hout, c.k, c.k, c.s, c.p, 0, rows};
    VdkGemmDesc d = {};
    d.A = dY; d.lda = c.co; d.B = in_nhwc; d.ldb = N; d.C = dwp; d.ldc = N; d.M = c.co; d.N = N; d.K = (int)up(rows, 128); d.c_dtype = VDK_F32; d.alpha = 1.0f;
    d.splitk = wgrad_splitk_tn(c.co, N, d.K); d.trans = 1; d.conv = &g; d.ab_dtype = t_dt16;
    const int rc = vdk_gemm_bf16_nt(&d, base + w.slabs, w.slabs_bytes, s);
    if (rc == VDK_OK) return vdk_conv_wgrad_unpermute(dwp, dW, c.co, c.ci, c.cip, c.k, c.k, s);
    if (rc != VDK_EUNSUPPORTED) return rc;
  }
  bf16_t* col = (bf16_t*)(base + w.col);
  RC(vdk_im2col_bf16(in_nhwc, col, B, c.hin, c.hin, c.cip, c.hout, c.hout, c.k, c.k, c.s, c.p, s));
  RC(linear_wgrad(s, w, base, dY, col, rows, c.co, N, dwp, nullptr));
  return vdk_conv_wgrad_unpermute(dwp, dW, c.co, c.ci, c.cip, c.k, c.k, s);
}

}  // namespace

extern "C" {

int vdk_resnet_param_count(const VdkResNetConfig* cfg, int64_t* n_floats, int32_t* n_tensors, int64_t* n_buffer_floats, int32_t* n_buffers, size_t* wx_bytes) {
  RnDims d; std::vector<PEntry> pe, be; RC(rn_dims(cfg, &d, &pe, &be));
  if (n_floats) *n_floats = d.ptotal;
  if (n_tensors) *n_tensors = (int32_t)pe.size();
  if (n_buffer_floats) *n_buffer_floats = d.btotal;
  if (n_buffers) *n_buffers = (int32_t)be.size();
  if (wx_bytes) *wx_bytes = d.xtotal;
  return VDK_OK;
}
// which = 0: trainable parameters (flat `params` / `grads`), 1: BatchNorm running statistics (flat `buffers`); timm names
int vdk_resnet_param_info(const VdkResNetConfig* cfg, int32_t which, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int64_t* shape4,
                          int32_t* ndim) {
  RnDims d; std::vector<PEntry> pe, be; RC(rn_dims(cfg, &d, &pe, &be));
  const std::vector<PEntry>& v = which ? be : pe;
  if (index < 0 || index >= (int32_t)v.size()) return vdk_fail(VDK_EINVAL, "vdk_resnet_param_info: index out of range");
  const PEntry& e = v[index];
  if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", e.name);
  if (offset) *offset = e.off;
  if (numel) *numel = e.numel;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = e.shape[i];
  if (ndim) *ndim = e.ndim;
  return VDK_OK;
}
int vdk_resnet_workspace_bytes(const VdkResNetConfig* cfg, size_t* bytes) {
  RnDims d; RC(rn_dims(cfg, &d));
  if (!bytes) return vdk_fail(VDK_EINVAL, "null");
  WsPlan w; rn_plan(d, &w);
  *bytes = w.total;
  return VDK_OK;
}
// wb16 = bf16 copy of the flat parameters (fc operand) unless skip_wb16; wx = implicit-GEMM weight layouts of every convolution + fc^T
int vdk_resnet_refresh_weights(const VdkResNetConfig* cfg, const float* params, void* wb16, void* wx, int32_t skip_wb16, void* stream) {
  RnDims d; RC(rn_dims(cfg, &d));
  if (!params || !wb16 || !wx) return vdk_fail(VDK_EINVAL, "vdk_resnet_refresh_weights: null pointer");
  FmtGuard fmt(cfg->operand_dtype);
  if (!skip_wb16) RC(t_dt16 == VDK_F16 ? vdk_cast_f32_f16(params, wb16, d.ptotal, stream) : vdk_cast_f32_bf16(params, wb16, d.ptotal, stream));
  char* xb = (char*)wx;
  auto prep = [&](const Conv& c) { return vdk_conv_weight_prep(params + c.w, xb + c.wf, c.wd ? xb + c.wd : nullptr, c.co, c.ci, c.cip, c.k, c.k, stream); };
  RC(prep(d.stem));
  for (const Blk& b : d.blk) { RC(prep(b.c1)); RC(prep(b.c2)); if (b.bott) RC(prep(b.c3)); if (b.ds) RC(prep(b.cd)); }
  if (t_dt16 == VDK_F16) {      // fc^T in IEEE half: cast (the refreshed wb16 holds fc.weight already; with skip_wb16 the caller's optimizer pass wrote it) + 16-bit transpose
    return vdk_transpose_16((const bf16_t*)wb16 + d.fc_w, d.wlast, d.Cp, d.wlast, xb + d.fct, d.Cp, d.Cp, 0, nullptr, VDK_OPF_F16, stream);
  }
  return vdk_transpose_cast_f32_bf16(params + d.fc_w, d.wlast, d.Cp, d.wlast, xb + d.fct, d.Cp, d.Cp, stream);
}

// x f32 [B, Cin, img, img] -> logits f32 [B, Cp]; training != 0: batch statistics (running statistics in `buffers` updated), activations kept for backward
int vdk_resnet_forward(const VdkResNetConfig* cfg, const float* x, const float* params, float* buffers, const void* wb16, const void* wx, int32_t training, void* ws,
                       size_t ws_bytes, float* logits, vdk_stat_sync_fn bn_sync, void* bn_user, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  RnDims d; RC(rn_dims(cfg, &d));
  WsPlan w; rn_plan(d, &w);
  if (!x || !params || !buffers || !wb16 || !wx || !ws || !logits) return vdk_fail(VDK_EINVAL, "vdk_resnet_forward: null pointer");
  FmtGuard fmt(cfg->operand_dtype);
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_resnet_forward: workspace too small");
  char* base = (char*)ws; const char* xb = (const char*)wx;
  void* bnws = base + w.bnws;
  auto bn = [&](const Bn& b, const float* xin, long R, size_t st, const float* rf, const void* rb, int relu, void* ob, float* of) {
    float* sm = (float*)(base + st);
    return vdk_bn_act_fwd(xin, R, b.c, params + b.g, params + b.b, d.eps, d.mom, training, buffers + b.rm, buffers + b.rv, rf, rb, relu, ob, of, sm, sm + b.c, bnws,
                          w.bnws_bytes, training ? bn_sync : nullptr, bn_user, s);
  };
  RC(vdk_nchw_to_nhwc_bf16(x, base + w.img, d.B, d.Cin, d.img, d.img, d.Cinp, s));
  RC(conv_gemm(s, d.stem, d.B, false, base + w.img, xb + d.stem.wf, base + w.y0, VDK_F32, nullptr));
  RC(bn(d.stem_bn, (const float*)(base + w.y0), d.R0, w.st0, nullptr, nullptr, 1, base + w.a0, nullptr));
  RC(vdk_maxpool3s2_fwd(base + w.a0, base + w.ap, (uint8_t*)(base + w.apk), d.B, d.H0, d.H0, d.stem.co, s));
  const void* ain = base + w.ap;
  for (size_t i = 0; i < d.blk.size(); ++i) {
    const Blk& b = d.blk[i]; const BlkW& bw = w.blk[i];
    RC(conv_gemm(s, b.c1, d.B, false, ain, xb + b.c1.wf, base + bw.y1, VDK_F32, nullptr));
    RC(bn(b.b1, (const float*)(base + bw.y1), b.Ra, bw.st1, nullptr, nullptr, 1, base + bw.a1, nullptr));
    RC(conv_gemm(s, b.c2, d.B, false, base + bw.a1, xb + b.c2.wf, base + bw.y2, VDK_F32, nullptr));
    const Bn* blast = &b.b2; const float* ylast = (const float*)(base + bw.y2); size_t stlast = bw.st2;
    if (b.bott) {
      RC(bn(b.b2, (const float*)(base + bw.y2), b.R, bw.st2, nullptr, nullptr, 1, base + bw.a2, nullptr));
      RC(conv_gemm(s, b.c3, d.B, false, base + bw.a2, xb + b.c3.wf, base + bw.y3, VDK_F32, nullptr));
      blast = &b.b3; ylast = (const float*)(base + bw.y3); stlast = bw.st3;
    }
    if (b.ds) {
      RC(conv_gemm(s, b.cd, d.B, false, ain, xb + b.cd.wf, base + bw.yd, VDK_F32, nullptr));
      RC(bn(b.bd, (const float*)(base + bw.yd), b.R, bw.std_, nullptr, nullptr, 0, nullptr, (float*)(base + bw.idf)));
      RC(bn(*blast, ylast, b.R, stlast, (const float*)(base + bw.idf), nullptr, 1, base + bw.out, nullptr));
    } else {
      RC(bn(*blast, ylast, b.R, stlast, nullptr, ain, 1, base + bw.out, nullptr));
    }
    ain = base + bw.out;
  }
  const Blk& last = d.blk.back();
  RC(vdk_avgpool_fwd(ain, base + w.feat, d.B, d.Bp, last.c2.hout * last.c2.hout, d.wlast, s));
  VdkGemmDesc g = {};
  g.A = base + w.feat; g.lda = d.wlast; g.B = (const bf16_t*)wb16 + d.fc_w; g.ldb = d.wlast; g.C = logits; g.ldc = d.Cp; g.M = d.B; g.N = d.Cp; g.K = d.wlast;
  g.c_dtype = VDK_F32; g.bias = params + d.fc_b; g.alpha = 1.0f; g.splitk = 1; g.ab_dtype = t_dt16;
  RC(vdk_gemm_bf16_nt(&g, nullptr, 0, s));
  return vdk_check_launch("vdk_resnet_forward");
}

// dlogits bf16 [B, Cp] (padding columns zero) -> grads (flat fp32, param layout, fully overwritten).  Needs the workspace of a training-mode forward.
int vdk_resnet_backward(const VdkResNetConfig* cfg, const void* dlogits, const float* params, const void* wb16, const void* wx, void* ws, size_t ws_bytes, float* grads,
                        vdk_grad_ready_fn on_ready, void* user, vdk_stat_sync_fn bn_sync, void* bn_user, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  RnDims d; RC(rn_dims(cfg, &d));
  WsPlan w; rn_plan(d, &w);
  if (!dlogits || !params || !wb16 || !wx || !ws || !grads) return vdk_fail(VDK_EINVAL, "vdk_resnet_backward: null pointer");
  FmtGuard fmt(cfg->operand_dtype);
  if (ws_bytes < w.total) return vdk_fail(VDK_EWORKSPACE, "vdk_resnet_backward: workspace too small");
  char* base = (char*)ws; const char* xb = (const char*)wx;
  float* da = (float*)(base + w.da); float* db = (float*)(base + w.db); bf16_t* dyb = (bf16_t*)(base + w.dyb);
  float* dres = (float*)(base + w.dres); float* tmp = (float*)(base + w.tmp);
  void* bnws = base + w.bnws;
  auto bnb = [&](const Bn& b, const float* xin, const float* dout, const void* mask, long R, size_t st, float* dr) {
    const float* sm = (const float*)(base + st);
    return vdk_bn_act_bwd(xin, dout, mask, R, b.c, params + b.g, sm, sm + b.c, dyb, dr, grads + b.g, grads + b.b, bnws, w.bnws_bytes, bn_sync, bn_user, s);
  };
  // fc: weight / bias gradient, feature gradient, average-pool backward
  const Blk& last = d.blk.back();
  const int hw = last.c2.hout * last.c2.hout;
  RC(linear_wgrad(s, w, base, (const bf16_t*)dlogits, (const bf16_t*)(base + w.feat), d.B, d.Cp, d.wlast, grads + d.fc_w, grads + d.fc_b));
  {
    VdkGemmDesc g = {};
    g.A = dlogits; g.lda = d.Cp; g.B = xb + d.fct; g.ldb = d.Cp; g.C = base + w.dfeat; g.ldc = d.wlast; g.M = d.B; g.N = d.wlast; g.K = d.Cp; g.c_dtype = t_dt16;
    g.alpha = 1.0f; g.splitk = 1; g.ab_dtype = t_dt16;
    RC(vdk_gemm_bf16_nt(&g, nullptr, 0, s));
  }
  RC(vdk_avgpool_bwd(base + w.dfeat, d.wlast, da, d.B, hw, d.wlast, s));
  if (on_ready) on_ready(user, d.fc_w, d.ptotal - d.fc_w);
  // blocks, last to first: da = dL/d(block output) f32 [R, co]
  for (int i = (int)d.blk.size() - 1; i >= 0; --i) {
    const Blk& b = d.blk[i]; const BlkW& bw = w.blk[i];
    const void* ain = i > 0 ? base + w.blk[i - 1].out : base + w.ap;
    if (b.bott) {
      RC(bnb(b.b3, (const float*)(base + bw.y3), da, base + bw.out, b.R, bw.st3, dres));        // dyb = dL/dy3, dres = masked dout (shortcut gradient)
      RC(conv_wgrad(s, w, base, b.c3, d.B, dyb, base + bw.a2, grads + b.c3.w));
      RC(conv_gemm(s, b.c3, d.B, true, dyb, xb + b.c3.wd, db, VDK_F32, nullptr));               // db = dL/da2
      RC(bnb(b.b2, (const float*)(base + bw.y2), db, base + bw.a2, b.R, bw.st2, nullptr));      // dyb = dL/dy2
      RC(conv_wgrad(s, w, base, b.c2, d.B, dyb, base + bw.a1, grads + b.c2.w));
      RC(conv_gemm(s, b.c2, d.B, true, dyb, xb + b.c2.wd, tmp, VDK_F32, nullptr));              // tmp = dL/da1 (rows at the block's input resolution)
      RC(bnb(b.b1, (const float*)(base + bw.y1), tmp, base + bw.a1, b.Ra, bw.st1, nullptr));    // dyb = dL/dy1
    } else {
      RC(bnb(b.b2, (const float*)(base + bw.y2), da, base + bw.out, b.R, bw.st2, dres));        // dyb = dL/dy2, dres = masked dout (shortcut gradient)
      RC(conv_wgrad(s, w, base, b.c2, d.B, dyb, base + bw.a1, grads + b.c2.w));
      RC(conv_gemm(s, b.c2, d.B, true, dyb, xb + b.c2.wd, db, VDK_F32, nullptr));               // db = dL/da1
      RC(bnb(b.b1, (const float*)(base + bw.y1), db, base + bw.a1, b.R, bw.st1, nullptr));      // dyb = dL/dy1
    }
    RC(conv_wgrad(s, w, base, b.c1, d.B, dyb, ain, grads + b.c1.w));
    if (b.ds) {
      RC(conv_gemm(s, b.c1, d.B, true, dyb, xb + b.c1.wd, tmp, VDK_F32, nullptr));              // main-branch part of dL/d(block input)
      RC(bnb(b.bd, (const float*)(base + bw.yd), dres, nullptr, b.R, bw.std_, nullptr));        // dyb = dL/dyd (no ReLU on the shortcut's BatchNorm)
      RC(conv_wgrad(s, w, base, b.cd, d.B, dyb, ain, grads + b.cd.w));
      RC(conv_gemm(s, b.cd, d.B, true, dyb, xb + b.cd.wd, da, VDK_F32, tmp));                   // + shortcut part
    } else {
      RC(conv_gemm(s, b.c1, d.B, true, dyb, xb + b.c1.wd, da, VDK_F32, dres));                  // identity shortcut: + masked dout
    }
    if (on_ready) {
      const int64_t end = (i + 1 < (int)d.blk.size()) ? d.blk[i + 1].c1.w : d.fc_w;
      on_ready(user, b.c1.w, end - b.c1.w);
    }
  }
  // stem: max-pool backward, BatchNorm + ReLU backward, conv weight gradient (no input gradient: the image is a leaf)
  RC(vdk_maxpool3s2_bwd(base + w.a0, (const uint8_t*)(base + w.apk), da, db, d.B, d.H0, d.H0, d.stem.co, s));
  RC(bnb(d.stem_bn, (const float*)(base + w.y0), db, base + w.a0, d.R0, w.st0, nullptr));
  RC(conv_wgrad(s, w, base, d.stem, d.B, dyb, base + w.img, grads + d.stem.w));
  if (on_ready) on_ready(user, 0, d.blk[0].c1.w);
  return vdk_check_launch("vdk_resnet_backward");
}

}  // extern "C"

// preprocess.hip — the step in front of the hot path: the reference's deterministic validation-time input pipeline
//     resize_and_padding(size, training=False) -> to_tensor -> normalize(mean, std)
// (/root/reference configs/classification/pet.yaml:94-101, configs/faceX/cbir.yaml:92-99; dataset/transforms.py:325-362 and :466-477) for a batch of
// decoded RGB images of DIFFERENT sizes, in one pass over the pixels: uint8 HWC in (ragged, back to back), float32 NCHW out.
//
// Bit-exact with what the reference executes on the CPU: Pillow's `Image.resize(.., BILINEAR)` (ImagingResample: separable, horizontal pass first,
// 8-bit fixed point with 22-bit coefficients, the horizontal result rounded to uint8 before the vertical pass), `ImageOps.expand` with zeros, and
// torchvision's float32 `(u8 / 255 - mean) / std`.  The arithmetic is restated in oracle/preprocess_ref.py (pinned against Pillow itself).
//
//   pp_coeff_kernel   per image and axis: output geometry (Python's `int(w * (size / max_side))` in double) and Pillow's `precompute_coeffs` +
//                     `normalize_coeffs_8bpc` (double arithmetic, no contraction) -> int32 coefficient rows + (first, count) bounds in the workspace.
//   pp_resample_kernel<CPT, TB>   workgroup = (image, band of TB output rows).  The input rows the band's vertical taps touch are staged through LDS in
//                     contiguous chunks (rows of an HWC image are back to back: one aligned dword copy per chunk); a thread owns output column(s) x:
//                     it forms the horizontally resampled uint8 of (row, x, c) and immediately accumulates it into the TB vertical accumulators of its
//                     column, so the intermediate image of the two-pass algorithm never exists.  Each input byte is read from HBM (TB + 2)/TB times.
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"

#define PP_PRECISION_BITS 22
#define PP_MAX_SIDE 8192
#define PP_LDS_BYTES 32768
#define PP_THREADS 256

struct PPGeom {
  int w, h, nw, nh, pl, pt, status, _pad;
};

__host__ __device__ static inline int pp_ksize_bound(int max_side, int S) {
  // a side's scale is < 2 * max_side / S (the short side's output length is truncated, never below 1), support = max(scale, 1)
  const int c = (2 * max_side + S - 1) / S;
  return (c < 1 ? 1 : c) * 2 + 1;
}

// workspace: [B] PPGeom | per image: kh int32 [S][KS], bh int32 [S][2], kv int32 [S][KS], bv int32 [S][2]
__host__ __device__ static inline size_t pp_image_stride(int S, int KS) { return (size_t)2 * S * (KS + 2); }

#pragma clang fp contract(off)
__global__ __launch_bounds__(PP_THREADS) void pp_coeff_kernel(const int* __restrict__ wh, int S, int KS, PPGeom* __restrict__ geom, int* __restrict__ tabs,
                                                              int* __restrict__ status) {
  const int b = blockIdx.x, axis = blockIdx.y;
  const int w = wh[2 * b], h = wh[2 * b + 1];
  const int max_side = w > h ? w : h;
  int nw = 0, nh = 0, st = 0;
  if (w <= 0 || h <= 0 || max_side > PP_MAX_SIDE) {
    st = 2;
  } else {
    const double scale_factor = (double)S / (double)max_side;   // dataset/transforms.py:345-349
    nw = (int)((double)w * scale_factor);
    nh = (int)((double)h * scale_factor);
    if (nw <= 0 || nh <= 0) st = 1;                              // Image.resize raises ValueError("height and width must be > 0")
  }
  if (axis == 0 && threadIdx.x == 0) {
    PPGeom g;
    g.w = w; g.h = h; g.nw = nw; g.nh = nh; g.pl = (S - nw) / 2; g.pt = (S - nh) / 2; g.status = st; g._pad = 0;
    geom[b] = g;
    if (status) status[b] = st;
  }
  if (st) return;
  const int in_size = axis == 0 ? w : h, out_size = axis == 0 ? nw : nh;
  int* kk = tabs + (size_t)b * pp_image_stride(S, KS) + (size_t)axis * S * (KS + 2);
  int* bounds = kk + (size_t)S * KS;
  // Resample.c precompute_coeffs (bilinear: support 1.0, triangle) + normalize_coeffs_8bpc
  const double scale = (double)((float)in_size - 0.0f) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  for (int xx = threadIdx.x; xx < out_size; xx += PP_THREADS) {
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      v = v < 1.0 ? 1.0 - v : 0.0;
      ww += v;
    }
    int* k = kk + (size_t)xx * KS;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      v = v < 1.0 ? 1.0 - v : 0.0;
      if (ww != 0.0) v /= ww;
      k[x] = v < 0.0 ? (int)(-0.5 + v * (double)(1 << PP_PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << PP_PRECISION_BITS));
    }
    for (int x = xmax; x < KS; ++x) k[x] = 0;
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
}

__device__ __forceinline__ int pp_clip8(int acc) {
  const int v = acc >> PP_PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// CPT output columns per thread (S <= CPT * 256), TB output rows per workgroup
template <int CPT, int TB>
__global__ __launch_bounds__(PP_THREADS) void pp_resample_kernel(const unsigned char* __restrict__ pixels, const long* __restrict__ offsets, int S, int KS,
                                                                 const PPGeom* __restrict__ geom, const int* __restrict__ tabs, float m0, float m1, float m2, float s0,
                                                                 float s1, float s2, float* __restrict__ out) {
  __shared__ unsigned int stage[PP_LDS_BYTES / 4 + 2];
  const int b = blockIdx.y, y0 = blockIdx.x * TB, tid = threadIdx.x;
  const PPGeom g = geom[b];
  const int* kh = tabs + (size_t)b * pp_image_stride(S, KS);
  const int* bh = kh + (size_t)S * KS;
  const int* kv = bh + 2 * S;
  const int* bv = kv + (size_t)S * KS;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};

  int acc[CPT][TB][3];
#pragma unroll
  for (int c = 0; c < CPT; ++c)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) acc[c][j][ch] = 1 << (PP_PRECISION_BITS - 1);

  // image rows of this band and the input rows their vertical taps cover
  int ya = y0 - g.pt, yb = y0 + TB - g.pt;
  if (ya < 0) ya = 0;
  if (yb > g.nh) yb = g.nh;
  const bool any_rows = g.status == 0 && ya < yb;
  int xmin[CPT], xcnt[CPT], xx[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    xx[c] = tid + c * PP_THREADS - g.pl;
    const bool on = any_rows && xx[c] >= 0 && xx[c] < g.nw;
    xmin[c] = on ? bh[2 * xx[c]] : 0;
    xcnt[c] = on ? bh[2 * xx[c] + 1] : 0;
  }
  if (any_rows) {
    const int rbeg = bv[2 * ya], rend = bv[2 * (yb - 1)] + bv[2 * (yb - 1) + 1];
    const int w3 = g.w * 3;
    int rb = PP_LDS_BYTES / w3;                      // rows per staged chunk
    if (rb > 16) rb = 16;
    const unsigned char* img = pixels + offsets[b];
    for (int r0 = rbeg; r0 < rend; r0 += rb) {
      const int nr = rend - r0 < rb ? rend - r0 : rb;
      const unsigned char* src = img + (size_t)r0 * w3;
      const size_t mis = (size_t)src & 3;
      const unsigned int* src4 = (const unsigned int*)(src - mis);
      const int ndw = (int)((mis + (size_t)nr * w3 + 3) / 4);
      __syncthreads();
      for (int i = tid; i < ndw; i += PP_THREADS) stage[i] = src4[i];
      __syncthreads();
      const unsigned char* lrow = (const unsigned char*)stage + mis;
      for (int rr = 0; rr < nr; ++rr, lrow += w3) {
        const int r = r0 + rr;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          // horizontal pass for (r, xx[c]): three channels at once
          int h0 = 1 << (PP_PRECISION_BITS - 1), h1 = h0, h2 = h0;
          const int* k = kh + (size_t)(xx[c] < 0 ? 0 : xx[c]) * KS;
          const unsigned char* p = lrow + xmin[c] * 3;
          for (int i = 0; i < xcnt[c]; ++i, p += 3) {
            const int kc = k[i];
            h0 += p[0] * kc;
            h1 += p[1] * kc;
            h2 += p[2] * kc;
          }
          h0 = pp_clip8(h0); h1 = pp_clip8(h1); h2 = pp_clip8(h2);
          // vertical pass: row r feeds the output rows whose tap window contains it
#pragma unroll
          for (int j = 0; j < TB; ++j) {
            const int y = y0 + j - g.pt;
            if (y >= ya && y < yb) {
              const int idx = r - bv[2 * y];
              if ((unsigned)idx < (unsigned)bv[2 * y + 1]) {
                const int kc = kv[(size_t)y * KS + idx];
                acc[c][j][0] += h0 * kc;
                acc[c][j][1] += h1 * kc;
                acc[c][j][2] += h2 * kc;
              }
            }
          }
        }
      }
    }
  }
  // to_tensor + normalize; everything outside the pasted image is the zero canvas of ImageOps.expand
  const size_t plane = (size_t)S * S;
  float* ob = out + (size_t)b * 3 * plane;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int ox = tid + c * PP_THREADS;
    if (ox >= S) continue;
    const bool col_in = xx[c] >= 0 && xx[c] < g.nw;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int oy = y0 + j;
      if (oy >= S) continue;
      const int y = oy - g.pt;
      const bool in = any_rows && col_in && y >= ya && y < yb;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const int u = in ? pp_clip8(acc[c][j][ch]) : 0;
        ob[ch * plane + (size_t)oy * S + ox] = ((float)u / 255.0f - mean[ch]) / stdv[ch];
      }
    }
  }
}

extern "C" {

int vdk_preprocess_workspace_bytes(int32_t B, int32_t S, int32_t max_side, size_t* bytes) {
  if (!bytes || B <= 0 || S <= 0 || S > 4 * PP_THREADS || max_side <= 0) return vdk_fail(VDK_EINVAL, "vdk_preprocess_workspace_bytes: bad argument");
  if (max_side > PP_MAX_SIDE) return vdk_fail(VDK_EUNSUPPORTED, "vdk_preprocess_workspace_bytes: image side above 8192");
  const int KS = pp_ksize_bound(max_side, S);
  *bytes = (size_t)B * sizeof(PPGeom) + (size_t)B * pp_image_stride(S, KS) * sizeof(int);
  return VDK_OK;
}

int vdk_preprocess_resize_pad_normalize(const uint8_t* pixels, const int64_t* offsets, const int32_t* wh, int32_t B, int32_t S, int32_t max_side, float mean0,
                                        float mean1, float mean2, float std0, float std1, float std2, float* out, int32_t* status, void* ws, size_t ws_bytes,
                                        void* stream) {
  if (!pixels || !offsets || !wh || !out || !ws || B <= 0 || S <= 0 || S > 4 * PP_THREADS || max_side <= 0)
    return vdk_fail(VDK_EINVAL, "vdk_preprocess_resize_pad_normalize: bad argument");
  if (max_side > PP_MAX_SIDE) return vdk_fail(VDK_EUNSUPPORTED, "vdk_preprocess_resize_pad_normalize: image side above 8192");
  if (std0 == 0.f || std1 == 0.f || std2 == 0.f) return vdk_fail(VDK_EINVAL, "vdk_preprocess_resize_pad_normalize: std evaluated to zero");   // torchvision's ValueError
  size_t need = 0;
  vdk_preprocess_workspace_bytes(B, S, max_side, &need);
  if (ws_bytes < need) return vdk_fail(VDK_EWORKSPACE, "vdk_preprocess_resize_pad_normalize: workspace too small");
  const int KS = pp_ksize_bound(max_side, S);
  PPGeom* geom = (PPGeom*)ws;
  int* tabs = (int*)(geom + B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pp_coeff_kernel, dim3((unsigned)B, 2), dim3(PP_THREADS), 0, st, (const int*)wh, (int)S, KS, geom, tabs, (int*)status);
  if (int e = vdk_check_launch("vdk_preprocess_resize_pad_normalize(coefficients)")) return e;
#define PP_LAUNCH(CPT, TB)                                                                                                                                    \
  hipLaunchKernelGGL((pp_resample_kernel<CPT, TB>), dim3((unsigned)((S + TB - 1) / TB), (unsigned)B), dim3(PP_THREADS), 0, st, (const unsigned char*)pixels, \
                     (const long*)offsets, (int)S, KS, (const PPGeom*)geom, (const int*)tabs, mean0, mean1, mean2, std0, std1, std2, out)
  if (S <= PP_THREADS) PP_LAUNCH(1, 16);
  else if (S <= 2 * PP_THREADS) PP_LAUNCH(2, 16);
  else PP_LAUNCH(4, 8);
#undef PP_LAUNCH
  return vdk_check_launch("vdk_preprocess_resize_pad_normalize");
}

}  // extern "C"

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

#define PP_KREG 8   // horizontal taps kept in registers (scale <= 3.5); wider windows read their coefficients from the table
#define PP_KV 24    // vertical taps per output row kept in LDS (scale <= 11.5)

// CPT output columns per thread (S <= CPT * 256), TB output rows per workgroup, NV 16-byte staging loads per thread and chunk: the chunk buffer holds
// NV * 4 KB (at least one input row), at most 2 * NV rows
template <int CPT, int TB, int NV>
__global__ __launch_bounds__(PP_THREADS, CPT == 1 ? 3 : 1) void pp_resample_kernel(const unsigned char* __restrict__ pixels, const long* __restrict__ offsets, int S, int KS,
                                                                 const PPGeom* __restrict__ geom, const int* __restrict__ tabs, float m0, float m1, float m2, float s0,
                                                                 float s1, float s2, float* __restrict__ out) {
  constexpr int PP_LDS_BYTES = NV * 16 * PP_THREADS, PP_RB = 2 * NV;
  __shared__ __attribute__((aligned(16))) unsigned int stage[PP_LDS_BYTES / 4 + 16];
  __shared__ unsigned int hbuf[PP_RB][CPT * PP_THREADS];        // horizontally resampled rows of the chunk, RGB packed per column (read back by the owner only)
  const int b = blockIdx.y, y0 = blockIdx.x * TB, tid = threadIdx.x;
  const PPGeom g = geom[b];
  const int* kh = tabs + (size_t)b * pp_image_stride(S, KS);
  const int* bh = kh + (size_t)S * KS;
  const int* kv = bh + 2 * S;
  const int* bv = kv + (size_t)S * KS;
  __shared__ float lut[3][256];                                 // to_tensor + normalize of every uint8 value, per channel
  __shared__ int kvl[TB][PP_KV];                                // vertical coefficients of the band's rows (tap windows up to PP_KV rows)
  {
    const float u = (float)tid / 255.0f;                        // PP_THREADS == 256
    lut[0][tid] = (u - m0) / s0;
    lut[1][tid] = (u - m1) / s1;
    lut[2][tid] = (u - m2) / s2;
  }

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
  int xmin[CPT], xcnt[CPT], xx[CPT], kreg[CPT][PP_KREG];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    xx[c] = tid + c * PP_THREADS - g.pl;
    const bool on = any_rows && xx[c] >= 0 && xx[c] < g.nw;
    xmin[c] = on ? bh[2 * xx[c]] : 0;
    xcnt[c] = on ? bh[2 * xx[c] + 1] : 0;
#pragma unroll
    for (int i = 0; i < PP_KREG; ++i) kreg[c][i] = i < xcnt[c] ? kh[(size_t)xx[c] * KS + i] : 0;
  }
  for (int t = tid; t < TB * PP_KV; t += PP_THREADS) {
    const int j = t / PP_KV, i = t % PP_KV, y = y0 + j - g.pt;
    kvl[j][i] = (any_rows && y >= ya && y < yb && i < bv[2 * y + 1]) ? kv[(size_t)y * KS + i] : 0;
  }
  __syncthreads();
  if (any_rows) {
    // tap windows of the band's output rows (uniform)
    int vmin[TB], vcnt[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int y = y0 + j - g.pt;
      const bool on = y >= ya && y < yb;
      vmin[j] = on ? bv[2 * y] : 0;
      vcnt[j] = on ? bv[2 * y + 1] : 0;
    }
    const int rbeg = bv[2 * ya], rend = bv[2 * (yb - 1)] + bv[2 * (yb - 1) + 1];
    const int w3 = g.w * 3;
    int rb = (PP_LDS_BYTES - 32) / w3;               // rows per staged chunk
    if (rb > PP_RB) rb = PP_RB;
    const unsigned char* img = pixels + offsets[b];
    for (int r0 = rbeg; r0 < rend; r0 += rb) {
      const int nr = rend - r0 < rb ? rend - r0 : rb;
      const unsigned char* src = img + (size_t)r0 * w3;
      const size_t mis = (size_t)src & 15;
      const u32x4* src16 = (const u32x4*)(src - mis);
      const int n16 = (int)((mis + (size_t)nr * w3 + 15) / 16);          // <= PP_LDS_BYTES / 16
      // all loads of the chunk in flight before the first LDS write (a load -> store loop would serialise on the HBM latency)
      u32x4 v[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (tid + i * PP_THREADS < n16) v[i] = src16[tid + i * PP_THREADS];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (tid + i * PP_THREADS < n16) ((u32x4*)stage)[tid + i * PP_THREADS] = v[i];
      __syncthreads();
      // horizontal pass of the chunk's rows: three channels per tap, result rounded to uint8 like Pillow's intermediate image
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const unsigned char* lrow = (const unsigned char*)stage + mis + xmin[c] * 3;
        if (xcnt[c] <= PP_KREG) {
          for (int rr = 0; rr < nr; ++rr, lrow += w3) {
            int h0 = 1 << (PP_PRECISION_BITS - 1), h1 = h0, h2 = h0;
#pragma unroll
            for (int i = 0; i < PP_KREG; ++i) {          // taps beyond the window carry a zero coefficient (the bytes they read lie inside `stage`)
              h0 += __mul24((int)lrow[3 * i], kreg[c][i]);
              h1 += __mul24((int)lrow[3 * i + 1], kreg[c][i]);
              h2 += __mul24((int)lrow[3 * i + 2], kreg[c][i]);
            }
            hbuf[rr][tid + c * PP_THREADS] = (unsigned)pp_clip8(h0) | ((unsigned)pp_clip8(h1) << 8) | ((unsigned)pp_clip8(h2) << 16);
          }
        } else {
          const int* k = kh + (size_t)xx[c] * KS;
          for (int rr = 0; rr < nr; ++rr, lrow += w3) {
            int h0 = 1 << (PP_PRECISION_BITS - 1), h1 = h0, h2 = h0;
            const unsigned char* p = lrow;
            for (int i = 0; i < xcnt[c]; ++i, p += 3) {
              const int kc = k[i];
              h0 += __mul24((int)p[0], kc);
              h1 += __mul24((int)p[1], kc);
              h2 += __mul24((int)p[2], kc);
            }
            hbuf[rr][tid + c * PP_THREADS] = (unsigned)pp_clip8(h0) | ((unsigned)pp_clip8(h1) << 8) | ((unsigned)pp_clip8(h2) << 16);
          }
        }
      }
      // vertical pass: every output row of the band takes the chunk rows inside its tap window
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        int lo = vmin[j] - r0, hi = vmin[j] + vcnt[j] - r0;
        if (lo < 0) lo = 0;
        if (hi > nr) hi = nr;
        const bool in_lds = vcnt[j] <= PP_KV;
        const int* kvj = in_lds ? &kvl[j][0] + (r0 - vmin[j]) : kv + (size_t)(y0 + j - g.pt) * KS + (r0 - vmin[j]);
        for (int rr = lo; rr < hi; ++rr) {
          const int kc = kvj[rr];
#pragma unroll
          for (int c = 0; c < CPT; ++c) {
            const unsigned px = hbuf[rr][tid + c * PP_THREADS];
            acc[c][j][0] += __mul24((int)(px & 255u), kc);
            acc[c][j][1] += __mul24((int)((px >> 8) & 255u), kc);
            acc[c][j][2] += __mul24((int)((px >> 16) & 255u), kc);
          }
        }
      }
    }
  }
  // to_tensor + normalize; everything outside the pasted image is the zero canvas of ImageOps.expand
  const size_t plane = (size_t)S * S;
  float* ob = out + (size_t)b * 3 * plane + (size_t)y0 * S;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int ox = tid + c * PP_THREADS;
    const bool col_in = any_rows && xx[c] >= 0 && xx[c] < g.nw;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int y = y0 + j - g.pt;
      const bool in = col_in && y >= ya && y < yb;
      const int u0 = in ? pp_clip8(acc[c][j][0]) : 0, u1 = in ? pp_clip8(acc[c][j][1]) : 0, u2 = in ? pp_clip8(acc[c][j][2]) : 0;
      if (ox < S && y0 + j < S) {
        ob[(size_t)j * S + ox] = lut[0][u0];
        ob[plane + (size_t)j * S + ox] = lut[1][u1];
        ob[2 * plane + (size_t)j * S + ox] = lut[2][u2];
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
#define PP_LAUNCH(CPT, TB, NV)                                                                                                                                    \
  hipLaunchKernelGGL((pp_resample_kernel<CPT, TB, NV>), dim3((unsigned)((S + TB - 1) / TB), (unsigned)B), dim3(PP_THREADS), 0, st, (const unsigned char*)pixels, \
                     (const long*)offsets, (int)S, KS, (const PPGeom*)geom, (const int*)tabs, mean0, mean1, mean2, std0, std1, std2, out)
  // NV = 8: 32 KB chunk buffer (one 8192-pixel row fits), 16 rows per chunk; a 16 KB / 8-row variant at 4 waves per SIMD measured 11 % slower (spills, twice the chunks)
  if (S <= PP_THREADS) PP_LAUNCH(1, 16, 8);
  else if (S <= 2 * PP_THREADS) PP_LAUNCH(2, 16, 8);
  else PP_LAUNCH(4, 8, 8);
#undef PP_LAUNCH
  return vdk_check_launch("vdk_preprocess_resize_pad_normalize");
}

}  // extern "C"

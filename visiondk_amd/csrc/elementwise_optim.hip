// elementwise_optim.hip — the HBM-bound glue of hot path A, each a single wide-access pass:
//   * patchify+cast: im2col-free operand for PatchEmbed's k=stride conv (timm PatchEmbed.proj,
//     Conv2d(3, D, p, p); reference builds it at models/classifier/classify_model.py:49-54)
//   * cls-token rows, weight casts (fp32 master -> bf16 [out,in] and transposed [in,out])
//   * K16/K15/K17 fused multi-tensor step over FLAT parameter buffers: global-norm clip
//     (engine/procedure/train.py:209 clip_grad_norm_(max_norm=10)), SGD momentum + weight decay
//     (engine/optimizer.py:119-121 -> torch.optim.SGD), EMA (models/ema.py:28-37), bf16 weight refresh
//   * K18 mixup blend (engine/procedure/train.py:24-32)
#include <hip/hip_runtime.h>
#include "vdk_device.h"
#include "vdk_host.h"

// P[(b*gh*gw + gy*gw + gx)][c*ps*ps + ky*ps + kx] = x[b][c][gy*ps+ky][gx*ps+kx]   (bf16, K padded to Kp with 0)
template <int OF = 0>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, int B, int Cin, int H, int W, int ps,
                                                       bf16_t* __restrict__ out, int Kp) {
  const int gh = H / ps, gw = W / ps, K = Cin * ps * ps;
  const long nchunk = (long)B * gh * gw * (Kp / 8);
  long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= nchunk) return;
  const int kc = (int)(id % (Kp / 8)) * 8;
  const long m = id / (Kp / 8);
  const int gx = (int)(m % gw), gy = (int)((m / gw) % gh), b = (int)(m / ((long)gw * gh));
  float v[8];
  if ((ps & 7) == 0 && kc + 8 <= K) {
    const int c = kc / (ps * ps), rem = kc % (ps * ps), ky = rem / ps, kx = rem % ps;
    const float* src = x + (((long)b * Cin + c) * H + gy * ps + ky) * W + gx * ps + kx;
    if ((((size_t)src) & 15) == 0) {
      f32x4 a = *(const f32x4*)src, bq = *(const f32x4*)(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = bq[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kc + e;
      float t = 0.f;
      if (k < K) {
        const int c = k / (ps * ps), rem = k % (ps * ps), ky = rem / ps, kx = rem % ps;
        t = x[(((long)b * Cin + c) * H + gy * ps + ky) * W + gx * ps + kx];
      }
      v[e] = t;
    }
  }
  *(u32x4*)(out + m * Kp + kc) = (u32x4){pack_op2<OF>(v[0], v[1]), pack_op2<OF>(v[2], v[3]), pack_op2<OF>(v[4], v[5]), pack_op2<OF>(v[6], v[7])};
}

// tok[b][0][:] = cls + pos[0]
__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ tok, long batch_stride, int B, int D,
                                                       const float* __restrict__ cls, const float* __restrict__ pos0) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * D) return;
  int b = (int)(i / D), d = (int)(i % D);
  tok[(long)b * batch_stride + d] = cls[d] + pos0[d];
}

template <int OF = 0>
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = *(const f32x4*)(in + i * 4);
  *(u32x2*)(out + i * 4) = (u32x2){pack_op2<OF>(v[0], v[1]), pack_op2<OF>(v[2], v[3])};
}

// out[c][r] (bf16, ld ldo) = in[r][c] (fp32, ld ldi); zero-fills rows r in [R, Rpad) of the output columns
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ in, long ldi, int R, int Cc,
                                                             bf16_t* __restrict__ out, long ldo, int Rpad) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int i = 0; i < 16; ++i) {
    int row = i * 4 + (tid >> 6), col = tid & 63;
    int gr = r0 + row, gc = c0 + col;
    tile[row][col] = (gr < R && gc < Cc) ? in[(long)gr * ldi + gc] : 0.f;
  }
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    int col = i * 4 + (tid >> 6), row = tid & 63;  // output row = input col
    int gc = c0 + col, gr = r0 + row;
    if (gc < Cc && gr < Rpad) out[(long)gc * ldo + gr] = f2bf(tile[row][col]);
  }
}

// the same for a table of jobs passed by value (<= TC_MAX per launch): block -> job by a scan over the tile prefix sums
#define TC_MAX 64
struct TcTable { VdkTcItem it[TC_MAX]; int tile0[TC_MAX + 1]; int n; };
template <int OF = 0>
__global__ __launch_bounds__(256) void transpose_cast_batch_kernel(TcTable t) {
  __shared__ float tile[64][65];
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.tile0[j + 1]) ++j;
  const VdkTcItem& a = t.it[j];
  const int local = blockIdx.x - t.tile0[j], ntr = (a.Rpad + 63) / 64;
  const int r0 = (local % ntr) * 64, c0 = (local / ntr) * 64, tid = threadIdx.x;
  const float* in = a.in; bf16_t* out = (bf16_t*)a.out;
  // wide form (round 6: 172 -> ~95 us for ViT-B/16's 86 M parameters): 16-byte loads along the input rows, 16-byte stores (8 values) along the output rows; the scalar
  // form below serves ragged shapes and unaligned pitches.  Same values either way (a transpose and one rounding per element).
  const bool wide_in = !(a.ldi & 3) && !(a.C & 3) && !((unsigned long long)in & 15);
  const bool wide_out = !(a.ldo & 7) && !(a.Rpad & 7) && !((unsigned long long)out & 15);
  if (wide_in) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 16 + (tid >> 4), col = (tid & 15) * 4;
      const int gr = r0 + row, gc = c0 + col;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gr < a.R && gc < a.C) v = *(const f32x4*)(in + (long)gr * a.ldi + gc);      // (C % 4 == 0: a chunk is inside the row or beyond it)
      tile[row][col] = v[0]; tile[row][col + 1] = v[1]; tile[row][col + 2] = v[2]; tile[row][col + 3] = v[3];
    }
  } else {
    for (int i = 0; i < 16; ++i) {
      int row = i * 4 + (tid >> 6), col = tid & 63;
      int gr = r0 + row, gc = c0 + col;
      tile[row][col] = (gr < a.R && gc < a.C) ? in[(long)gr * a.ldi + gc] : 0.f;
    }
  }
  __syncthreads();
  if (wide_out) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = i * 32 + (tid >> 3), row = (tid & 7) * 8;
      const int gc = c0 + col, gr = r0 + row;
      if (gc < a.C && gr < a.Rpad) {                                                   // (Rpad % 8 == 0: a chunk is inside the padded row or beyond it)
        const u32x4 o = {pack_op2<OF>(tile[row][col], tile[row + 1][col]), pack_op2<OF>(tile[row + 2][col], tile[row + 3][col]),
                         pack_op2<OF>(tile[row + 4][col], tile[row + 5][col]), pack_op2<OF>(tile[row + 6][col], tile[row + 7][col])};
        *(u32x4*)(out + (long)gc * a.ldo + gr) = o;
      }
    }
  } else {
    for (int i = 0; i < 16; ++i) {
      int col = i * 4 + (tid >> 6), row = tid & 63;
      int gc = c0 + col, gr = r0 + row;
      if (gc < a.C && gr < a.Rpad) out[(long)gc * a.ldo + gr] = f2op<OF>(tile[row][col]);
    }
  }
}

template <int OF = 0>
__global__ __launch_bounds__(256) void cast_pad_rows_kernel(const float* __restrict__ in, long ldi, int R, int Cc, bf16_t* __restrict__ out, long ldo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * ldo) return;
  const int c = (int)(i % ldo);
  const long r = i / ldo;
  out[i] = f2op<OF>(c < Cc ? in[r * ldi + c] : 0.f);
}
// partial[b] = sum of g[i]^2 over the block's slice
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = *(const f32x4*)(g + i * 4);
    s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
  }
  if (blockIdx.x == 0) for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s = fmaf(g[i], g[i], s);
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

struct SgdArgs {
  float* p; const float* g; float* m; float* ema; bf16_t* pb;
  long n;
  float lr, momentum, weight_decay, max_norm, ema_decay, grad_scale;
  const float* normsq;   // device scalar: sum g^2 (before grad_scale); NULL = no clipping
  int first_step;
  const float* hyper;    // device [5] or NULL: lr, momentum, weight_decay, ema_decay, first_step (!= 0) override the by-value fields, so that a step
                         // captured in a hipGraph can be replayed while the schedule and the EMA warm-up move on
  const float* loss_scale;   // device scalar or NULL: GradScaler's scale S -- the gradients (and normsq) carry it: g / S is what is clipped and applied
                             // (scaler.unscale_, train.py:208), and a non-finite sum g^2 skips the whole update (scaler.step, train.py:210)
};
// torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max=1); g *= coef
// torch.optim.SGD (nesterov=False, dampening=0): g += wd*p; buf = first ? g : mom*buf + g; p -= lr*buf
// ModelEMA.update: v = d*v + (1-d)*p
template <int OF = 0>
__global__ __launch_bounds__(256) void sgd_step_kernel(SgdArgs a) {
  if (a.hyper) { a.lr = a.hyper[0]; a.momentum = a.hyper[1]; a.weight_decay = a.hyper[2]; a.ema_decay = a.hyper[3]; a.first_step = a.hyper[4] != 0.f; }
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n4 = (a.n + 3) >> 2;
  if (i >= n4) return;
  float coef = a.grad_scale;
  if (a.loss_scale) {
    // inf / NaN anywhere in the scaled gradient makes its sum of squares non-finite: GradScaler.step skips optimizer.step(); the reference's loop still runs
    // ema.update(model) on the unchanged weights (train.py:210-215), and so does this pass
    if (a.normsq && !(fabsf(a.normsq[0]) < 3.0e38f)) {
      if (!a.ema) return;
      const long base = i * 4;
      for (long j = base; j < a.n && j < base + 4; ++j) a.ema[j] = a.ema[j] * a.ema_decay + (1.0f - a.ema_decay) * a.p[j];
      return;
    }
    a.grad_scale /= a.loss_scale[0];
    coef = a.grad_scale;
  }
  if (a.normsq) {
    float nrm = sqrtf(a.normsq[0]) * fabsf(a.grad_scale);
    float c = a.max_norm / (nrm + 1e-6f);
    coef *= (c < 1.0f ? c : 1.0f);
  }
  const long base = i * 4;
  if (base + 4 <= a.n) {
    f32x4 p = *(const f32x4*)(a.p + base), g = *(const f32x4*)(a.g + base);
    f32x4 m = a.first_step ? (f32x4){0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(a.m + base);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = g[e] * coef + a.weight_decay * p[e];
      float buf = a.first_step ? gg : a.momentum * m[e] + gg;
      m[e] = buf;
      p[e] = p[e] - a.lr * buf;
    }
    *(f32x4*)(a.p + base) = p;
    *(f32x4*)(a.m + base) = m;
    if (a.ema) {
      f32x4 v = *(const f32x4*)(a.ema + base);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * a.ema_decay + (1.0f - a.ema_decay) * p[e];
      *(f32x4*)(a.ema + base) = v;
    }
    if (a.pb) *(u32x2*)(a.pb + base) = (u32x2){pack_op2<OF>(p[0], p[1]), pack_op2<OF>(p[2], p[3])};
  } else {
    for (long j = base; j < a.n; ++j) {
      float p = a.p[j];
      float gg = a.g[j] * coef + a.weight_decay * p;
      float buf = a.first_step ? gg : a.momentum * a.m[j] + gg;
      a.m[j] = buf;
      p -= a.lr * buf;
      a.p[j] = p;
      if (a.ema) a.ema[j] = a.ema[j] * a.ema_decay + (1.0f - a.ema_decay) * p;
      if (a.pb) a.pb[j] = f2op<OF>(p);
    }
  }
}

// torch.cuda.amp.GradScaler.update (train.py:211) on the device: state = {scale, growth tracker, skipped steps}.  found_inf = the scaled gradient's sum of squares is
// not finite.  found_inf: scale *= backoff, tracker = 0; else ++tracker == growth_interval: scale *= growth, tracker = 0.  (torch defaults: 65536, 2, 0.5, 2000.)
__global__ void loss_scale_update_kernel(float* __restrict__ state, const float* __restrict__ normsq, float growth, float backoff, int interval) {
  if (threadIdx.x || blockIdx.x) return;
  const bool bad = !(fabsf(normsq[0]) < 3.0e38f);
  float sc = state[0], tr = state[1];
  if (bad) { sc *= backoff; tr = 0.f; state[2] += 1.f; }
  else { tr += 1.f; if (tr >= (float)interval) { const float g = sc * growth; if (g < 3.0e38f) sc = g; tr = 0.f; } }
  state[0] = sc; state[1] = tr;
}

// out[b] = lam * x[b] + (1 - lam) * x[perm[b]]   (mixup_data, train.py:24-32)
__global__ __launch_bounds__(256) void mixup_kernel(const float* __restrict__ x, const long long* __restrict__ perm, float lam,
                                                    long per_sample4, int B, float* __restrict__ out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_sample4 * B) return;
  int b = (int)(i / per_sample4); long o = i % per_sample4;
  f32x4 a = *(const f32x4*)(x + ((long)b * per_sample4 + o) * 4);
  f32x4 c = *(const f32x4*)(x + ((long)perm[b] * per_sample4 + o) * 4);
  *(f32x4*)(out + i * 4) = a * lam + c * (1.0f - lam);
}


// ---- SAM (engine/optimizer.py:29-87): grad norm of (|p| * g) [adaptive] or g, then the climb p += (p^2 | 1) * g * rho / (norm + 1e-12)
__global__ __launch_bounds__(256) void sumsq_prod_partial_kernel(const float* __restrict__ p, const float* __restrict__ g, long n, int adaptive,
                                                                 float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = adaptive ? fabsf(p[i]) * g[i] : g[i];
    s = fmaf(v, v, s);
  }
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sam_perturb_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ old_p, long n,
                                                          float rho, int adaptive, const float* __restrict__ normsq) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float scale = rho / (sqrtf(normsq[0]) + 1e-12f);
  const float w = p[i];
  old_p[i] = w;
  p[i] = w + (adaptive ? w * w : 1.0f) * g[i] * scale;
}

// ---- OHEM (structure/sampler.py:11-31): keep the samples whose target probability is below
// max(sorted_prob[min(min_kept, n-1)], thresh).  Stage 1: target probability per row; stage 2: one workgroup ranks them.
__global__ __launch_bounds__(256) void target_prob_kernel(const float* __restrict__ logits, long ldl, int C, const long long* __restrict__ y,
                                                          long long ignore_index, float* __restrict__ prob) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long)row * ldl;
  float mx = -3.0e38f;
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, x[c]);
  mx = block_max<4>(mx, red);
  float se = 0.f;
  for (int c = tid; c < C; c += 256) se += expf(x[c] - mx);
  se = block_sum<4>(se, red);
  if (tid == 0) {
    const long long t = y[row];
    prob[row] = (t == ignore_index || t < 0 || t >= C) ? __uint_as_float(0x7f800000u) : expf(x[t] - mx) / se;   // +inf = ignored
  }
}
__global__ __launch_bounds__(256) void ohem_mask_kernel(const float* __restrict__ prob, int B, int min_kept, float thresh, unsigned char* __restrict__ mask) {
  __shared__ float s_thr;
  __shared__ int s_valid;
  if (threadIdx.x == 0) { s_valid = 0; s_thr = 0.f; }
  __syncthreads();
  int cnt = 0;
  for (int i = threadIdx.x; i < B; i += 256) cnt += (prob[i] < __uint_as_float(0x7f800000u)) ? 1 : 0;
  atomicAdd(&s_valid, cnt);
  __syncthreads();
  const int nv = s_valid;
  if (nv == 0) { for (int i = threadIdx.x; i < B; i += 256) mask[i] = 0; return; }
  const int kth = min_kept < nv - 1 ? min_kept : nv - 1;
  for (int i = threadIdx.x; i < B; i += 256) {   // the element with exactly `kth` elements ordered before it is sort_prob[kth]
    const float pi = prob[i];
    if (!(pi < __uint_as_float(0x7f800000u))) continue;
    int rank = 0;
    for (int j = 0; j < B; ++j) { const float pj = prob[j]; rank += (pj < pi || (pj == pi && j < i)) ? 1 : 0; }
    if (rank == kth) s_thr = pi;
  }
  __syncthreads();
  const float threshold = fmaxf(s_thr, thresh);
  for (int i = threadIdx.x; i < B; i += 256) mask[i] = (prob[i] < threshold) ? 1 : 0;
}

// ---- K14: top-k of each logits row, descending, ties -> lower index (engine/procedure/evaluation.py:106 uses a full argsort)
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, long ld, int B, int C, int k, long long* __restrict__ idx,
                                                        float* __restrict__ val) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const float* xr = x + (long)row * ld;
  unsigned long long last = ~0ull;   // key of the previously selected element; keys are unique (value, ~index)
  for (int j = 0; j < k; ++j) {
    unsigned long long best = 0ull;
    for (int c = lane; c < C; c += 64) {
      const unsigned long long key = ((unsigned long long)f2ord(xr[c]) << 32) | (unsigned)(0x7fffffff - c);
      if (key < last && key > best) best = key;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { unsigned long long o = __shfl_xor(best, m); best = o > best ? o : best; }
    last = best;
    if (lane == 0) {
      const int c = 0x7fffffff - (int)(unsigned)(best & 0xffffffffu);
      idx[(long)row * k + j] = (j < C) ? c : -1;
      if (val) val[(long)row * k + j] = (j < C) ? ord2f((unsigned)(best >> 32)) : -3.4028234663852886e38f;
    }
  }
}

int vdk_transpose_cast_batch(const VdkTcItem* items, int n, void* stream, int opf) {
  if (!items || n < 0) return vdk_fail(VDK_EINVAL, "vdk_transpose_cast_batch: bad argument");
  for (int base = 0; base < n; base += TC_MAX) {
    TcTable t;
    t.n = n - base < TC_MAX ? n - base : TC_MAX;
    int tiles = 0;
    for (int j = 0; j < t.n; ++j) {
      const VdkTcItem& a = items[base + j];
      if (!a.in || !a.out || a.R <= 0 || a.C <= 0 || a.Rpad < a.R || a.ldo < a.Rpad) return vdk_fail(VDK_EINVAL, "vdk_transpose_cast_batch: bad item");
      t.it[j] = a; t.tile0[j] = tiles;
      tiles += ((a.Rpad + 63) / 64) * ((a.C + 63) / 64);
    }
    t.tile0[t.n] = tiles;
    if (opf) hipLaunchKernelGGL(transpose_cast_batch_kernel<VDK_OPF_F16>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, t);
    else hipLaunchKernelGGL(transpose_cast_batch_kernel<0>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, t);
  }
  return vdk_check_launch("vdk_transpose_cast_batch");
}

int vdk_cast_pad_rows(const float* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, void* stream, int opf) {
  if (!in || !out || R <= 0 || C <= 0 || ldo < C || ldi < C) return vdk_fail(VDK_EINVAL, "vdk_cast_pad_rows: bad argument");
  if (opf) hipLaunchKernelGGL(cast_pad_rows_kernel<VDK_OPF_F16>, dim3((unsigned)(((long)R * ldo + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, (long)ldi, (int)R, (int)C,
                              (bf16_t*)out, (long)ldo);
  else
  hipLaunchKernelGGL(cast_pad_rows_kernel<0>, dim3((unsigned)(((long)R * ldo + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, (long)ldi, (int)R, (int)C, (bf16_t*)out,
                     (long)ldo);
  return vdk_check_launch("vdk_cast_pad_rows");
}

int vdk_patchify_16(const float* x, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* out, int32_t Kp, int opf, void* stream) {
  if (!x || !out || B <= 0 || Cin <= 0 || patch <= 0 || H % patch || W % patch || (Kp & 7) || Kp < Cin * patch * patch)
    return vdk_fail(VDK_EINVAL, "vdk_patchify_bf16: bad argument");
  long nchunk = (long)B * (H / patch) * (W / patch) * (Kp / 8);
  if (opf) hipLaunchKernelGGL(patchify_kernel<VDK_OPF_F16>, dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (int)B, (int)Cin, (int)H, (int)W, (int)patch,
                              (bf16_t*)out, (int)Kp);
  else hipLaunchKernelGGL(patchify_kernel<0>, dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (int)B, (int)Cin,
                          (int)H, (int)W, (int)patch, (bf16_t*)out, (int)Kp);
  return vdk_check_launch("vdk_patchify_bf16");
}
int vdk_cast_f32_16(const float* in, void* out, int64_t n, int opf, void* stream) {
  if (!in || !out || n < 0 || (n & 3)) return vdk_fail(VDK_EINVAL, "vdk_cast_f32_bf16: n % 4 == 0 required");
  if (n == 0) return VDK_OK;
  if (opf) hipLaunchKernelGGL(cast_f32_bf16_kernel<VDK_OPF_F16>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (long)(n / 4));
  else hipLaunchKernelGGL(cast_f32_bf16_kernel<0>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (long)(n / 4));
  return vdk_check_launch("vdk_cast_f32_bf16");
}

// x16[r, c] *= rs[r / rps] in place (C % 8 == 0, 16-byte chunks): see vdk_rowscale_16 in vdk_host.h
template <int OF>
__global__ __launch_bounds__(256) void rowscale16_kernel(bf16_t* __restrict__ x, long T, int C, const float* __restrict__ rs, int rps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int c8 = C / 8;
  if (i >= T * c8) return;
  const long r = i / c8;
  const float f = rs[r / rps];
  u32x4 v = *(u32x4*)(x + i * 8);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = pack_op2<OF>(op_lo<OF>(v[e]) * f, op_hi<OF>(v[e]) * f);
  *(u32x4*)(x + i * 8) = v;
}
int vdk_rowscale_16(void* x16, int64_t T, int32_t C, const float* rs, int32_t rps, int opf, void* stream) {
  if (!x16 || !rs || T <= 0 || C <= 0 || (C & 7) || rps <= 0) return vdk_fail(VDK_EINVAL, "vdk_rowscale_16: bad argument (C % 8 == 0)");
  const long n = T * (C / 8);
  if (opf) hipLaunchKernelGGL(rowscale16_kernel<VDK_OPF_F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x16, (long)T, (int)C, rs, (int)rps);
  else hipLaunchKernelGGL(rowscale16_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x16, (long)T, (int)C, rs, (int)rps);
  return vdk_check_launch("vdk_rowscale_16");
}

extern "C" {

int vdk_patchify_bf16(const float* x, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* out, int32_t Kp,
                      void* stream) {
  return vdk_patchify_16(x, B, Cin, H, W, patch, out, Kp, VDK_OPF_BF16, stream);
}

int vdk_cls_rows(float* tok, int64_t batch_stride, int32_t B, int32_t D, const float* cls, const float* pos0, void* stream) {
  if (!tok || !cls || !pos0 || B <= 0 || D <= 0) return vdk_fail(VDK_EINVAL, "vdk_cls_rows: bad argument");
  hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)(((long)B * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tok,
                     (long)batch_stride, (int)B, (int)D, cls, pos0);
  return vdk_check_launch("vdk_cls_rows");
}

int vdk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) { return vdk_cast_f32_16(in, out, n, VDK_OPF_BF16, stream); }
int vdk_cast_f32_f16(const float* in, void* out, int64_t n, void* stream) { return vdk_cast_f32_16(in, out, n, VDK_OPF_F16, stream); }

int vdk_transpose_cast_f32_bf16(const float* in, int64_t ldi, int32_t R, int32_t C, void* out, int64_t ldo, int32_t Rpad,
                                void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || Rpad < R || ldo < Rpad) return vdk_fail(VDK_EINVAL, "vdk_transpose_cast_f32_bf16: bad argument");
  hipLaunchKernelGGL(transpose_cast_kernel, dim3((unsigned)((Rpad + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, in, (long)ldi, (int)R, (int)C, (bf16_t*)out, (long)ldo, (int)Rpad);
  return vdk_check_launch("vdk_transpose_cast_f32_bf16");
}

#define SUMSQ_BLOCKS 1024
int vdk_sumsq_workspace_bytes(size_t* bytes) { if (!bytes) return vdk_fail(VDK_EINVAL, "null"); *bytes = SUMSQ_BLOCKS * 4; return VDK_OK; }
// out[0] = sum g[i]^2 (deterministic two-stage reduce); g 16-B aligned
int vdk_sumsq_f32(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !out || n < 0) return vdk_fail(VDK_EINVAL, "vdk_sumsq_f32: bad argument");
  if (!ws || ws_bytes < SUMSQ_BLOCKS * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_sumsq_f32: workspace too small");
  long n4 = n / 4;
  int nb = (int)((n4 + 255) / 256); if (nb > SUMSQ_BLOCKS) nb = SUMSQ_BLOCKS; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, g, (long)n, (float*)ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, nb, out);
  return vdk_check_launch("vdk_sumsq_f32");
}

// One fused pass over flat buffers: [clip by global norm] -> SGD(momentum, wd) -> [EMA] -> [bf16 refresh].
// grad_scale multiplies g first (1/loss_scale, or 1/world for summed all-reduce).  normsq = device scalar
// holding sum g^2 of the UNscaled grads, or NULL to skip clipping.  m / ema / params_bf16 may be NULL.
static int sgd_launch(const char* what, float* params, const float* grads, float* momentum_buf, float* ema, void* params_bf16, int64_t n, float lr, float momentum,
                      float weight_decay, float grad_scale, const float* normsq, float max_norm, float ema_decay, int32_t first_step, const float* hyper,
                      void* stream, int pb_opf = 0, const float* loss_scale = nullptr) {
  if (!params || !grads || !momentum_buf || n < 0) return vdk_fail(VDK_EINVAL, "vdk_sgd_step: bad argument");
  if (n == 0) return VDK_OK;
  SgdArgs a;
  a.p = params; a.g = grads; a.m = momentum_buf; a.ema = ema; a.pb = (bf16_t*)params_bf16; a.n = n;
  a.lr = lr; a.momentum = momentum; a.weight_decay = weight_decay; a.max_norm = max_norm; a.ema_decay = ema_decay;
  a.grad_scale = grad_scale; a.normsq = normsq; a.first_step = first_step; a.hyper = hyper; a.loss_scale = loss_scale;
  long n4 = (n + 3) / 4;
  if (pb_opf) hipLaunchKernelGGL(sgd_step_kernel<VDK_OPF_F16>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(sgd_step_kernel<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return vdk_check_launch(what);
}
int vdk_sgd_step(float* params, const float* grads, float* momentum_buf, float* ema, void* params_bf16, int64_t n, float lr,
                 float momentum, float weight_decay, float grad_scale, const float* normsq, float max_norm, float ema_decay,
                 int32_t first_step, void* stream) {
  return sgd_launch("vdk_sgd_step", params, grads, momentum_buf, ema, params_bf16, n, lr, momentum, weight_decay, grad_scale, normsq, max_norm, ema_decay, first_step,
                    nullptr, stream);
}
// The same pass with lr, momentum, weight_decay, ema_decay and first_step read from DEVICE memory (hyper f32 [5]) at run time: the form a step captured in a
// hipGraph uses, because by-value kernel arguments are frozen at capture while the LR schedule and ModelEMA's warm-up decay change every step.
int vdk_sgd_step_graph(float* params, const float* grads, float* momentum_buf, float* ema, void* params_bf16, int64_t n, const float* hyper, float grad_scale,
                       const float* normsq, float max_norm, void* stream) {
  if (!hyper) return vdk_fail(VDK_EINVAL, "vdk_sgd_step_graph: hyper is NULL");
  return sgd_launch("vdk_sgd_step_graph", params, grads, momentum_buf, ema, params_bf16, n, 0.f, 0.f, 0.f, grad_scale, normsq, max_norm, 0.f, 0, hyper, stream);
}

// vdk_sgd_step under GradScaler (train.py:205-211): the gradients carry the loss scale loss_state[0] (device); they are un-scaled, clipped and applied in the same pass,
// the whole update is skipped when normsq is not finite, and params16 is refreshed in `p16_dtype` (VDK_BF16 | VDK_F16).  Follow with vdk_loss_scale_update.
int vdk_sgd_step_amp(float* params, const float* grads, float* momentum_buf, float* ema, void* params16, int32_t p16_dtype, int64_t n, float lr, float momentum,
                     float weight_decay, float grad_scale, const float* loss_state, const float* normsq, float max_norm, float ema_decay, int32_t first_step, void* stream) {
  if (p16_dtype != VDK_BF16 && p16_dtype != VDK_F16) return vdk_fail(VDK_EINVAL, "vdk_sgd_step_amp: bad p16_dtype");
  if (loss_state && !normsq) return vdk_fail(VDK_EINVAL, "vdk_sgd_step_amp: a loss scale needs normsq (the overflow check reads it)");
  return sgd_launch("vdk_sgd_step_amp", params, grads, momentum_buf, ema, params16, n, lr, momentum, weight_decay, grad_scale, normsq, max_norm, ema_decay, first_step,
                    nullptr, stream, p16_dtype == VDK_F16 ? VDK_OPF_F16 : VDK_OPF_BF16, loss_state);
}
// GradScaler.update (train.py:211): loss_state f32 [3] = {scale, growth tracker, skipped steps} on the device, normsq = the sum of squares vdk_sgd_step_amp looked at
int vdk_loss_scale_update(float* loss_state, const float* normsq, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream) {
  if (!loss_state || !normsq || !(growth_factor >= 1.0f) || !(backoff_factor > 0.f && backoff_factor <= 1.0f) || growth_interval < 1)
    return vdk_fail(VDK_EINVAL, "vdk_loss_scale_update: bad argument");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_state, normsq, growth_factor, backoff_factor, (int)growth_interval);
  return vdk_check_launch("vdk_loss_scale_update");
}

// SAM.first_step: normsq_out[0] = sum ((|p| | 1) * g)^2, then p_old = p; p += (p^2 | 1) * g * rho / (sqrt(normsq) + 1e-12)
int vdk_sam_first_step(float* params, const float* grads, float* old_params, int64_t n, float rho, int32_t adaptive, float* normsq_out, void* ws,
                       size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!params || !grads || !old_params || !normsq_out || n <= 0) return vdk_fail(VDK_EINVAL, "vdk_sam_first_step: bad argument");
  if (!ws || ws_bytes < SUMSQ_BLOCKS * 4) return vdk_fail(VDK_EWORKSPACE, "vdk_sam_first_step: workspace too small");
  int nb = (int)((n + 255) / 256); if (nb > SUMSQ_BLOCKS) nb = SUMSQ_BLOCKS;
  hipLaunchKernelGGL(sumsq_prod_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)params, grads, (long)n, (int)adaptive, (float*)ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, nb, normsq_out);
  hipLaunchKernelGGL(sam_perturb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, params, grads, old_params, (long)n, rho, (int)adaptive,
                     (const float*)normsq_out);
  return vdk_check_launch("vdk_sam_first_step");
}

// OHEMImageSampler.sample: mask[b] = 1 for the hard examples to keep.  prob_ws: f32 [B] scratch.
int vdk_ohem_mask(const float* logits, int64_t ldl, int32_t B, int32_t C, const int64_t* labels, int32_t min_kept, float thresh, int64_t ignore_index,
                  float* prob_ws, uint8_t* mask, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!logits || !labels || !prob_ws || !mask || B <= 0 || C <= 0) return vdk_fail(VDK_EINVAL, "vdk_ohem_mask: bad argument");
  hipLaunchKernelGGL(target_prob_kernel, dim3((unsigned)B), dim3(256), 0, stream, logits, (long)ldl, (int)C, (const long long*)labels, (long long)ignore_index,
                     prob_ws);
  hipLaunchKernelGGL(ohem_mask_kernel, dim3(1), dim3(256), 0, stream, (const float*)prob_ws, (int)B, (int)min_kept, thresh, (unsigned char*)mask);
  return vdk_check_launch("vdk_ohem_mask");
}

int vdk_topk_rows(const float* x, int64_t ld, int32_t B, int32_t C, int32_t k, int64_t* idx, float* values, void* stream) {
  if (!x || !idx || B <= 0 || C <= 0 || k <= 0 || k > 64) return vdk_fail(VDK_EINVAL, "vdk_topk_rows: bad argument (1 <= k <= 64)");
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long)ld, (int)B, (int)C, (int)k, (long long*)idx,
                     values);
  return vdk_check_launch("vdk_topk_rows");
}

int vdk_mixup(const float* x, const int64_t* perm, float lam, int32_t B, int64_t per_sample, float* out, void* stream) {
  if (!x || !perm || !out || B <= 0 || per_sample <= 0 || (per_sample & 3)) return vdk_fail(VDK_EINVAL, "vdk_mixup: bad argument");
  long tot = per_sample / 4 * B;
  hipLaunchKernelGGL(mixup_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (const long long*)perm, lam,
                     (long)(per_sample / 4), (int)B, out);
  return vdk_check_launch("vdk_mixup");
}

}  // extern "C"

// vdk_device.h — device-side helpers shared by every gfx950 kernel in this library.
// Written for CDNA4 only: wave = 64 lanes, MFMA fragments per cdna_hip_programming.md §3.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define VDK_WAVE 64

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// address-space casts for the LDS-DMA builtin (the CPU SIMT emulation used by tests pre-defines them as plain casts)
#ifndef VDK_LDS_PTR
#define VDK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define VDK_LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))
#define VDK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#endif

// device-scope (all XCDs) relaxed accesses: sc1 write-through stores / L2-bypassing loads, so data handed from one workgroup to another needs no L2 write-back
// fence (a __threadfence() is a buffer_wbl2 + buffer_inv of the whole 4 MB L2 on gfx950: measured +300 us on a 72 us GEMM when every wave issued two)
#ifndef VDK_AGENT_ST_U64
#define VDK_AGENT_ST_U64(p, v) __hip_atomic_store((unsigned long long*)(p), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VDK_AGENT_LD_U64(p) __hip_atomic_load((const unsigned long long*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VDK_AGENT_ST_I32(p, v) __hip_atomic_store((int*)(p), (int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VDK_AGENT_LD_I32(p) __hip_atomic_load((const int*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VDK_AGENT_ADD_I32(p, v) __hip_atomic_fetch_add((int*)(p), (int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// dynamic LDS region of a kernel (16-byte aligned base, cdna_hip_programming.md G17); the test emulator substitutes a per-thread arena
#ifndef VDK_DYN_LDS
#define VDK_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// ds_read_b64_tr_b16 (gfx950 LDS transpose read).  Measured semantics (tools/probes/tr_probe.hip): within each 16-lane
// group, lane i receives element (i % 4) of the 8-byte chunks addressed by lanes i/4, 4 + i/4, 8 + i/4, 12 + i/4.
// tr_frag8() builds an MFMA 32x32x16 A/B fragment from a ROW-major tile X[t][c] (pitch in elements): lane (l & 31 = column
// c0 + (l & 31), hi = l >> 5) gets X[t1 + 0..3][c] in slots 0..3 and X[t2 + 0..3][c] in slots 4..7, i.e. the operand whose
// contraction index runs along the tile's ROWS - no transposed copy of the tile is ever materialised.
__device__ __forceinline__ s16x8 tr_frag8(const bf16_t* tile, int pitch, int t1, int t2, int c0, int lane) {
  const int s = lane & 15, chalf = (lane >> 4) & 1;
  const bf16_t* p1 = tile + (t1 + (s >> 2)) * pitch + c0 + 16 * chalf + 4 * (s & 3);
  const bf16_t* p2 = tile + (t2 + (s >> 2)) * pitch + c0 + 16 * chalf + 4 * (s & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1));
  s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p2));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return r;
}

// pin two MFMA accumulators at a program point: the compiler may not move their producers below / consumers above it
// (hipcc sinks register-only MFMAs across sched_barrier; cdna_hip_programming.md §5.7 item 3)
// wave-synchronous LDS hand-off: the LDS unit executes one wave's ds ops in issue order, so a lane's ds_write is visible to
// the next ds_read of any lane of the SAME wave; only the compiler has to be kept from reordering across the point.
#ifndef VDK_WAVE_LDS_SYNC
#define VDK_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
// value of a 32-bit VGPR in a wave-uniform lane -> SGPR broadcast (v_readlane_b32, no LDS round trip)
#ifndef VDK_READLANE
#define VDK_READLANE(v, l) __builtin_amdgcn_readlane((int)(v), (l))
#endif
#ifndef VDK_PIN2
#define VDK_PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))
#endif

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// f32 -> bf16 through the compiler's native conversion: v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN-preserving)
typedef __bf16 vdk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  f32x2 v = {lo, hi};
  vdk_bf16x2 b = __builtin_convertvector(v, vdk_bf16x2);
  return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// two fp32 -> packed fp16 (round to nearest even) and back
typedef _Float16 vdk_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 vdk_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  const vdk_f16x2 h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float h_lo(unsigned u) { return (float)__builtin_bit_cast(vdk_f16x2, u)[0]; }
__device__ __forceinline__ float h_hi(unsigned u) { return (float)__builtin_bit_cast(vdk_f16x2, u)[1]; }

// ---- the 16-bit OPERAND FORMAT of the GEMM path (template parameter OF of every kernel that reads or writes GEMM operands) -----------------------------
// OF = 0: bfloat16 (BASELINE.json configs[1] "bf16").  OF = 1: IEEE fp16 -- what the reference's `torch.autocast(device_type=...)` (engine/procedure/train.py:118, no
// dtype argument => float16 on a GPU) holds its matmul operands in, with GradScaler's loss scale (train.py:205-211) carrying the gradients through the narrower
// exponent.  Same storage (bf16_t = 16 raw bits), same MFMA rate (v_mfma_f32_32x32x16_f16 = v_mfma_f32_32x32x16_bf16), 8x smaller operand rounding.
#define VDK_OPF_BF16 0
#define VDK_OPF_F16 1
template <int OF> __device__ __forceinline__ unsigned pack_op2(float lo, float hi) { if constexpr (OF == VDK_OPF_F16) return pack_h2(lo, hi); else return pack_bf2(lo, hi); }
template <int OF> __device__ __forceinline__ float op_lo(unsigned w) { if constexpr (OF == VDK_OPF_F16) return h_lo(w); else return bf_lo(w); }
template <int OF> __device__ __forceinline__ float op_hi(unsigned w) { if constexpr (OF == VDK_OPF_F16) return h_hi(w); else return bf_hi(w); }
template <int OF> __device__ __forceinline__ float op2f(bf16_t h) {
  if constexpr (OF == VDK_OPF_F16) return (float)__builtin_bit_cast(_Float16, h); else return bf2f(h);
}
template <int OF> __device__ __forceinline__ bf16_t f2op(float f) {
  if constexpr (OF == VDK_OPF_F16) { const _Float16 h = (_Float16)f; return __builtin_bit_cast(bf16_t, h); } else return f2bf(f);
}
// D = A B + C on 32x32x16 / 16x16x32 tiles of 16-bit operands (fragment layouts are the same for both formats)
template <int OF> __device__ __forceinline__ f32x16 vdk_mfma32(s16x8 a, s16x8 b, f32x16 c) {
#ifdef VDK_EMU
  return emu_mfma_32x32x16_op<OF>(a, b, c);
#else
  if constexpr (OF == VDK_OPF_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vdk_f16x8, a), __builtin_bit_cast(vdk_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
template <int OF> __device__ __forceinline__ f32x4 vdk_mfma16(s16x8 a, s16x8 b, f32x4 c) {
#ifdef VDK_EMU
  return emu_mfma_16x16x32_op<OF>(a, b, c);
#else
  if constexpr (OF == VDK_OPF_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vdk_f16x8, a), __builtin_bit_cast(vdk_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// ---- wave / block reductions -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
// Sum over a whole workgroup of NW waves; `red` is LDS scratch of >= NW floats.
// Result is returned to every thread.  Contains two __syncthreads().
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  __syncthreads();
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  __syncthreads();
  return t;
}

// 2^x on the transcendental unit (v_exp_f32); inputs here are <= ~0 so flushed subnormal results are harmless
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#define VDK_LOG2E 1.4426950408889634f

// erf with |abs err| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): one v_exp_f32, one v_rcp_f32, a 5-term Horner.
// `e` returns exp(-z*z) so that GELU' can reuse it (pdf(x) = exp(-x^2/2)/sqrt(2 pi) with z = x/sqrt(2)).
__device__ __forceinline__ float erf_as(float z, float& e) {
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  e = fast_exp2(-a * a * VDK_LOG2E);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * e;
  return z < 0.f ? -r : r;
}
// The two terms exact-erf GELU (timm nn.GELU default, approximate='none') and its derivative are made of, from the same approximation: q = erfc(|x| / sqrt 2)
// (the tail itself, never 1 - erf: no cancellation for negative x) and e = exp(-x^2 / 2).  Constants folded: exp(-x^2 / 2) = 2^-(s^2) with s = x sqrt(log2(e) / 2), and
// t = 1 / (1 + 0.3275911 |x| / sqrt 2).  x erf(x / sqrt 2) = |x| (1 - q) is even in x, so neither function selects on the sign:
//   GELU(x) = max(x, 0) - |x| q / 2,      GELU'(x) = 1/2 + copysign(1/2 - q/2, x) + x e / sqrt(2 pi).
// Written on PAIRS of values: every step except the two transcendentals (v_exp_f32, v_rcp_f32), |x| (an AND) and max(x, 0) is one packed fp32 instruction for two
// values (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) -- a wave that has its SIMD to itself issues one VALU instruction every ~5 cycles whatever the dependencies,
// so the epilogues that evaluate these cost their instruction COUNT.  The scalar forms below evaluate the same operations in the same order (bit-identical).
typedef float vdk_f32x2 __attribute__((ext_vector_type(2)));
// generic over T = vdk_f32x2 (packed instructions) or float; the per-component pieces (|x| as an AND, v_rcp, v_exp, max, copysign) are overloaded
__device__ __forceinline__ vdk_f32x2 vdk_fma2(vdk_f32x2 a, vdk_f32x2 b, vdk_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float gx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ vdk_f32x2 gx_fma(vdk_f32x2 a, vdk_f32x2 b, vdk_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float gx_abs(float x) { return __uint_as_float(__float_as_uint(x) & 0x7fffffffu); }
__device__ __forceinline__ vdk_f32x2 gx_abs(vdk_f32x2 x) { return (vdk_f32x2){gx_abs(x[0]), gx_abs(x[1])}; }
__device__ __forceinline__ float gx_exp2(float x) { return fast_exp2(x); }
__device__ __forceinline__ vdk_f32x2 gx_exp2(vdk_f32x2 x) { return (vdk_f32x2){fast_exp2(x[0]), fast_exp2(x[1])}; }
__device__ __forceinline__ float gx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ vdk_f32x2 gx_rcp(vdk_f32x2 x) { return (vdk_f32x2){__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1])}; }
__device__ __forceinline__ float gx_max0(float x) { return fmaxf(x, 0.f); }
__device__ __forceinline__ vdk_f32x2 gx_max0(vdk_f32x2 x) { return (vdk_f32x2){fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)}; }
__device__ __forceinline__ float gx_copysign(float m, float s) { return copysignf(m, s); }
__device__ __forceinline__ vdk_f32x2 gx_copysign(vdk_f32x2 m, vdk_f32x2 s) { return (vdk_f32x2){copysignf(m[0], s[0]), copysignf(m[1], s[1])}; }
template <class T> __device__ __forceinline__ T gx_c(float c);
template <> __device__ __forceinline__ float gx_c<float>(float c) { return c; }
template <> __device__ __forceinline__ vdk_f32x2 gx_c<vdk_f32x2>(float c) { return (vdk_f32x2){c, c}; }
template <class T>
__device__ __forceinline__ void gelu_terms_t(T x, T& ax, T& q, T& e) {
  ax = gx_abs(x);
  const T xx = x * x;
  const T ea = xx * gx_c<T>(-0.72134752044448170f);             // -log2(e) / 2
  const T den = gx_fma(ax, gx_c<T>(0.23164188843369479f), gx_c<T>(1.0f));   // 0.3275911 / sqrt 2
  e = gx_exp2(ea);
  const T t = gx_rcp(den);
  T p = gx_fma(t, gx_c<T>(1.061405429f), gx_c<T>(-1.453152027f));
  p = gx_fma(p, t, gx_c<T>(1.421413741f));
  p = gx_fma(p, t, gx_c<T>(-0.284496736f));
  p = gx_fma(p, t, gx_c<T>(0.254829592f));
  q = (p * t) * e;
}
template <class T> __device__ __forceinline__ T gelu_t(T x) {
  T ax, q, e;
  gelu_terms_t(x, ax, q, e);
  return gx_fma(ax * gx_c<T>(-0.5f), q, gx_max0(x));
}
template <class T> __device__ __forceinline__ T gelu_grad_t(T x) {
  T ax, q, e;
  gelu_terms_t(x, ax, q, e);
  const T h = gx_fma(q, gx_c<T>(-0.5f), gx_c<T>(0.5f));
  return gx_fma(x * gx_c<T>(0.3989422804014327f), e, gx_copysign(h, x) + gx_c<T>(0.5f));
}
template <class T> __device__ __forceinline__ void gelu_both_t(T x, T& g, T& d) {
  T ax, q, e;
  gelu_terms_t(x, ax, q, e);
  g = gx_fma(ax * gx_c<T>(-0.5f), q, gx_max0(x));
  const T h = gx_fma(q, gx_c<T>(-0.5f), gx_c<T>(0.5f));
  d = gx_fma(x * gx_c<T>(0.3989422804014327f), e, gx_copysign(h, x) + gx_c<T>(0.5f));
}
// VDK_GELU_SCALAR (A/B builds, with -fno-slp-vectorize): the same arithmetic on single-value instructions -- bit-identical results.  Measured: single-value forms help the
// 256x128 kernel, whose epilogue runs beside the other workgroup's MFMAs (dfc2 x GELU' 324 -> 314 us), and cost the 256x256 kernel (307 -> 317 us); a per-kernel choice
// needs the file compiled without the SLP vectorizer, which put scratch into other variants for 0.04 ms of the step: not adopted (DESIGN_HISTORY.md)
#ifdef VDK_GELU_SCALAR
__device__ __forceinline__ vdk_f32x2 gelu_f2(vdk_f32x2 x) { return (vdk_f32x2){gelu_t<float>(x[0]), gelu_t<float>(x[1])}; }
__device__ __forceinline__ vdk_f32x2 gelu_grad_f2(vdk_f32x2 x) { return (vdk_f32x2){gelu_grad_t<float>(x[0]), gelu_grad_t<float>(x[1])}; }
__device__ __forceinline__ void gelu_both_f2(vdk_f32x2 x, vdk_f32x2& g, vdk_f32x2& d) {
  float g0, g1, d0, d1;
  gelu_both_t<float>(x[0], g0, d0); gelu_both_t<float>(x[1], g1, d1);
  g = (vdk_f32x2){g0, g1}; d = (vdk_f32x2){d0, d1};
}
#else
__device__ __forceinline__ vdk_f32x2 gelu_f2(vdk_f32x2 x) { return gelu_t<vdk_f32x2>(x); }
__device__ __forceinline__ vdk_f32x2 gelu_grad_f2(vdk_f32x2 x) { return gelu_grad_t<vdk_f32x2>(x); }
__device__ __forceinline__ void gelu_both_f2(vdk_f32x2 x, vdk_f32x2& g, vdk_f32x2& d) { gelu_both_t<vdk_f32x2>(x, g, d); }
#endif
__device__ __forceinline__ float gelu_f(float x) { return gelu_f2((vdk_f32x2){x, x})[0]; }
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_grad_f2((vdk_f32x2){x, x})[0]; }

__device__ __forceinline__ void gelu_both_f(float x, float& g, float& d) {
  vdk_f32x2 g2, d2;
  gelu_both_f2((vdk_f32x2){x, x}, g2, d2);
  g = g2[0]; d = d2[0];
}

// monotone float -> uint32 key (larger float -> larger key)
__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

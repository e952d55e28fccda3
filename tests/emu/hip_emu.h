// hip_emu.h — TEST INFRASTRUCTURE ONLY.
//
// A tiny SIMT emulator that lets the HIP kernels under visiondk_amd/csrc/ be compiled by the
// HOST clang++ (-x c++ -include tests/emu/hip_emu.h) and executed on CPU threads, so that the
// index math of every kernel (MFMA fragment layouts, LDS tiling, masks, reductions) can be
// parity-checked against the oracle in the GPU-less build container, through the very same C ABI
// (include/visiondk.h) that the gfx950 library exports.
//
// It is NOT a product path: nothing under visiondk_amd/ loads the emulated library; only
// tests/ build and load it (tests/emu/build_emu.py).  The product library is compiled by hipcc for
// gfx950 from the same sources with no preprocessor switches.
//
// Model: one OS worker thread runs one workgroup at a time; every HIP thread is a ucontext fiber;
// __syncthreads / wave collectives are cooperative barriers between fibers.  A wave is 64
// consecutive threads.  MFMA builtins are emulated with the gfx950 fragment layouts documented in
// /opt/skills/guides/cdna_hip_programming.md §3 (A/B: lane l holds row/col (l & (M-1)) and the 8
// consecutive k starting at 8*(l / M); C/D: col = l & (M-1), row = f(reg, l / M)).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <atomic>
#include <mutex>
#include <algorithm>

#define VDK_EMU 1
#define VDK_PIN2(x, y) ((void)0)
#define VDK_AGENT_ST_U64(p, v) (*(unsigned long long*)(p) = (unsigned long long)(v))
#define VDK_AGENT_LD_U64(p) (*(const unsigned long long*)(p))
#define VDK_AGENT_ST_I32(p, v) (*(int*)(p) = (int)(v))
#define VDK_AGENT_LD_I32(p) (*(const int*)(p))
#define VDK_AGENT_ADD_I32(p, v) __atomic_fetch_add((int*)(p), (int)(v), __ATOMIC_RELAXED)
#define VDK_LDS_PTR(p) ((void*)(p))
#define VDK_LDS_S16X4(p) (p)
#define VDK_GLOBAL_PTR(p) ((const void*)(p))

// ---------------------------------------------------------------- keywords
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };

// ---------------------------------------------------------------- runtime API subset
typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return 0;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static int token; *s = (hipStream_t)&token; return 0; }   // everything runs in program order
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 16; return 0; }   // the emulated "device" has 16 CUs (stream-K grids)
typedef void* hipEvent_t;
#define VDK_EMU_NO_HIP_EXT 1
#define hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, ev0, ev1, flags, ...) hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__)
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4

namespace emu {

struct Barrier { int arrived = 0; unsigned gen = 0; };

struct Wave {
  Barrier bar;
  int alive = 0;
  uint64_t x64[64];
  uint32_t a32[64][8];   // operand exchange (A)
  uint32_t b32[64][8];   // operand exchange (B)
};

struct Block;
// minimal cooperative context: the callee-saved registers live on the fiber's own stack, only the stack pointer is kept here (glibc's swapcontext
// costs a sigprocmask system call per switch, which dominated the emulated run time)
struct Ctx { void* sp = nullptr; };
extern "C" void emu_switch(Ctx* from, Ctx* to);

struct Fiber {
  Ctx ctx;
  Block* blk = nullptr;
  emu_uint3 tid{0, 0, 0};
  int lin = 0;
  bool done = false;
  char* stack = nullptr;
};

struct Block {
  Ctx sched;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  Barrier bar;
  int alive = 0;
  emu_uint3 bid{0, 0, 0};
  dim3 bdim, gdim;
  const std::function<void()>* body = nullptr;
  Fiber* cur = nullptr;
  bool progress = false;
  size_t dyn_lds_bytes = 0;
};

extern thread_local Block* g_blk;
static const size_t kStack = 256 * 1024;

inline Fiber* cur() { return g_blk->cur; }
inline void yield() { Fiber* f = cur(); emu_switch(&f->ctx, &g_blk->sched); }

inline void barrier_wait(Barrier& b, int& alive) {
  unsigned gen = b.gen;
  if (++b.arrived >= alive) { b.arrived = 0; b.gen++; g_blk->progress = true; return; }
  while (b.gen == gen) yield();
}
inline void block_barrier() { barrier_wait(g_blk->bar, g_blk->alive); }
inline Wave& wave() { return g_blk->waves[cur()->lin >> 6]; }
inline int lane() { return cur()->lin & 63; }
inline void wave_barrier() { Wave& w = wave(); barrier_wait(w.bar, w.alive); }

void fiber_entry();
void run_block(Block& b);
void launch(dim3 grid, dim3 block, const std::function<void()>& body, size_t dyn_lds_bytes = 0);
unsigned char* dyn_lds();

}  // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); }, (size_t)(shmem))
// dynamic LDS (`extern __shared__`): one 160 KB arena per worker thread, poisoned with bf16 NaNs before every workgroup so that reads of
// never-written LDS show up in the parity tests instead of passing on stale data
#define VDK_DYN_LDS(name) unsigned char* const name = emu::dyn_lds()
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }

static inline void __syncthreads() { emu::block_barrier(); }

// ---------------------------------------------------------------- cross-lane
template <class T> static inline T emu_xchg(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "xchg");
  emu::Wave& w = emu::wave();
  uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
  w.x64[emu::lane()] = bits;
  emu::wave_barrier();
  uint64_t r = w.x64[src_lane & 63];
  emu::wave_barrier();
  T out; memcpy(&out, &r, sizeof(T));
  return out;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return emu_xchg(v, emu::lane() ^ mask); }
template <class T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return emu_xchg(v, src); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) { (void)width; int s = emu::lane() + d; if (s > 63) s = emu::lane(); return emu_xchg(v, s); }
static inline unsigned long long __ballot(int pred) {
  emu::Wave& w = emu::wave();
  w.x64[emu::lane()] = pred ? 1 : 0;
  emu::wave_barrier();
  unsigned long long m = 0;
  int base = (emu::cur()->lin >> 6) << 6;
  int n = (int)emu::g_blk->fibers.size() - base; if (n > 64) n = 64;
  for (int i = 0; i < n; ++i) if (!emu::g_blk->fibers[base + i].done && w.x64[i]) m |= 1ull << i;
  emu::wave_barrier();
  return m;
}
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) {
  unsigned long long m = __ballot(!p);
  return m == 0;
}
template <class T> static inline T __builtin_amdgcn_readfirstlane_emu(T v) { return emu_xchg(v, 0); }
#define __builtin_amdgcn_readfirstlane(v) __builtin_amdgcn_readfirstlane_emu(v)

// ---------------------------------------------------------------- atomics (global or LDS)
static inline float atomicAdd(float* p, float v) {
  uint32_t* u = (uint32_t*)p; uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
  float f;
  do { memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4); } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4); return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) { int old = __atomic_load_n(p, __ATOMIC_RELAXED); while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return old; }
static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED); while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return old; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---------------------------------------------------------------- bit casts / math
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }

// ---------------------------------------------------------------- MFMA
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef short emu_s16x8 __attribute__((ext_vector_type(8)));

static inline float emu_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
static inline float emu_h2f(unsigned short h) { _Float16 v; memcpy(&v, &h, 2); return (float)v; }
template <int OF> static inline float emu_op2f(unsigned short h) { return OF ? emu_h2f(h) : emu_bf2f(h); }

// v_mfma_f32_32x32x16_bf16: A lane l -> row l&31, k = 8*(l>>5)+e ; B lane l -> col l&31, k = 8*(l>>5)+e
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
template <int OF> static inline emu_f32x16 emu_mfma_32x32x16_op(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c) {
  emu::Wave& w = emu::wave();
  int l = emu::lane();
  memcpy(w.a32[l], &a, 16); memcpy(w.b32[l], &b, 16);
  emu::wave_barrier();
  emu_f32x16 d;
  int j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      const unsigned short* pa = (const unsigned short*)w.a32[i + 32 * (k >> 3)];
      const unsigned short* pb = (const unsigned short*)w.b32[j + 32 * (k >> 3)];
      acc += emu_op2f<OF>(pa[k & 7]) * emu_op2f<OF>(pb[k & 7]);
    }
    d[r] = acc;
  }
  emu::wave_barrier();
  return d;
}
static inline emu_f32x16 emu_mfma_32x32x16_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c, int, int, int) { return emu_mfma_32x32x16_op<0>(a, b, c); }
// v_mfma_f32_16x16x32_bf16: A lane l -> row l&15, k = 8*(l>>4)+e ; D: col = l&15, row = 4*(l>>4)+r
template <int OF> static inline emu_f32x4 emu_mfma_16x16x32_op(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c) {
  emu::Wave& w = emu::wave();
  int l = emu::lane();
  memcpy(w.a32[l], &a, 16); memcpy(w.b32[l], &b, 16);
  emu::wave_barrier();
  emu_f32x4 d;
  int j = l & 15, q = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int i = 4 * q + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      const unsigned short* pa = (const unsigned short*)w.a32[i + 16 * (k >> 3)];
      const unsigned short* pb = (const unsigned short*)w.b32[j + 16 * (k >> 3)];
      acc += emu_op2f<OF>(pa[k & 7]) * emu_op2f<OF>(pb[k & 7]);
    }
    d[r] = acc;
  }
  emu::wave_barrier();
  return d;
}
static inline emu_f32x4 emu_mfma_16x16x32_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c, int, int, int) { return emu_mfma_16x16x32_op<0>(a, b, c); }
// v_mfma_f32_32x32x2_f32: A lane l -> A[l&31][l>>5], B lane l -> B[l>>5][l&31];
// exact f32: D = fma(a_k1, b_k1, fma(a_k0, b_k0, C))  (k-ordered fmaf chain, guide §3)
static inline emu_f32x16 emu_mfma_32x32x2_f32(float a, float b, emu_f32x16 c, int, int, int) {
  emu::Wave& w = emu::wave();
  int l = emu::lane();
  memcpy(&w.a32[l][0], &a, 4); memcpy(&w.b32[l][0], &b, 4);
  emu::wave_barrier();
  emu_f32x16 d;
  int j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float a0, a1, b0, b1;
    memcpy(&a0, &w.a32[i][0], 4); memcpy(&a1, &w.a32[i + 32][0], 4);
    memcpy(&b0, &w.b32[j][0], 4); memcpy(&b1, &w.b32[j + 32][0], 4);
    d[r] = fmaf(a1, b1, fmaf(a0, b0, c[r]));
  }
  emu::wave_barrier();
  return d;
}
// ---- OCP fp8 (e4m3fn: bias 7, no inf, S.1111.111 = NaN; e5m2: IEEE-like, bias 15) ---------------------------------------------------------
static inline float emu_fp8_to_f32(unsigned char v, int fmt) {
  const int s = v >> 7;
  float r;
  if (fmt == 0) {
    const int e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = ldexpf((float)m, -9);             // subnormal: m * 2^-3 * 2^-6
    else r = ldexpf(1.0f + m / 8.0f, e - 7);
  } else {
    const int e = (v >> 2) & 31, m = v & 3;
    if (e == 31) r = m ? NAN : INFINITY;
    else if (e == 0) r = ldexpf((float)m, -16);            // m * 2^-2 * 2^-14
    else r = ldexpf(1.0f + m / 4.0f, e - 15);
  }
  return s ? -r : r;
}
// round to nearest even, saturating to the format's largest finite value (the kernels clamp before converting, so saturation is never exercised differently)
static inline unsigned char emu_f32_to_fp8(float x, int fmt) {
  const int mb = fmt == 0 ? 3 : 2, bias = fmt == 0 ? 7 : 15, emax = fmt == 0 ? 8 : 15;
  const float maxv = fmt == 0 ? 448.0f : 57344.0f;
  unsigned char sgn = std::signbit(x) ? 0x80 : 0;
  if (std::isnan(x)) return sgn | 0x7f;
  float a = fabsf(x);
  if (a > maxv) a = maxv;
  if (a == 0.f) return sgn;
  int e; frexpf(a, &e); e -= 1;                             // a = 1.f * 2^e
  if (e < 1 - bias) e = 1 - bias;                           // subnormal range: fixed exponent
  const float q = ldexpf(1.0f, e - mb);                     // spacing
  float r = nearbyintf(a / q) * q;                          // RNE (default rounding mode)
  if (r > maxv) r = maxv;
  int e2; frexpf(r, &e2); e2 -= 1;
  unsigned char bits;
  if (r < ldexpf(1.0f, 1 - bias)) bits = (unsigned char)lrintf(r / ldexpf(1.0f, 1 - bias - mb));     // subnormal mantissa
  else bits = (unsigned char)(((e2 + bias) << mb) | ((int)lrintf((r / ldexpf(1.0f, e2) - 1.0f) * (1 << mb)) & ((1 << mb) - 1)));
  (void)emax;
  return sgn | bits;
}
static inline int emu_cvt_pk_f8(float a, float b, int old, bool hi_word, int fmt) {
  const unsigned pk = (unsigned)emu_f32_to_fp8(a, fmt) | ((unsigned)emu_f32_to_fp8(b, fmt) << 8);
  const unsigned o = (unsigned)old;
  return (int)(hi_word ? ((o & 0x0000ffffu) | (pk << 16)) : ((o & 0xffff0000u) | pk));
}
#define __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, sel) emu_cvt_pk_f8((a), (b), (old), (sel), 0)
#define __builtin_amdgcn_cvt_pk_bf8_f32(a, b, old, sel) emu_cvt_pk_f8((a), (b), (old), (sel), 1)
// v_mfma_scale_f32_32x32x64_f8f6f4 with 8-bit formats: A lane l -> row l & 31, k = 32 * (l >> 5) + byte; B likewise with col; D as the other 32x32 shapes.
// scale_a / scale_b: E8M0 bytes (127 = 2^0)
typedef int emu_i32x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 emu_mfma_scale_32x32x64_f8(emu_i32x8 a, emu_i32x8 b, emu_f32x16 c, int fa, int fb, int, int sa, int, int sb) {
  emu::Wave& w = emu::wave();
  int l = emu::lane();
  memcpy(w.a32[l], &a, 32); memcpy(w.b32[l], &b, 32);
  emu::wave_barrier();
  emu_f32x16 d;
  const float sc = ldexpf(1.0f, (sa & 255) - 127) * ldexpf(1.0f, (sb & 255) - 127);
  int j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) {
      const unsigned char* pa = (const unsigned char*)w.a32[i + 32 * (k >> 5)];
      const unsigned char* pb = (const unsigned char*)w.b32[j + 32 * (k >> 5)];
      acc += emu_fp8_to_f32(pa[k & 31], fa) * emu_fp8_to_f32(pb[k & 31], fb);
    }
    d[r] = c[r] + acc * sc;
  }
  emu::wave_barrier();
  return d;
}
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4 emu_mfma_scale_32x32x64_f8
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu_mfma_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2_f32
// LDS-DMA: data of lane i lands at (wave-uniform LDS base taken from the first lane) + i*size
static inline void emu_global_load_lds(const void* gsrc, void* lds_dst, unsigned size, int offset, unsigned) {
  unsigned long long base = emu_xchg((unsigned long long)(uintptr_t)lds_dst, 0);
  memcpy((char*)(uintptr_t)base + offset + (size_t)emu::lane() * size, gsrc, size);
  emu::wave_barrier();
}
#define __builtin_amdgcn_global_load_lds(g, l, sz, off, aux) emu_global_load_lds((const void*)(g), (void*)(l), sz, off, aux)
// buffer_load ... lds through a raw buffer descriptor: byte offset = voffset + soffset + imm; accesses that do not fit below num_records return zeros
struct emu_buffer_rsrc { const unsigned char* base; unsigned num_records; };
typedef emu_buffer_rsrc __amdgpu_buffer_rsrc_t;
static inline emu_buffer_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num_records, int) { return emu_buffer_rsrc{(const unsigned char*)p, (unsigned)num_records}; }
static inline void emu_buffer_load_lds(emu_buffer_rsrc rs, void* lds_dst, unsigned size, unsigned voff, unsigned soff, int ioff, int) {
  unsigned long long base = emu_xchg((unsigned long long)(uintptr_t)lds_dst, 0);
  const unsigned long long off = (unsigned long long)voff + soff + (unsigned)ioff;
  char* dst = (char*)(uintptr_t)base + ioff + (size_t)emu::lane() * size;
  if (off + size <= rs.num_records) memcpy(dst, rs.base + off, size); else memset(dst, 0, size);
  emu::wave_barrier();
}
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
static inline void emu_buffer_store_b128(emu_u32x4 v, emu_buffer_rsrc rs, unsigned voff, unsigned soff, int) {
  const unsigned long long off = (unsigned long long)voff + soff;
  if (off + 16 <= rs.num_records) memcpy((unsigned char*)rs.base + off, &v, 16);
}
static inline emu_u32x4 emu_buffer_load_b128(emu_buffer_rsrc rs, unsigned voff, unsigned soff, int) {
  const unsigned long long off = (unsigned long long)voff + soff;
  emu_u32x4 v = {0u, 0u, 0u, 0u};
  if (off + 16 <= rs.num_records) memcpy(&v, rs.base + off, 16);
  return v;
}
#define __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, aux) emu_buffer_store_b128((v), (rs), (unsigned)(voff), (unsigned)(soff), (aux))
#define __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux) emu_buffer_load_b128((rs), (unsigned)(voff), (unsigned)(soff), (aux))
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, sz, voff, soff, ioff, aux) emu_buffer_load_lds((rs), (void*)(l), sz, voff, soff, ioff, aux)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
// ds_read_b64_tr_b16, semantics measured on gfx950 (tools/probes/tr_probe.hip): within each 16-lane group, lane i's
// element j = element (i % 4) of the 8-byte chunk addressed by lane 4*j + i/4 of the same group.
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
static inline emu_s16x4 emu_ds_read_tr16_b64(const void* p) {
  emu::Wave& w = emu::wave();
  int l = emu::lane();
  w.x64[l] = (uint64_t)(uintptr_t)p;
  emu::wave_barrier();
  emu_s16x4 out;
  int g = l >> 4, i = l & 15;
  for (int j = 0; j < 4; ++j) {
    const short* src = (const short*)(uintptr_t)w.x64[g * 16 + 4 * j + (i >> 2)];
    out[j] = src[i & 3];
  }
  emu::wave_barrier();
  return out;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void*)(p))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define VDK_WAVE_LDS_SYNC() emu::wave_barrier()
#define VDK_LDS_ADD_F32(p, v) (*(p) += (v))
#define VDK_READLANE(v, l) __shfl((int)(v), (l))
#define __builtin_readcyclecounter() 0ull
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_alignbit(hi_, lo_, sh_) ((unsigned)(((((unsigned long long)(unsigned)(hi_)) << 32) | (unsigned)(lo_)) >> ((sh_) & 31)))   /* v_alignbit_b32 */
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_barrier() __syncthreads()
static inline int __mul24(int a, int b) { return a * b; }

// vdk_attn_tile.h — LDS tile helpers shared by the LDS-resident attention kernels (attention_small.hip, attention_long.hip): the swizzled row layout, LDS-DMA row loads,
// row / transposed MFMA fragments with lane-only offsets, C-layout -> B-operand packing and the coalesced tile store.
#pragma once
#include <hip/hip_runtime.h>
#include "vdk_device.h"

#define AS_ROW 128   // bytes per staged row: 64 bf16

__device__ __forceinline__ int as_f(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// rows [0, R8) of a [N, 64] bf16 operand (row stride ld elements) -> arr, 8 rows per wave-instruction, rows >= N read row N-1 (finite filler)
__device__ __forceinline__ void as_dma_rows(unsigned char* arr, const bf16_t* __restrict__ src, long ld, int N, int R8, int w, int nw, int lane) {
  for (int j = w; j < (R8 >> 3); j += nw) {
    const int row = 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ as_f(row);
    const int srow = row < N ? row : N - 1;
    const bf16_t* g = src + (long)srow * ld + c * 8;
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(g), VDK_LDS_PTR(arr + j * 1024), 16, 0, 0);
  }
}
// The same rows by an LDS-DMA the compiler does not see (inline asm, cdna_hip_programming.md section 5.7).  With the builtin form hipcc keeps the DMA on its own vmcnt
// scoreboard: it waits vmcnt(0) in front of the next ds_read_b64_tr_b16 and in front of every __syncthreads(), i.e. a tile requested two tiles ahead was waited for inside
// the tile that requested it (one exposed memory round trip per query tile of the one-pass backward).  The caller counts completion by hand (s_waitcnt vmcnt(n)) and makes
// the rows visible to the other waves with a barrier behind that wait.
__device__ __forceinline__ void as_dma16_raw(const bf16_t* g, unsigned char* lds_dst) {
#ifdef VDK_EMU
  __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(g), VDK_LDS_PTR(lds_dst), 16, 0, 0);
#else
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)VDK_LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory");
#endif
}
__device__ __forceinline__ void as_dma_rows_raw(unsigned char* arr, const bf16_t* __restrict__ src, long ld, int N, int R8, int w, int nw, int lane) {
  for (int j = w; j < (R8 >> 3); j += nw) {
    const int row = 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ as_f(row);
    const int srow = row < N ? row : N - 1;
    as_dma16_raw(src + (long)srow * ld + c * 8, arr + j * 1024);
  }
}
// MFMA A/B fragment of a row-major staged tile: lane (row, hi) -> the 16 bytes at k = 16*ks + 8*hi
__device__ __forceinline__ s16x8 as_row_frag(const unsigned char* arr, int row, int ks, int hi) {
  return *(const s16x8*)(arr + row * AS_ROW + (((2 * ks + hi) ^ as_f(row)) << 4));
}
// transposed fragment: lane (column d0 + (lane & 31), hi) gets rows t1 + 4*hi + {0..3} in slots 0..3 and the same + 8 in slots 4..7 (the permuted
// contraction order that makes an MFMA C-layout tile directly usable as the other operand, see attention.hip)
__device__ __forceinline__ s16x8 as_tr_frag(const unsigned char* arr, int t1, int d0, int lane) {
  const int s = lane & 15, chalf = (lane >> 4) & 1, hi = lane >> 5;
  const int r1 = t1 + 4 * hi + (s >> 2), r2 = r1 + 8;
  const int byte = 2 * d0 + 32 * chalf + 8 * (s & 3);
  const unsigned char* p1 = arr + r1 * AS_ROW + (((byte >> 4) ^ as_f(r1)) << 4) + (byte & 8);
  const unsigned char* p2 = arr + r2 * AS_ROW + (((byte >> 4) ^ as_f(r2)) << 4) + (byte & 8);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p1));
  s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(p2));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return r;
}
// Lane-only parts of the two fragment addresses.  as_f() looks at bits 1..3 of the row, so for a tile that starts at a multiple of 32 (row fragments) / 16 (transposed
// fragments) the swizzle depends on the lane alone: address = array + tile_row0 * 128 + lane offset.  With the tile loops fully unrolled the middle term is an immediate of
// the ds_read, and the ~20 VALU instructions per fragment address (266 VALU instructions per 16 MFMAs in the backward: PMC SQ_INSTS_VALU, 52 % VALU-active against 23 %
// MFMA-busy) drop out of the inner loops.
struct AsLane { int row[4]; int tr[2][2]; };
__device__ __forceinline__ AsLane as_lane(int lane) {
  AsLane a;
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) a.row[ks] = l31 * AS_ROW + (((2 * ks + hi) ^ as_f(l31)) << 4);
  const int s = lane & 15, chalf = (lane >> 4) & 1;
  const int r1 = 4 * hi + (s >> 2), r2 = r1 + 8;
#pragma unroll
  for (int dh = 0; dh < 2; ++dh) {
    const int byte = 64 * dh + 32 * chalf + 8 * (s & 3);
    a.tr[dh][0] = r1 * AS_ROW + (((byte >> 4) ^ as_f(r1)) << 4) + (byte & 8);
    a.tr[dh][1] = r2 * AS_ROW + (((byte >> 4) ^ as_f(r2)) << 4) + (byte & 8);
  }
  return a;
}
// row fragment of the 32-row tile at `tile` (= array + tile_row0 * AS_ROW)
__device__ __forceinline__ s16x8 as_row_frag_l(const unsigned char* tile, const AsLane& a, int ks) { return *(const s16x8*)(tile + a.row[ks]); }
// transposed fragment of the 16 rows at `tile`, column half dh (d0 = 32 * dh)
__device__ __forceinline__ s16x8 as_tr_frag_l(const unsigned char* tile, const AsLane& a, int dh) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(tile + a.tr[dh][0]));
  s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(VDK_LDS_S16X4(tile + a.tr[dh][1]));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return r;
}
__device__ __forceinline__ f32x16 as_zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// 16 C-layout values -> the two B-operand fragments (k-slot j of step s <-> accumulator register 8*s + j)
template <int OF = 0>
__device__ __forceinline__ void as_pack_b(const f32x16& p, s16x8 (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 u = {pack_op2<OF>(p[8 * s + 0], p[8 * s + 1]), pack_op2<OF>(p[8 * s + 2], p[8 * s + 3]), pack_op2<OF>(p[8 * s + 4], p[8 * s + 5]), pack_op2<OF>(p[8 * s + 6], p[8 * s + 7])};
    f[s] = *(s16x8*)&u;
  }
}
// a wave's 32 x 64 bf16 output tile (C-layout of a transposed product: lane = row, registers = 4 consecutive columns per group) -> its private 4 KB
// LDS tile (16-byte chunk ^ (row & 7)) -> 128-byte coalesced global rows
template <int OF = 0>
__device__ __forceinline__ void as_store_tile(unsigned char* tile, const f32x16& x0, const f32x16& x1, float mul, bf16_t* __restrict__ dst, long ld, int row0, int N,
                                              int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  if (row0 + l31 < N) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = g ^ (l31 & 7), ch1 = (4 + g) ^ (l31 & 7);
      *(u32x2*)(tile + l31 * AS_ROW + (ch << 4) + 8 * hi) = (u32x2){pack_op2<OF>(x0[4 * g] * mul, x0[4 * g + 1] * mul), pack_op2<OF>(x0[4 * g + 2] * mul, x0[4 * g + 3] * mul)};
      *(u32x2*)(tile + l31 * AS_ROW + (ch1 << 4) + 8 * hi) = (u32x2){pack_op2<OF>(x1[4 * g] * mul, x1[4 * g + 1] * mul), pack_op2<OF>(x1[4 * g + 2] * mul, x1[4 * g + 3] * mul)};
    }
  }
  VDK_WAVE_LDS_SYNC();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = 8 * p + (lane >> 3), cp = lane & 7;
    if (row0 + r < N) {
      const u32x4 v = *(const u32x4*)(tile + r * AS_ROW + ((cp ^ (r & 7)) << 4));
      *(u32x4*)(dst + (long)(row0 + r) * ld + cp * 8) = v;
    }
  }
  VDK_WAVE_LDS_SYNC();
}

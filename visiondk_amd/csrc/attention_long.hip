// attention_long.hip — K3 forward and backward for sequences beyond the LDS-resident range (N > 256 keys: ViT-L/14 at 336 has N = 576 / 577, ViT-L/14 at 224 N = 257).
// Same operator as attention.hip / attention_small.hip (timm `Attention`: softmax(q k^T / sqrt(hd)) v behind models/classifier/classify_model.py:49-54 and
// models/faceX/backbone/timm_wrapper.py:16-21).  The flash-style forward of attention.hip stages K / V chunks through REGISTERS (270 VGPRs = one wave per SIMD, a
// load -> barrier -> compute sequence per chunk): 1037 us per layer at B*H = 2048, N = 576 = 168 TFLOP/s, against 381 TFLOP/s of the short-sequence kernel on its shapes.
// This kernel keeps the short kernel's machinery and streams the keys:
//   * a work unit is (batch, head, group of 4 query tiles): wave w owns query tile 4g + w, Q fragments straight from global, O^T (lane = query) in registers;
//   * K / V arrive in chunks of 96 rows by LDS-DMA into a DOUBLE buffer (2 x 24 KB): the chunk after the current one -- of this unit or of the workgroup's next unit -- is
//     requested right after the single barrier of a chunk, so it travels during the MFMAs; 64 KB of LDS per workgroup = two workgroups per CU;
//   * online softmax per chunk with the deferred rescale (the running maximum only moves when a chunk exceeds it by 2^8), P rounded to bf16 unnormalised, O scaled by 1 / l at
//     the end -- the rounding points of attention.hip's forward, whose parity tests this kernel shares;
//   * units are numbered so that the groups of one (batch, head) item run at the same time on ONE XCD (workgroup b lives on XCD b mod 8): its K / V rows come from HBM once
//     and from that XCD's L2 for the other groups.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "vdk_device.h"
#include "vdk_host.h"
#include "vdk_attn_tile.h"

#define AL_CT 3                            // key tiles per chunk
#define AL_CROWS (32 * AL_CT)              // 96 rows: a multiple of 16, so the row swizzle of a chunk-local row equals the one of the global row
#define AL_ARR (AL_CROWS * AS_ROW)         // one operand of one chunk: 12 KB
#define AL_BUF (2 * AL_ARR)                // K rows | V rows

// rows [r0, r0 + 96) of two [N, 64] operands (row strides lda / ldb) -> buf, rows >= N read row N-1 (finite filler; its contribution is masked)
__device__ __forceinline__ void al_dma_chunk2(unsigned char* buf, const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b, long ldb, int N, int r0, int w, int lane) {
#pragma unroll
  for (int j0 = 0; j0 < AL_CROWS / 32; ++j0) {
    const int j = w + 4 * j0;
    const int lrow = 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ as_f(lrow);
    int srow = r0 + lrow;
    srow = srow < N ? srow : N - 1;
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(a + (long)srow * lda + c * 8), VDK_LDS_PTR(buf + j * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(b + (long)srow * ldb + c * 8), VDK_LDS_PTR(buf + AL_ARR + j * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ void al_dma_chunk(unsigned char* buf, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld, int N, int r0, int w, int lane) {
  al_dma_chunk2(buf, k, ld, v, ld, N, r0, w, lane);
}

// LDS (dynamic): chunk buffer 0 | chunk buffer 1 | 4 wave store tiles of 4 KB
template <int OF>
__global__ __launch_bounds__(256, 2) void attn_l_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                            bf16_t* __restrict__ o, long ldo, float* __restrict__ lse, int N, int H, float scale, int nitems, int G) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const Wt = smem + 2 * AL_BUF + w * 4096;
  const int nt = (N + 31) >> 5, nch = (N + AL_CROWS - 1) / AL_CROWS;
  const float scale2 = scale * VDK_LOG2E;
  const AsLane al = as_lane(lane);
  // unit t of XCD x: item (t / G) * 8 + x, query-tile group t % G; this workgroup walks t = blockIdx / 8, + gridDim / 8, ...
  const int x = blockIdx.x & 7, tstride = gridDim.x >> 3;
  int t = blockIdx.x >> 3;
  int item = (t / G) * 8 + x;
  int cc = 0;                                                        // chunks consumed so far: buffer parity runs on across units
  if (item < nitems) {
    const long off0 = (long)(item / H) * N * ld + (item % H) * 64;
    al_dma_chunk(smem, k + off0, v + off0, ld, N, 0, w, lane);
  }
  while (item < nitems) {
    const int g = t % G;
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64;
    const int qt = 4 * g + w;
    const bool active = qt < nt;                                     // (wave-uniform) the last group of an item may be short; idle waves still load and meet the barriers
    const int qrow = qt * 32 + l31;
    const int qr = qrow < N ? qrow : N - 1;
    s16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const s16x8*)(q + off + (long)qr * ld + ks * 16 + hi * 8);
    const int tn = t + tstride;
    const int itemn = (tn / G) * 8 + x;
    const long offn = (long)(itemn / H) * N * ld + (itemn % H) * 64;
    float m = -INFINITY, l = 0.f;                                    // running maximum (log2 domain, scaled) and this half-wave's part of the row sum
    f32x16 o0 = as_zero16(), o1 = as_zero16();
    for (int c = 0; c < nch; ++c, ++cc) {
      __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): this wave's part of chunk cc has landed (and its Q fragments)
      __syncthreads();                                               // everybody's part has; everybody is done with chunk cc - 1, whose buffer the next request overwrites
      unsigned char* const nb = smem + ((cc + 1) & 1) * AL_BUF;
      if (c + 1 < nch) al_dma_chunk(nb, k + off, v + off, ld, N, (c + 1) * AL_CROWS, w, lane);
      else if (itemn < nitems) al_dma_chunk(nb, k + offn, v + offn, ld, N, 0, w, lane);
      if (!active) continue;
      const unsigned char* const Kb = smem + (cc & 1) * AL_BUF;
      const unsigned char* const Vb = Kb + AL_ARR;
      const int key0 = c * AL_CROWS;
      const int nv = (N - key0 + 31) >> 5;                            // key tiles of this chunk that hold a valid key (wave-uniform; >= AL_CT except in the last chunk)
      f32x16 st[AL_CT];
#pragma unroll
      for (int kt = 0; kt < AL_CT; ++kt) {
        st[kt] = as_zero16();
        if (kt < nv) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) st[kt] = vdk_mfma32<OF>(as_row_frag_l(Kb + kt * 32 * AS_ROW, al, ks), qf[ks], st[kt]);
        }
      }
      if (key0 + AL_CROWS > N) {                                     // the last chunk holds keys beyond N (wave-uniform)
#pragma unroll
        for (int kt = 0; kt < AL_CT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) st[kt][r] = -INFINITY;
      }
      float mt = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < AL_CT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[kt][r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32));                            // the two half-waves hold the same queries, different keys
      const float mt2 = mt * scale2;
      // deferred rescale: keep the old maximum while the chunk exceeds it by < 2^8; P is then bounded by 2^8 instead of 1 (harmless in bf16 / fp32).  The first chunk always
      // takes the branch (m = -inf): alpha = 0 on zero accumulators.  Every chunk-0 row has a valid key, so the new maximum is finite.
      if (__any(mt2 > m + 8.0f)) {
        const float mn = fmaxf(m, mt2);
        const float alpha = fast_exp2(m - mn);
        l *= alpha;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
#pragma unroll
      for (int kt = 0; kt < AL_CT; ++kt) {
        if (kt >= nv) break;                                         // nothing but masked keys: P = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float p = fast_exp2(fmaf(st[kt][r], scale2, -m)); st[kt][r] = p; l += p; }
        s16x8 pf[2];
        as_pack_b<OF>(st[kt], pf);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          o0 = vdk_mfma32<OF>(as_tr_frag_l(Vb + (kt * 32 + 16 * s) * AS_ROW, al, 0), pf[s], o0);
          o1 = vdk_mfma32<OF>(as_tr_frag_l(Vb + (kt * 32 + 16 * s) * AS_ROW, al, 1), pf[s], o1);
        }
      }
    }
    if (active) {
      l += __shfl_xor(l, 32);
      as_store_tile<OF>(Wt, o0, o1, 1.0f / l, o + (long)b * N * ldo + h * 64, ldo, qt * 32, N, lane);
      if (lse && hi == 0 && qrow < N) lse[((long)b * H + h) * N + qrow] = (m + log2f(l)) * 0.6931471805599453f;
    }
    t = tn;
    item = itemn;
  }
}

// =====================================================================================  backward
// The recompute form of attention_small.hip (dQ by query-tile owner, dK / dV by key-tile owner: 28 instead of 20 MFMAs per tile pair, no shared accumulator, no atomics, fixed
// summation order) on the forward's streaming structure.  The q kernel runs first: it also computes D = rowsum(dO * O) for its query tile (it holds the dO fragments) and
// writes it to `dvec` for the kv kernel.
//   q kernel:  unit = (b, head, group of 4 query tiles); Q / dO / O fragments from global, K / V chunks through the double buffer, dQ^T (lane = query) in registers.
//   kv kernel: unit = (b, head, group of 4 key tiles); K / V fragments from global, Q / dO chunks through the double buffer together with the chunk's 96 lse and D values
//              (staged through one register per thread: loaded under the previous chunk, written to LDS before the chunk's barrier); dK^T, dV^T (lane = key) in registers.
template <int OF>
__global__ __launch_bounds__(256, 2) void attn_l_bwd_q_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                              const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                              float* __restrict__ dvec, bf16_t* __restrict__ dq, long ldd, int N, int H, float scale, int nitems, int G) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const Wt = smem + 2 * AL_BUF + w * 4096;
  const int nt = (N + 31) >> 5, nch = (N + AL_CROWS - 1) / AL_CROWS;
  const float scale2 = scale * VDK_LOG2E;
  const AsLane al = as_lane(lane);
  const int x = blockIdx.x & 7, tstride = gridDim.x >> 3;
  int t = blockIdx.x >> 3;
  int item = (t / G) * 8 + x;
  int cc = 0;
  if (item < nitems) {
    const long off0 = (long)(item / H) * N * ld + (item % H) * 64;
    al_dma_chunk(smem, k + off0, v + off0, ld, N, 0, w, lane);
  }
  while (item < nitems) {
    const int g = t % G;
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    const int qt = 4 * g + w;
    const bool active = qt < nt;
    const int qrow = qt * 32 + l31;
    const int qr = qrow < N ? qrow : N - 1;
    s16x8 qf[4], gf[4];
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const s16x8*)(q + off + (long)qr * ld + ks * 16 + hi * 8);
      gf[ks] = *(const s16x8*)(dout + offo + (long)qr * ldo + ks * 16 + hi * 8);
      const u32x4 of = *(const u32x4*)(o + offo + (long)qr * ldo + ks * 16 + hi * 8);
      const u32x4 gu = *(const u32x4*)&gf[ks];
#pragma unroll
      for (int e = 0; e < 4; ++e) { dsum = fmaf(op_lo<OF>(gu[e]), op_lo<OF>(of[e]), dsum); dsum = fmaf(op_hi<OF>(gu[e]), op_hi<OF>(of[e]), dsum); }
    }
    dsum += __shfl_xor(dsum, 32);                                    // the two half-waves hold the two halves of a query's 64 channels
    const float lq = lse[((long)b * H + h) * N + qr] * VDK_LOG2E;
    if (active && hi == 0 && qrow < N) dvec[((long)b * H + h) * N + qrow] = dsum;
    const int tn = t + tstride;
    const int itemn = (tn / G) * 8 + x;
    const long offn = (long)(itemn / H) * N * ld + (itemn % H) * 64;
    f32x16 gq0 = as_zero16(), gq1 = as_zero16();
    for (int c = 0; c < nch; ++c, ++cc) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      unsigned char* const nb = smem + ((cc + 1) & 1) * AL_BUF;
      if (c + 1 < nch) al_dma_chunk(nb, k + off, v + off, ld, N, (c + 1) * AL_CROWS, w, lane);
      else if (itemn < nitems) al_dma_chunk(nb, k + offn, v + offn, ld, N, 0, w, lane);
      if (!active) continue;
      const unsigned char* const Kb = smem + (cc & 1) * AL_BUF;
      const unsigned char* const Vb = Kb + AL_ARR;
      const int key0 = c * AL_CROWS;
      const bool edge = key0 + AL_CROWS > N;                         // keys beyond N in this chunk (wave-uniform); a lane (= query) beyond N only spoils its own, unstored column
#pragma unroll
      for (int kt = 0; kt < AL_CT; ++kt) {
        if (key0 + kt * 32 >= N) break;                              // a tile of nothing but keys beyond N (wave-uniform)
        f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = vdk_mfma32<OF>(as_row_frag_l(Kb + kt * 32 * AS_ROW, al, ks), qf[ks], st);   // S^T[key][q]: lane = query, registers = keys
          dp = vdk_mfma32<OF>(as_row_frag_l(Vb + kt * 32 * AS_ROW, al, ks), gf[ks], dp);   // dP^T[key][q]
        }
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = fast_exp2(fmaf(st[r], scale2, -lq));
          if (edge && key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) p = 0.f;
          ds[r] = p * (dp[r] - dsum);
        }
        s16x8 df[2];
        as_pack_b<OF>(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gq0 = vdk_mfma32<OF>(as_tr_frag_l(Kb + (kt * 32 + 16 * s2) * AS_ROW, al, 0), df[s2], gq0);   // dQ^T[d][q] += K^T dS^T
          gq1 = vdk_mfma32<OF>(as_tr_frag_l(Kb + (kt * 32 + 16 * s2) * AS_ROW, al, 1), df[s2], gq1);
        }
      }
    }
    if (active) as_store_tile<OF>(Wt, gq0, gq1, scale, dq + (long)b * N * ldd + h * 64, ldd, qt * 32, N, lane);
    t = tn;
    item = itemn;
  }
}

#define AL_BUFKV (AL_BUF + 1024)           // Q rows | dO rows | 128 lse values | 128 D values
template <int OF>
__global__ __launch_bounds__(256, 2) void attn_l_bwd_kv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld,
                                                               const bf16_t* __restrict__ dout, long ldo, const float* __restrict__ lse, const float* __restrict__ dvec,
                                                               bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long ldd, int N, int H, float scale, int nitems, int G) {
  VDK_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const Wt = smem + 2 * AL_BUFKV + w * 4096;
  const int nt = (N + 31) >> 5, nch = (N + AL_CROWS - 1) / AL_CROWS;
  const float scale2 = scale * VDK_LOG2E;
  const AsLane al = as_lane(lane);
  const int x = blockIdx.x & 7, tstride = gridDim.x >> 3;
  int t = blockIdx.x >> 3;
  int item = (t / G) * 8 + x;
  int cc = 0;
  // thread tid < 128 stages lse value tid of a chunk, thread 128 + i the D value i (96 of each are used)
  const int srow_l = tid & 127;
  float staged = 0.f;
  auto stage_load = [&](int it, int r0) {
    int row = r0 + srow_l;
    row = row < N ? row : N - 1;
    const float* src = tid < 128 ? lse : dvec;
    staged = src[(long)it * N + row];
  };
  if (item < nitems) {
    const long off0 = (long)(item / H) * N * ld + (item % H) * 64, offo0 = (long)(item / H) * N * ldo + (item % H) * 64;
    al_dma_chunk2(smem, q + off0, ld, dout + offo0, ldo, N, 0, w, lane);
    stage_load(item, 0);
  }
  while (item < nitems) {
    const int g = t % G;
    const int b = item / H, h = item - b * H;
    const long off = (long)b * N * ld + h * 64, offo = (long)b * N * ldo + h * 64;
    const int kt = 4 * g + w;
    const bool active = kt < nt;
    const int krow = kt * 32 + l31;
    const long kr = (long)(krow < N ? krow : N - 1) * ld;
    s16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const s16x8*)(k + off + kr + ks * 16 + hi * 8); vf[ks] = *(const s16x8*)(v + off + kr + ks * 16 + hi * 8); }
    const int tn = t + tstride;
    const int itemn = (tn / G) * 8 + x;
    const long offn = (long)(itemn / H) * N * ld + (itemn % H) * 64, offon = (long)(itemn / H) * N * ldo + (itemn % H) * 64;
    f32x16 gk0 = as_zero16(), gk1 = as_zero16(), gv0 = as_zero16(), gv1 = as_zero16();
    for (int c = 0; c < nch; ++c, ++cc) {
      __builtin_amdgcn_s_waitcnt(0x0F70);                            // this wave's part of chunk cc has landed, and so has its staged lse / D value
      ((float*)(smem + (cc & 1) * AL_BUFKV + AL_BUF))[tid] = staged;  // (buffer cc & 1 was last read in chunk cc - 2: everybody is past that since the previous barrier)
      __syncthreads();
      unsigned char* const nb = smem + ((cc + 1) & 1) * AL_BUFKV;
      if (c + 1 < nch) { al_dma_chunk2(nb, q + off, ld, dout + offo, ldo, N, (c + 1) * AL_CROWS, w, lane); stage_load(item, (c + 1) * AL_CROWS); }
      else if (itemn < nitems) { al_dma_chunk2(nb, q + offn, ld, dout + offon, ldo, N, 0, w, lane); stage_load(itemn, 0); }
      if (!active) continue;
      const unsigned char* const Qb = smem + (cc & 1) * AL_BUFKV;
      const unsigned char* const Ob = Qb + AL_ARR;
      const float* const lseb = (const float*)(Qb + AL_BUF);
      const float* const Db = lseb + 128;
      const int q0c = c * AL_CROWS;
      const bool edge = q0c + AL_CROWS > N;                          // query rows beyond N in this chunk must be silenced (they would add into valid sums); wave-uniform
#pragma unroll
      for (int qt = 0; qt < AL_CT; ++qt) {
        if (q0c + qt * 32 >= N) break;                               // a tile of nothing but query rows beyond N (wave-uniform)
        f32x16 st = as_zero16(), dp = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = vdk_mfma32<OF>(as_row_frag_l(Qb + qt * 32 * AS_ROW, al, ks), kf[ks], st);   // S[q][key]: lane = key, registers = queries
          dp = vdk_mfma32<OF>(as_row_frag_l(Ob + qt * 32 * AS_ROW, al, ks), vf[ks], dp);   // dP[q][key]
        }
        f32x16 pv, ds;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 lv = *(const f32x4*)(lseb + qt * 32 + 8 * g4 + 4 * hi);
          const f32x4 dd = *(const f32x4*)(Db + qt * 32 + 8 * g4 + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g4 + e;
            float p = fast_exp2(fmaf(st[r], scale2, -lv[e] * VDK_LOG2E));
            if (edge && q0c + qt * 32 + 8 * g4 + 4 * hi + e >= N) p = 0.f;
            pv[r] = p;
            ds[r] = p * (dp[r] - dd[e]);
          }
        }
        s16x8 pf[2], df[2];
        as_pack_b<OF>(pv, pf);
        as_pack_b<OF>(ds, df);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gv0 = vdk_mfma32<OF>(as_tr_frag_l(Ob + (qt * 32 + 16 * s2) * AS_ROW, al, 0), pf[s2], gv0);     // dV^T[d][key] += dO^T P
          gv1 = vdk_mfma32<OF>(as_tr_frag_l(Ob + (qt * 32 + 16 * s2) * AS_ROW, al, 1), pf[s2], gv1);
          gk0 = vdk_mfma32<OF>(as_tr_frag_l(Qb + (qt * 32 + 16 * s2) * AS_ROW, al, 0), df[s2], gk0);     // dK^T[d][key] += Q^T dS
          gk1 = vdk_mfma32<OF>(as_tr_frag_l(Qb + (qt * 32 + 16 * s2) * AS_ROW, al, 1), df[s2], gk1);
        }
      }
    }
    if (active) {
      as_store_tile<OF>(Wt, gk0, gk1, scale, dk + (long)b * N * ldd + h * 64, ldd, kt * 32, N, lane);
      as_store_tile<OF>(Wt, gv0, gv1, 1.0f, dv + (long)b * N * ldd + h * 64, ldd, kt * 32, N, lane);
    }
    t = tn;
    item = itemn;
  }
}

static int al_grid(long units) {
  int cap = 512;                                                     // two workgroups per CU
  if (const char* e = getenv("VDK_ATTN_GRID")) { const int v = atoi(e); if (v > 0) cap = v; }   // tests: force several units per workgroup
  long grid = units < cap ? units : cap;
  return (int)((grid + 7) / 8 * 8);                                  // every XCD residue must be present: items are dealt to XCDs by item mod 8
}

// dqkv: bf16 [B, N, 3, H, 64]; dvec: f32 scratch [B, H, N].  opf: the tensors' 16-bit format (VDK_OPF_BF16 | VDK_OPF_F16)
template <int OF>
static int al_launch_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv, int64_t ldd, float* dvec, int32_t B, int32_t N,
                         int32_t H, float scale, void* stream) {
  const bf16_t* base = (const bf16_t*)qkv;
  bf16_t* dbase = (bf16_t*)dqkv;
  const long D = (long)H * 64;
  const int nt = (N + 31) / 32, G = (nt + 3) / 4;
  const int grid = al_grid((long)B * H * G);
  const size_t lds_q = 2 * AL_BUF + 4 * 4096, lds_kv = 2 * AL_BUFKV + 4 * 4096;
  if (hipFuncSetAttribute((const void*)attn_l_bwd_q_kernel<OF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q) != hipSuccess ||
      hipFuncSetAttribute((const void*)attn_l_bwd_kv_kernel<OF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_bwd: LDS attribute");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_l_bwd_q_kernel<OF>, dim3((unsigned)grid), dim3(256), lds_q, s, base, base + D, base + 2 * D, (long)ld, (const bf16_t*)o, (const bf16_t*)dout, (long)ldo, lse, dvec,
                     dbase, (long)ldd, (int)N, (int)H, scale, (int)(B * H), G);
  hipLaunchKernelGGL(attn_l_bwd_kv_kernel<OF>, dim3((unsigned)grid), dim3(256), lds_kv, s, base, base + D, base + 2 * D, (long)ld, (const bf16_t*)dout, (long)ldo, lse, (const float*)dvec,
                     dbase + D, dbase + 2 * D, (long)ldd, (int)N, (int)H, scale, (int)(B * H), G);
  return VDK_OK;
}
int vdk_attention_long_bwd(const void* qkv, int64_t ld, const void* o, const void* dout, int64_t ldo, const float* lse, void* dqkv, int64_t ldd, float* dvec, int32_t B, int32_t N,
                           int32_t H, float scale, int opf, void* stream) {
  return opf ? al_launch_bwd<VDK_OPF_F16>(qkv, ld, o, dout, ldo, lse, dqkv, ldd, dvec, B, N, H, scale, stream)
             : al_launch_bwd<0>(qkv, ld, o, dout, ldo, lse, dqkv, ldd, dvec, B, N, H, scale, stream);
}

// in-library entry point (attention.hip routes N > 256 here)
template <int OF>
static int al_launch_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H, float scale, void* stream) {
  const bf16_t* base = (const bf16_t*)qkv;
  const long D = (long)H * 64;
  const int nt = (N + 31) / 32, G = (nt + 3) / 4;
  const long units = (long)B * H * G;
  const int grid = al_grid(units);
  const size_t lds = 2 * AL_BUF + 4 * 4096;
  if (hipFuncSetAttribute((const void*)attn_l_fwd_kernel<OF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_fwd: LDS attribute");
  hipLaunchKernelGGL(attn_l_fwd_kernel<OF>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, base, base + D, base + 2 * D, (long)ld, (bf16_t*)o, (long)ldo, lse, (int)N, (int)H,
                     scale, (int)(B * H), G);
  return VDK_OK;
}
int vdk_attention_long_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H, float scale, int opf, void* stream) {
  return opf ? al_launch_fwd<VDK_OPF_F16>(qkv, ld, o, ldo, lse, B, N, H, scale, stream) : al_launch_fwd<0>(qkv, ld, o, ldo, lse, B, N, H, scale, stream);
}

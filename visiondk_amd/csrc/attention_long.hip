// attention_long.hip — K3 forward for sequences beyond the LDS-resident range (N > 256 keys: ViT-L/14 at 336 has N = 576 / 577, ViT-L/14 at 224 N = 257).
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

// rows [r0, r0 + 96) of K and V -> buf, rows >= N read row N-1 (finite filler; its scores are masked)
__device__ __forceinline__ void al_dma_chunk(unsigned char* buf, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long ld, int N, int r0, int w, int lane) {
#pragma unroll
  for (int j0 = 0; j0 < AL_CROWS / 32; ++j0) {
    const int j = w + 4 * j0;
    const int lrow = 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ as_f(lrow);
    int srow = r0 + lrow;
    srow = srow < N ? srow : N - 1;
    const long so = (long)srow * ld + c * 8;
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(k + so), VDK_LDS_PTR(buf + j * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(VDK_GLOBAL_PTR(v + so), VDK_LDS_PTR(buf + AL_ARR + j * 1024), 16, 0, 0);
  }
}

// LDS (dynamic): chunk buffer 0 | chunk buffer 1 | 4 wave store tiles of 4 KB
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
      f32x16 st[AL_CT];
#pragma unroll
      for (int kt = 0; kt < AL_CT; ++kt) {
        st[kt] = as_zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_row_frag_l(Kb + kt * 32 * AS_ROW, al, ks), qf[ks], st[kt], 0, 0, 0);
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
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float p = fast_exp2(fmaf(st[kt][r], scale2, -m)); st[kt][r] = p; l += p; }
        s16x8 pf[2];
        as_pack_b(st[kt], pf);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Vb + (kt * 32 + 16 * s) * AS_ROW, al, 0), pf[s], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_tr_frag_l(Vb + (kt * 32 + 16 * s) * AS_ROW, al, 1), pf[s], o1, 0, 0, 0);
        }
      }
    }
    if (active) {
      l += __shfl_xor(l, 32);
      as_store_tile(Wt, o0, o1, 1.0f / l, o + (long)b * N * ldo + h * 64, ldo, qt * 32, N, lane);
      if (lse && hi == 0 && qrow < N) lse[((long)b * H + h) * N + qrow] = (m + log2f(l)) * 0.6931471805599453f;
    }
    t = tn;
    item = itemn;
  }
}

// in-library entry point (attention.hip routes N > 256 here)
int vdk_attention_long_fwd(const void* qkv, int64_t ld, void* o, int64_t ldo, float* lse, int32_t B, int32_t N, int32_t H, float scale, void* stream) {
  const bf16_t* base = (const bf16_t*)qkv;
  const long D = (long)H * 64;
  const int nt = (N + 31) / 32, G = (nt + 3) / 4;
  const long units = (long)B * H * G;
  int cap = 512;                                                     // two workgroups per CU
  if (const char* e = getenv("VDK_ATTN_GRID")) { const int v = atoi(e); if (v > 0) cap = v; }   // tests: force several units per workgroup
  long grid = units < cap ? units : cap;
  grid = (grid + 7) / 8 * 8;                                         // every XCD residue must be present: items are dealt to XCDs by item mod 8
  const size_t lds = 2 * AL_BUF + 4 * 4096;
  if (hipFuncSetAttribute((const void*)attn_l_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_attention_fwd: LDS attribute");
  hipLaunchKernelGGL(attn_l_fwd_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, base, base + D, base + 2 * D, (long)ld, (bf16_t*)o, (long)ldo, lse, (int)N, (int)H,
                     scale, (int)(B * H), G);
  return VDK_OK;
}

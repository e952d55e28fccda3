// vdk_internal.h -- tuning / debug / test knobs of libvisiondk_hip.so.  NOT part of the ABI a host binds (include/visiondk.h): these exist for the repository's own tests, A/B
// tools (tools/) and profiling; they may change between builds.  visiondk_amd/_abi.py binds them from its INTERNAL table.
#pragma once
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
int vdk_gemm_streamk_grid(int32_t workgroups);   /* tests / tuning: persistent workgroups of the stream-K launch (multiple of 8; 0 = one per CU) */
/* tests / A-B benchmarking only: 0 = automatic choice, 1 = 128x128 register-staged kernel, 2 = 256x256 LDS-DMA kernel
 * (the latter still requires K and the split size to be multiples of 64), 3 = stream-K whenever splitk == -1 lends a workspace, 4 = never stream-K. */
int vdk_gemm_force_kernel(int32_t which);
int vdk_gemm_force_band_cw(int32_t cw);   /* tests: tile order of the one-wave-per-SIMD kernels in column bands of cw tile columns (-1: decided by size, the default) */
/* tests / tuning: which structure served the calling thread's last vdk_gemm_bf16_nt / vdk_margin_cos_pass: 1 = 128x128 register-staged, 2 = 256x256 eight waves,
 * 3 = its stream-K form, 5 = 256x256 four waves (one per SIMD, persistent; gemm_w4.hip; the default for big problems; which = 5 forces it wherever it can
 * serve, environment VDK_GEMM_W4=0 disables it), 6 = 256x128 four waves with two workgroups per CU (gemm_w4h_kernel: the default for the long epilogues --
 * GELU, dGELU, fp32 residual; which = 6 forces it; environment VDK_GEMM_W4H = bit mask 1 GELU | 2 dGELU | 4 residual | 8 other NT | 16 TN) */
int vdk_gemm_last_kernel(void);
/* diagnostic: `workgroups` workgroups (256 threads, 32 KB of LDS) that stay resident for `microseconds` on `stream` -- a stand-in for a collective's kernel in flight */
int vdk_debug_occupy_cus(int32_t workgroups, int64_t microseconds, void* stream);
/* profiling aid: when non-NULL, every 256x256 workgroup writes 4 shader-cycle stamps (start, operands landed, main loop done,
 * stores issued) to buf[4 * workgroup]; NULL (default) disables it */
int vdk_gemm_debug_stamps(void* device_u64_buffer);
/* 1: route every attention call of this thread to the flash-style kernels of csrc/attention.hip (bf16 only), 0: never, -1: environment VDK_ATTN_LEGACY (tests, A/B) */
int vdk_attention_force_legacy(int32_t on);
/* tests: one job of the batched row reduction (vdk_reduce_rows_batch) */
int vdk_debug_reduce_rows_job(float* buf, int64_t ld, int32_t S, int64_t n, float* out, float scale, void* stream);
#ifdef __cplusplus
}
#endif

// gemm_w4_f16.hip — the four-wave 256x256 / 256x128 GEMM kernels of gemm_w4.hip instantiated for IEEE fp16 operands (v_mfma_f32_32x32x16_f16): the operand format of
// the reference's autocast on a GPU (engine/procedure/train.py:118).  Same source, same tile code; a separate translation unit so that the two formats compile in parallel.
#define VDK_W4_OF 1
#include "gemm_w4.hip"

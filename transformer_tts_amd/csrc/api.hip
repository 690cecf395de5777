// Error reporting and ABI version of libfs2_hip.so.
#include "fs2_common.h"

static thread_local char g_err[512] = "";

void fs2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fs2_last_error(void) { return g_err; }
extern "C" int fs2_abi_version(void) { return 9; }   // 2: FS2Gemm.tile_order, attention strip kernels, split-K finish; 3: fs2_gemm_last_tile; 4: fp8 operands (FS2Gemm.scale_a/b), AR decoder kernels; 5: fs2_flash_attn_fwd/bwd; 6: key_info[B][3] (longest-first order); 7: sliced split-K (FS2Gemm.accumulate = 2, fs2_splitk_reduce, fs2_gemm_last_splits); 8: fs2_flash_attention_fwd/bwd (general descriptor), fs2_wgrad_sliced / fs2_wgrad_grouped / fs2_wgrad_reduce, fs2_quantize_fp8_batched; 9: FS2Gemm.q8 (fp8 copy of C written by the epilogue), fs2_quantize_fp8_repair

// One-byte-operand instances of the 16-wave row-major ring kernel (gemm_ring_impl.h, ES = 1): OCP fp8 operands quantised per tensor
// (fs2_quantize_fp8), v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales; 128- and 192-row tiles.
#include "gemm_ring_impl.h"

int fs2_gemm_ring_f8_launch(const FS2Gemm& g, int bm, bool f32, hipStream_t st) {
    // the 192-row tile holds 48 accumulators + 24 fragment registers: only the plain and the residual epilogues fit beside them
    // without scratch (mask / statistics forms: 8-44 B per lane); those run on the 128-row tile
    const int epi = epi_code(g);
    if (epi != 0 && epi != EPI_RES_F32 && epi != EPI_RES_BF16) bm = 128;       // (incl. the forms that also write the fp8 copy of C)
    g_last_tile = bm == 128 ? 130 : 192;
    if (bm == 128) return f32 ? launch_ring1<float, 32, 1>(g, 1, st) : launch_ring1<bf16_t, 32, 1>(g, 1, st);
    return f32 ? launch_ring1<float, 48, 1>(g, 1, st) : launch_ring1<bf16_t, 48, 1>(g, 1, st);
}

// Entry of the 16-wave row-major ring kernel (gemm_ring_impl.h): bf16 instances here, the one-byte-operand instances in gemm_ring_f8.hip.
#include "gemm_ring_impl.h"

int fs2_gemm_ring_f8_launch(const FS2Gemm& g, int bm, bool f32, hipStream_t st);      // gemm_ring_f8.hip

// false: not eligible / not chosen; true: the product was launched on the ring kernel and *rc holds the result.
// Split-K form: g.split_k > 1 with g.accumulate == 2 -- C is an fp32 workspace of split_k slices, slice s (stride g.sC1 elements)
// receives the partial sums of k-range s with plain stores; the caller finishes with fs2_splitk_reduce.
bool fs2_gemm_ring_try(const FS2Gemm& g0, hipStream_t st, int* rc) {
    const FS2Gemm& g = g0;
    // FS2_GEMM_RING: 0 never, 1 (default) where the shape heuristic says so, 2 wherever eligible; FS2_GEMM_BIG_BM forces the row-slab
    // height (128 / 192 / 256).  Read per call so that tests and A/B measurements can switch inside one process.
    const char* e1 = getenv("FS2_GEMM_RING");
    const char* e2 = getenv("FS2_GEMM_BIG_BM");
    const int mode = e1 ? atoi(e1) : 1;
    int bm = e2 ? atoi(e2) : 0;
    const bool sliced = g.accumulate == 2;
    if (mode == 0 && !sliced) return false;
    const bool f8 = g.dtype == FS2_FP8 || g.dtype == FS2_BF8_FP8;
    const int es = f8 ? 1 : 2, sk = 128 / es;
    if ((g.dtype != FS2_BF16 && !f8) || g.a_kmajor || g.b_kmajor || g.conv > 1) return false;
    if (g.q8 != nullptr) {      // fp8 copy of C: the fp8 form of this kernel only, bf16 C with contiguous rows of a multiple of 16 columns
        if (!f8 || g.c_dtype != FS2_BF16 || g.N % 16 != 0 || g.ldc != g.N || !g.q8_state || !g.q8_prev || !fs2_aligned16(g.q8)) return false;
    }
    if (f8 && (sliced || g.K % 16 != 0 || g.lda % 16 != 0 || g.ldb % 16 != 0)) return false;
    if (g.accumulate != 0 && !sliced) return false;
    if ((long)g.batch1 * g.batch2 != 1) return false;
    if (!sliced && g.split_k != 1) return false;
    if (g.N % 8 != 0 || g.K % 8 != 0) return false;
    if (g.N > 2048 && (g.bias != nullptr || g.colstats != nullptr)) return false;       // (bias / column statistics of a launch live in LDS)
    if (g.conv == 1 && (g.pad < 0 || g.pad > g.taps)) return false;
    const int epi = epi_code(g);
    if (!epi_compiled(epi)) return false;
    int splits = 1;
    if (sliced) {
        if (epi != 0 || g.bias || g.relu || g.c_dtype != FS2_F32 || g.alpha != 1.f) return false;
        const int ntot = (g.conv == 1 ? g.taps : 1) * ((g.K + sk - 1) / sk);
        splits = g.split_k < 1 ? 1 : g.split_k;
        if (splits > ntot) splits = ntot;
        const int per = (ntot + splits - 1) / splits;
        splits = (ntot + per - 1) / per;                          // no empty split
        if (((long)splits * g.sC1 + ((long)g.M + 512) * g.ldc) * 4 >= 0x7FFFFFF0L) return false;
        g_last_splits = splits;
    }
    {   // the epilogue addresses C / mask / residual with 32-bit byte offsets (rows up to M + 255 enter the arithmetic)
        const long rows = (long)g.M + 512;
        if (rows * g.ldc * 4 >= 0x7FFFFFF0L) return false;
        if (g.relu_mask != nullptr && rows * g.ldm * 2 >= 0x7FFFFFF0L) return false;
        if (g.residual != nullptr && rows * g.ldr * 4 >= 0x7FFFFFF0L) return false;
        if (rows * g.lda * es >= 0x7FFFFFF0L || ((long)g.N + 512) * g.ldb * es >= 0x7FFFFFF0L) return false;
    }
    const long tn = (g.N + BN - 1) / BN;
    auto fill = [&](int b) {       // fraction of the 256 CUs' rounds that carry a tile
        const long tiles = (long)((g.M + b - 1) / b) * tn * splits;
        return (double)tiles / (double)(((tiles + 255) / 256) * 256);
    };
    if (bm != 128 && bm != 192 && bm != 256) {
        bm = fill(192) - 0.02 > fill(256) ? 192 : 256;
        // Short matrices (the encoder side: 6144 rows) give the 192-row tile fewer tiles than CUs: every CU runs `rounds` tiles of
        // bm rows one after the other, so rounds * bm is the time; the 128-row tile (less operand reuse per MFMA: +10 %) wins when
        // it spreads the same rows over more CUs.
        auto serial_rows = [&](int b) { const long tiles = (long)((g.M + b - 1) / b) * tn * splits; return (double)((tiles + 255) / 256) * b; };
        if (serial_rows(128) * 1.10 < serial_rows(bm)) bm = 128;
    }
    if (bm == 256 && (epi == (EPI_STATS | EPI_SUMSQ) || epi == (EPI_MASK | EPI_RES_F32) || epi == (EPI_MASK | EPI_STATS))) bm = 192;   // (not compiled for 256 rows: spills)
    const bool f32 = g.c_dtype == FS2_F32;
    if (f8 && bm == 256) bm = 192;      // (the 32-byte fp8 fragments leave no registers for the 256-row tile's 64 accumulators)
    if (mode == 1 && !sliced && !f8) {
        const long tiles = (long)((g.M + bm - 1) / bm) * tn;
        if (tiles < 128 || g.N < 192) return false;
    }
    g_last_tile = bm == 128 ? 130 : bm;       // (measurement aid: 130 / 192 / 256 = rows of the 16-wave row-major tile)
    FS2Gemm gd = g0;
    {
        const char* e3 = getenv("FS2_RING_DBG");      // timing-only: drop one buffer's traffic (the kernel's `dbg`); 0 in every product run
        gd.tile_order = e3 ? atoi(e3) : 0;
    }
    if (f8) { *rc = fs2_gemm_ring_f8_launch(gd, bm, f32, st); return true; }
    if (bm == 128) *rc = f32 ? launch_ring1<float, 32>(gd, splits, st) : launch_ring1<bf16_t, 32>(gd, splits, st);
    else if (bm == 192) *rc = f32 ? launch_ring1<float, 48>(gd, splits, st) : launch_ring1<bf16_t, 48>(gd, splits, st);
    else *rc = f32 ? launch_ring1<float, 64>(gd, splits, st) : launch_ring1<bf16_t, 64>(gd, splits, st);
    return true;
}

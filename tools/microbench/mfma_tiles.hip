// Micro-benchmark: which workgroup / wave-tile geometry of a register-staged, double-buffered bf16 stage loop
// (K = 64 per stage, row-major swizzled LDS images, one barrier per stage, L2-resident 2 MiB source) gets how close to
// the 2.5 PFLOP/s MFMA peak on gfx950.  Measurement tool for the next GEMM design, not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
// k-major bf16 image of 64 k-rows x 128 columns (256-byte rows) and its ds_read_tr16 fragment (gemm.hip's Tile<bf16,true>)
__device__ __forceinline__ int km_off(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }
__device__ __forceinline__ bf16x8 km_frag(const unsigned char* lds, int n0, int ks, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int ch = (n0 >> 3) + (pp >> 1), r0 = ks * 32 + 8 * g + q;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + km_off(r0, ch) + 8 * (pp & 1)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + km_off(r0 + 4, ch) + 8 * (pp & 1)));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

// WR x WC waves, each TI x TJ tiles of 16 x 16; OCC = workgroups per CU the launch bound asks for
//   KM : both operands k-major (weight-gradient products): images of 64 k-rows x 128 columns, ds_read_b64_tr_b16 fragments
//   DMA: LDS-DMA staging (global_load_lds_dwordx4 + vmcnt(0) before the barrier) instead of registers + ds_write_b128
template <int WR, int WC, int TI, int TJ, int OCC, bool KM, bool DMA>
__global__ __launch_bounds__(WR * WC * 64, OCC) void tiles_k(const u32x4* __restrict__ src, float* __restrict__ out, int iters, long src_chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = WR * WC * 64, BM = WR * TI * 16, BN = WC * TJ * 16;
    constexpr int ABYTES = BM * 128, BBYTES = BN * 128, STAGE = ABYTES + BBYTES;
    constexpr int CH = STAGE / 16 / NT;            // 16-byte chunks per thread per stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC, g = lane >> 4, i16 = lane & 15;
    int wro[CH];
    for (int i = 0; i < CH; ++i) {
        const int c = tid + NT * i;
        if constexpr (!KM) wro[i] = lds_off(c >> 3, c & 7);                     // A rows then B rows: one image
        else wro[i] = (c >> 10) * 16384 + km_off((c & 1023) >> 4, c & 15);      // consecutive 16 KiB images of 128 columns
    }
    int rdA[2], rdB[2];
    for (int ks = 0; ks < 2; ++ks) { rdA[ks] = lds_off(wr * TI * 16 + i16, ks * 4 + g); rdB[ks] = ABYTES + lds_off(wc * TJ * 16 + i16, ks * 4 + g); }
    for (int i = tid; i < 2 * STAGE / 16; i += NT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x4 acc[TI][TJ];
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 r[CH];
    for (int i = 0; i < CH; ++i) r[i] = u32x4{0x3f803f80u, 0u, 0u, 0u};
    long cursor = (long)blockIdx.x * (CH * NT) + tid;
    for (int s = 0; s < iters; ++s) {
        const unsigned char* cur = smem + (s & 1) * STAGE;
        unsigned char* nxt = smem + ((s + 1) & 1) * STAGE;
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < CH; ++i) *reinterpret_cast<u32x4*>(nxt + wro[i]) = r[i];
#pragma unroll
            for (int i = 0; i < CH; ++i) r[i] = src[(cursor + i * NT) % src_chunks];
        } else {
            // LDS-DMA: each wave-instruction writes 1 KiB (64 lanes x 16 B) linearly at the wave-uniform LDS address
#pragma unroll
            for (int i = 0; i < CH; ++i)
                __builtin_amdgcn_global_load_lds(src + (cursor + i * NT) % src_chunks,
                                                 (__attribute__((address_space(3))) void*)(nxt + (i * NT + wave * 64) * 16), 16, 0, 0);
        }
        cursor += (long)gridDim.x * (CH * NT);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[TI], fb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                if constexpr (!KM) fa[i] = *reinterpret_cast<const bf16x8*>(cur + rdA[ks] + i * 2048);
                else { const int col = wr * TI * 16 + i * 16; fa[i] = km_frag(cur + (col >> 7) * 16384, col & 127, ks, lane); }
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                if constexpr (!KM) fb[j] = *reinterpret_cast<const bf16x8*>(cur + rdB[ks] + j * 2048);
                else { const int col = wc * TJ * 16 + j * 16; fb[j] = km_frag(cur + ABYTES + (col >> 7) * 16384, col & 127, ks, lane); }
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float t = 0.f;
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) out[blockIdx.x * NT + tid] = t;
}

template <int WR, int WC, int TI, int TJ, int OCC, bool KM = false, bool DMA = false> void run(const char* name, const u32x4* src, float* out, long chunks) {
    constexpr int STAGE = (WR * TI + WC * TJ) * 16 * 128;
    auto k = tiles_k<WR, WC, TI, TJ, OCC, KM, DMA>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    const int grid = 256 * OCC, iters = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(grid), dim3(WR * WC * 64), 2 * STAGE, 0, src, out, 64, chunks);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(grid), dim3(WR * WC * 64), 2 * STAGE, 0, src, out, iters, chunks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = (double)grid * WR * WC * iters * (2.0 * TI * TJ) * 16384.0;
    printf("%-58s LDS %3d KiB x %d WG/CU  %8.1f TFLOP/s\n", name, 2 * STAGE / 1024, OCC, flops / (ms * 1e-3) / 1e12);
}

int main() {
    const long chunks = (2L << 20) / 16;
    u32x4* src; float* out;
    hipMalloc(&src, 2L << 20); hipMemset(src, 0x3f, 2L << 20);
    hipMalloc(&out, 8 << 20);
    run<2, 2, 4, 4, 2>("128x128 block, 4 waves of 64x64 (gemm.hip today)", src, out, chunks);
    run<2, 2, 4, 4, 1>("128x128 block, 4 waves of 64x64, 1 WG/CU", src, out, chunks);
    run<4, 2, 4, 4, 1>("256x128 block, 8 waves of 64x64", src, out, chunks);
    run<2, 4, 4, 4, 1>("128x256 block, 8 waves of 64x64", src, out, chunks);
    run<4, 4, 4, 4, 1>("256x256 block, 16 waves of 64x64", src, out, chunks);
    run<2, 2, 8, 4, 1>("256x128 block, 4 waves of 128x64", src, out, chunks);
    run<2, 2, 4, 8, 1>("128x256 block, 4 waves of 64x128", src, out, chunks);
    run<2, 4, 8, 4, 1>("256x256 block, 8 waves of 128x64", src, out, chunks);
    run<4, 2, 4, 8, 1>("256x256 block, 8 waves of 64x128", src, out, chunks);
    run<2, 2, 8, 8, 1>("256x256 block, 4 waves of 128x128", src, out, chunks);
    printf("row-major operands, LDS-DMA staging:\n");
    run<2, 2, 4, 4, 2, false, true>("128x128 block, 4 waves of 64x64", src, out, chunks);
    run<4, 2, 4, 4, 1, false, true>("256x128 block, 8 waves of 64x64", src, out, chunks);
    run<4, 4, 4, 4, 1, false, true>("256x256 block, 16 waves of 64x64", src, out, chunks);
    run<4, 2, 4, 8, 1, false, true>("256x256 block, 8 waves of 64x128", src, out, chunks);
    printf("both operands k-major (weight gradients), register staging:\n");
    run<2, 2, 4, 4, 2, true>("128x128 block, 4 waves of 64x64 (gemm.hip today)", src, out, chunks);
    run<4, 2, 4, 4, 1, true>("256x128 block, 8 waves of 64x64", src, out, chunks);
    run<4, 4, 4, 4, 1, true>("256x256 block, 16 waves of 64x64", src, out, chunks);
    run<4, 2, 4, 8, 1, true>("256x256 block, 8 waves of 64x128", src, out, chunks);
    return 0;
}

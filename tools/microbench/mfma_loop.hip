// Micro-benchmark of the GEMM stage loop on gfx950 (measurement tool, not product code).
// One workgroup = 4 waves (2 x 2), wave tile 64 x 64 (4 x 4 MFMA 16x16x32 bf16), stage = 128 x 64 of A and of B in
// LDS (row-major images, 128-byte rows, XOR swizzle) -- the geometry of gemm.hip.  Variants switch components on:
//   bit 0: fragments re-read from LDS every stage (16 ds_read_b128 per wave)      else: registers, loaded once
//   bit 1: the next stage is written to the other LDS buffer (8 ds_write_b128 per thread) + one barrier per stage
//   bit 2: 8 global 16-byte loads per thread per stage (a 1 GiB stream) feed those writes     else: stale registers
// Reports TFLOP/s for 512 workgroups (2 per CU) and 256 (1 per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ int lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

template <int V, int LEAD>
__global__ __launch_bounds__(256, 2) void loop_k(const u32x4* __restrict__ src, float* __restrict__ out, int iters, long src_chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 buffers x (A 16 KB + B 16 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, g = lane >> 4, i16 = lane & 15;
    int wrA[4], wrB[4];
    for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; wrA[i] = lds_off(c >> 3, c & 7); wrB[i] = wrA[i]; }
    int rdA[2], rdB[2];
    for (int ks = 0; ks < 2; ++ks) { rdA[ks] = lds_off(wr * 64 + i16, ks * 4 + g); rdB[ks] = lds_off(wc * 64 + i16, ks * 4 + g); }
    // fill both buffers once
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4], fb[2][4];
    for (int ks = 0; ks < 2; ++ks) for (int i = 0; i < 4; ++i) {
        fa[ks][i] = *reinterpret_cast<const bf16x8*>(smem + rdA[ks] + i * 2048);
        fb[ks][i] = *reinterpret_cast<const bf16x8*>(smem + 16384 + rdB[ks] + i * 2048);
    }
    u32x4 ra[LEAD][4], rb[LEAD][4];
    for (int l = 0; l < LEAD; ++l) for (int i = 0; i < 4; ++i) { ra[l][i] = u32x4{0x3f803f80u, 0u, 0u, 0u}; rb[l][i] = ra[l][i]; }
    long cursor = (long)blockIdx.x * 2048 + tid;      // a stage = 8 pieces of 256 consecutive 16-byte chunks (coalesced)
    for (int s0 = 0; s0 < iters; s0 += LEAD) {
#pragma unroll
      for (int l = 0; l < LEAD; ++l) {
        const int s = s0 + l;
        const unsigned char* la = smem + (s & 1) * 32768;
        const unsigned char* lb = la + 16384;
        unsigned char* na = smem + ((s + 1) & 1) * 32768;
        if constexpr (V & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { *reinterpret_cast<u32x4*>(na + wrA[i]) = ra[l][i]; *reinterpret_cast<u32x4*>(na + 16384 + wrB[i]) = rb[l][i]; }
        }
        if constexpr (V & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra[l][i] = src[(cursor + i * 256) % src_chunks]; rb[l][i] = src[(cursor + (4 + i) * 256) % src_chunks]; }
            cursor += (long)gridDim.x * 2048;
        }
        if constexpr (V & 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[ks][i] = *reinterpret_cast<const bf16x8*>(la + rdA[ks] + i * 2048);
                    fb[ks][i] = *reinterpret_cast<const bf16x8*>(lb + rdB[ks] + i * 2048);
                }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
        if constexpr (V & 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) out[blockIdx.x * 256 + tid] = t;    // keep the accumulators alive
}

template <int V, int LEAD = 1> double run(int grid, int iters, const u32x4* src, float* out, long chunks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_k<V, LEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((loop_k<V, LEAD>), dim3(grid), dim3(256), 65536, 0, src, out, 64, chunks);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((loop_k<V, LEAD>), dim3(grid), dim3(256), 65536, 0, src, out, iters, chunks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = (double)grid * 4 * iters * 32 * 16384.0;
    return flops / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    // source window in MiB (default 1024 = an HBM stream; 2 = L2-resident, the panel-reuse regime of a real product)
    const long mib = argc > 1 ? atol(argv[1]) : 1024;
    const long chunks = (mib << 20) / 16;
    printf("source window %ld MiB\n", mib);
    u32x4* src; float* out;
    hipMalloc(&src, 1L << 30); hipMemset(src, 0x3f, 1L << 30);
    hipMalloc(&out, 4 << 20);
    const int iters = 3072;
    const char* names[8] = {"mfma only", "+ds_read", "+ds_write+barrier", "+ds_read +ds_write+barrier", "+global", "+ds_read +global", "+ds_write+barrier +global", "all"};
    for (int grid : {512, 256}) {
        printf("grid %d (%d workgroup(s) per CU)\n", grid, grid / 256);
        double r[8];
        r[0] = run<0>(grid, iters, src, out, chunks); r[1] = run<1>(grid, iters, src, out, chunks);
        r[2] = run<2>(grid, iters, src, out, chunks); r[3] = run<3>(grid, iters, src, out, chunks);
        r[4] = run<4>(grid, iters, src, out, chunks); r[5] = run<5>(grid, iters, src, out, chunks);
        r[6] = run<6>(grid, iters, src, out, chunks); r[7] = run<7>(grid, iters, src, out, chunks);
        for (int v = 0; v < 8; ++v) printf("   %-32s %8.1f TFLOP/s\n", names[v], r[v]);
        printf("   all, loads consumed 2 stages later %8.1f TFLOP/s\n", run<7, 2>(grid, iters, src, out, chunks));
        printf("   all, loads consumed 3 stages later %8.1f TFLOP/s\n", run<7, 3>(grid, iters, src, out, chunks));
        printf("   all, loads consumed 4 stages later %8.1f TFLOP/s\n", run<7, 4>(grid, iters, src, out, chunks));
    }
    return 0;
}

// Large-tile bf16 MFMA GEMM for gfx950: the tall row-major x row-major products of the decoder side of the model
// (M = batch * frames ~ 44k rows; linear layers and implicit-GEMM Conv1d, forward and data gradient).
//
// Why a second kernel: the 128x128 / 4-wave stage loop of gemm.hip tops out at ~775 TFLOP/s with
// L2-resident operands; a 256x256 block tile worked on by 16 waves of 64x64 (<= 128 VGPRs, four waves per SIMD, one
// workgroup per CU) halves the staging traffic per MFMA and reaches ~1400 in the same micro-benchmark.
//
// Structure
//   * block tile BM x BN = (WR*64) x (WC*64), WR x WC waves, each wave a 64x64 output tile = 4x4 MFMA 16x16x32 tiles;
//   * K in stages of 64 bf16 (128 B per row); two LDS stage buffers [A: BM rows][B: BN rows] x 128 B, 16-B chunk c of
//     row r stored at c ^ ((r>>1)&7) (conflict-free ds_read_b128);
//   * staging by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  One wave-instruction
//     writes 1 KiB = 8 rows linearly, so the swizzle is applied on the per-lane SOURCE address (lane -> logical chunk
//     (lane&7) ^ ((row>>1)&7) of row lane>>3).  Rows outside the matrix, conv halo rows and the K tail use an
//     out-of-range buffer offset: the DMA writes zeros;
//   * operands are passed to the MFMA swapped (A-operand = weight rows, B-operand = activation rows): a lane then
//     holds FOUR CONSECUTIVE OUTPUT COLUMNS of one output row, so the epilogue goes straight from the accumulators to
//     8-byte (bf16) / 16-byte (fp32) global stores -- no LDS round trip, no barriers, and the LDS stays free for the
//     next tile's first stage, which is already in flight while the epilogue runs (persistent workgroups walk a
//     flattened stream of stages over all their tiles);
//   * tile walk: workgroups b and b+8 share an XCD (round-robin dispatch).  XCD x owns the row slabs mt = x (mod 8);
//     the column tiles of a slab are taken by different CUs of that XCD at the same time, so an A slab is fetched
//     from HBM once and hit in that XCD's L2 by the other column tiles (placement is a speed assumption only).
//   * fused epilogue: alpha, bias, ReLU, ReLU mask, residual (fp32 / bf16), fp32 / bf16 output, per-column statistics
//     (BatchNorm sums or bias-gradient column sums: DPP row reduction over the 16 lanes that share a column group,
//     LDS accumulators, one global atomic per column per workgroup).
#include "common.cuh"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr unsigned OOB = 0x80000000u;
constexpr int BIG_COLSTAT_N = 1024;

template <int WR, int WC> struct BG {
    static constexpr int NW = WR * WC, NT = NW * 64, BM = WR * 64, BN = WC * 64;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;     // LDS-DMA wave-instructions per wave per stage
    static constexpr int SMEM = 2 * STAGE;
};

__device__ __forceinline__ int sw_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

template <int N> struct IC { static constexpr int value = N; };

}  // namespace

template <typename TC, int WR, int WC>
__global__ __launch_bounds__(WR * WC * 64, WR * WC / 4) void fs2_gemm_big_kernel(const FS2Gemm p, const int tilesM, const int tilesN) {
    typedef BG<WR, WC> G;
    constexpr int NW = G::NW, NT = G::NT, BM = G::BM, BN = G::BN, AI = G::AI, BI = G::BI;
    constexpr int ESC = (int)sizeof(TC);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int g = lane >> 4, i16 = lane & 15;

    // ---- work of this block: items j = slot, slot + nslots, ... of its XCD group's list (slab-major, column tile fastest)
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int slabs = tilesM > x ? (tilesM - x + 7) >> 3 : 0;
    const int items = slabs * tilesN;
    const int nmine = items > slot ? (items - slot + nslots - 1) / nslots : 0;
    if (nmine == 0) return;
    const bool conv = p.conv == 1;
    const int nkt = (p.K + 63) >> 6;
    const int ntot = (conv ? p.taps : 1) * nkt;
    const int nst = nmine * ntot;
    const int pad = conv ? p.pad : 0;
    const int lda = (int)p.lda, ldb = (int)p.ldb;

    float* cacc = reinterpret_cast<float*>(smem + G::SMEM);
    const bool stats = p.colstats != nullptr;
    if (stats) {
        for (int n = tid; n < 2 * BIG_COLSTAT_N; n += NT) cacc[n] = 0.f;      // visible after the prologue's barrier
    }

    // conv: the descriptor base is moved back by `pad` rows so that the scalar stage offset (tap*lda + kb) is never negative
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const bf16_t*>(p.A) - (int64_t)pad * lda), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7FFFFFF0, 0x00020000);
    // epilogue operands through buffer descriptors (32-bit offsets, out-of-range offset = no access); an absent operand
    // gets a descriptor with zero records: its loads return zeros without a branch
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)p.relu_mask, 0, p.relu_mask ? 0x7FFFFFF0 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? 0x7FFFFFF0 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);

    // ---- LDS-DMA source coordinates of this lane: instruction i of this wave covers tile rows 8*(i*NW + wave) .. +7;
    //      the lane fetches logical chunk (lane&7) ^ ((row>>1)&7) of row lane>>3 of those (swizzle on the source side)
    auto dma_row = [&](int i) { return 8 * (i * NW + wave) + (lane >> 3); };
    auto dma_k8 = [&](int i) { return ((lane & 7) ^ ((dma_row(i) >> 1) & 7)) * 8; };

    // ---- load cursor
    int lj = slot, lst = 0, ltap = 0, lkb = 0, lleft = nmine;
    unsigned voffA[AI], voffB[BI];
    int tA[AI];
    auto prep_item = [&](int j) {
        const int q = j / tilesN, nt = j - q * tilesN;
        const int m0 = (x + 8 * q) * BM, n0 = nt * BN;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = m0 + dma_row(i);
            voffA[i] = (m < p.M) ? (unsigned)((m * lda + dma_k8(i)) * 2) : OOB;
            tA[i] = conv ? (m % p.seq_len) - pad : 0;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int n = n0 + dma_row(i);
            voffB[i] = (n < p.N) ? (unsigned)((n * ldb + dma_k8(i)) * 2) : OOB;
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        const int kb = lkb, tap = ltap;
        const int sA = (tap * lda + kb) * 2;
        const int sB = (tap * p.K + kb) * 2;
        unsigned char* base = smem + buf * G::STAGE + 1024 * wave;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            bool ok = (voffA[i] != OOB) && (kb + dma_k8(i) < p.K);
            if (conv) ok = ok && ((unsigned)(tA[i] + tap) < (unsigned)p.seq_len);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(base + 1024 * NW * i), 16, (int)(ok ? voffA[i] : OOB), sA, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const bool ok = (voffB[i] != OOB) && (kb + dma_k8(i) < p.K);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(base + G::A_BYTES + 1024 * NW * i), 16, (int)(ok ? voffB[i] : OOB), sB, 0, 0);
        }
        lkb += 64;
        if (lkb >= p.K) { lkb = 0; ++ltap; }
        if (++lst == ntot) {
            lst = 0; ltap = 0; lkb = 0;
            lj += nslots;
            if (--lleft > 0) prep_item(lj);
        }
    };

    // ---- fragment read addresses (lane part; tile i adds i*2048, stage buffer b adds b*STAGE)
    int rdA[2], rdB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        rdA[ks] = sw_off(wr * 64 + i16, ks * 4 + g);
        rdB[ks] = G::A_BYTES + sw_off(wc * 64 + i16, ks * 4 + g);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- compute cursor
    int cj = slot, cst = 0;
    bool pending = false;
    int pj = 0;

    prep_item(lj);
    issue(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int s = 0; ; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) issue(buf ^ 1);      // stage s+1 -> the buffer every wave finished reading at the last barrier
        if (pending) {
            // ---- epilogue of item pj: lane holds C[m0 + wr*64 + it*16 + i16][n0 + wc*64 + jt*16 + 4g + 0..3] in acc[it][jt]
            pending = false;
            const int q = pj / tilesN, nt = pj - q * tilesN;
            const int mb = (x + 8 * q) * BM + wr * 64 + i16;
            const int nb = nt * BN + wc * 64 + 4 * g;
            const bool res_f32 = p.res_dtype == FS2_F32;
            const unsigned offC = (unsigned)((mb * (int)p.ldc + nb) * ESC);
            const unsigned offM = (unsigned)((mb * (int)p.ldm + nb) * 2);
            const unsigned offR = (unsigned)((mb * (int)p.ldr + nb) * (res_f32 ? 4 : 2));
            const bool has_mask = p.relu_mask != nullptr;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int n = nb + 16 * jt;
                const bool col_ok = n < p.N;
                const u32x4 braw = __builtin_amdgcn_raw_buffer_load_b128(rsBias, col_ok ? (unsigned)(n * 4) : OOB, 0, 0);
                float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f);
                u32x2 mraw[4];
                u32x4 rraw[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const bool ok = col_ok && (mb + 16 * it < p.M);
                    mraw[it] = __builtin_amdgcn_raw_buffer_load_b64(rsM, ok ? offM : OOB, (16 * it * (int)p.ldm + 16 * jt) * 2, 0);
                    if (res_f32) rraw[it] = __builtin_amdgcn_raw_buffer_load_b128(rsR, ok ? offR : OOB, (16 * it * (int)p.ldr + 16 * jt) * 4, 0);
                    else {
                        const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rsR, ok ? offR : OOB, (16 * it * (int)p.ldr + 16 * jt) * 2, 0);
                        rraw[it] = u32x4{h.x << 16, h.x & 0xFFFF0000u, h.y << 16, h.y & 0xFFFF0000u};
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const bool ok = col_ok && (mb + 16 * it < p.M);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = acc[it][jt][r] * p.alpha + __uint_as_float(braw[r]);
                        if (p.relu) t = fmaxf(t, 0.f);
                        v[r] = t;
                    }
                    if (has_mask) {
                        v[0] = __uint_as_float(mraw[it].x << 16) > 0.f ? v[0] : 0.f;
                        v[1] = __uint_as_float(mraw[it].x & 0xFFFF0000u) > 0.f ? v[1] : 0.f;
                        v[2] = __uint_as_float(mraw[it].y << 16) > 0.f ? v[2] : 0.f;
                        v[3] = __uint_as_float(mraw[it].y & 0xFFFF0000u) > 0.f ? v[3] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += __uint_as_float(rraw[it][r]);
                    const int soff = (16 * it * (int)p.ldc + 16 * jt) * ESC;
                    if constexpr (ESC == 4) {
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                               rsC, ok ? offC : OOB, soff, 0);
                    } else {
                        union { bf16x4 h; u32x2 u; } o;
                        o.h[0] = (bf16_t)v[0]; o.h[1] = (bf16_t)v[1]; o.h[2] = (bf16_t)v[2]; o.h[3] = (bf16_t)v[3];
                        __builtin_amdgcn_raw_buffer_store_b64(o.u, rsC, ok ? offC : OOB, soff, 0);
                        if (stats) {   // statistics of the values as stored
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (float)o.h[r];
                        }
                    }
                    if (stats && ok) {
                        cs.x += v[0]; cs.y += v[1]; cs.z += v[2]; cs.w += v[3];
                        cq.x += v[0] * v[0]; cq.y += v[1] * v[1]; cq.z += v[2] * v[2]; cq.w += v[3] * v[3];
                    }
                }
                if (stats) {
                    // the 16 lanes of a DPP row share g (the column group) and hold 16 different rows: four DPP steps
                    // leave the row total in every lane of the row
                    float* f[2] = {&cs.x, &cq.x};
#pragma unroll
                    for (int wq = 0; wq < 2; ++wq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float sv = f[wq][e];
                            sv += dpp_mov<0xB1>(sv); sv += dpp_mov<0x4E>(sv); sv += dpp_mov<0x124>(sv); sv += dpp_mov<0x128>(sv);
                            f[wq][e] = sv;
                        }
                    if (i16 == 0 && col_ok) {
                        atomicAdd(cacc + n + 0, cs.x); atomicAdd(cacc + n + 1, cs.y);
                        atomicAdd(cacc + n + 2, cs.z); atomicAdd(cacc + n + 3, cs.w);
                        if (p.colstats_mode == 0) {
                            atomicAdd(cacc + BIG_COLSTAT_N + n + 0, cq.x); atomicAdd(cacc + BIG_COLSTAT_N + n + 1, cq.y);
                            atomicAdd(cacc + BIG_COLSTAT_N + n + 2, cq.z); atomicAdd(cacc + BIG_COLSTAT_N + n + 3, cq.w);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (s == nst) break;
        const unsigned char* lb = smem + buf * G::STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(lb + rdA[ks] + i * 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(lb + rdB[ks] + j * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D[n][m]
        }
        if (++cst == ntot) { pending = true; pj = cj; cst = 0; cj += nslots; }
        // own DMA landed, own fragment reads retired; then every wave's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    if (stats) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave's LDS adds are done
        for (int n = tid; n < p.N; n += NT) {
            const float a = cacc[n];
            if (a != 0.f) atomicAdd(p.colstats + n, a);
            if (p.colstats_mode == 0) {
                const float q = cacc[BIG_COLSTAT_N + n];
                if (q != 0.f) atomicAdd(p.colstats + p.N + n, q);
            }
        }
    }
}

namespace {

template <typename TC, int WR, int WC>
int launch_big1(const FS2Gemm& g, hipStream_t st) {
    typedef BG<WR, WC> G;
    const int tilesM = (g.M + G::BM - 1) / G::BM, tilesN = (g.N + G::BN - 1) / G::BN;
    const int lds = G::SMEM + (g.colstats != nullptr ? 2 * BIG_COLSTAT_N * 4 : 0);
    static bool attr_set = false;          // > 64 KiB of dynamic LDS must be allowed once per kernel
    if (!attr_set) {
        // (voffset of the LDS-DMA builtin must be an int expression: an unsigned one makes the host-side instantiation of
        //  the kernel template fail silently -- no stub, undefined symbol at load time)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_big_kernel<TC, WR, WC>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM + 2 * BIG_COLSTAT_N * 4) != hipSuccess) {
            fs2_set_error("fs2_gemm: cannot raise the dynamic LDS limit of the large-tile kernel");
            return FS2_ELAUNCH;
        }
        attr_set = true;
    }
    const long per_xcd = (long)((tilesM + 7) / 8) * tilesN;
    const int grid = 8 * (int)(per_xcd < 32 ? per_xcd : 32);
    hipLaunchKernelGGL((fs2_gemm_big_kernel<TC, WR, WC>), dim3(grid), dim3(G::NT), lds, st, g, tilesM, tilesN);
    FS2_CHECK_LAUNCH("fs2_gemm(big)");
    return FS2_OK;
}

}  // namespace

// 0: not eligible / not chosen; otherwise the product was launched on the large-tile kernel and *rc holds the result
bool fs2_gemm_big_try(const FS2Gemm& g, hipStream_t st, int* rc) {
    // FS2_GEMM_BIG: 0 never, 1 (default) where the shape heuristic says so, 2 wherever eligible; read per call so that
    // tests and A/B measurements can switch inside one process
    const char* e1 = getenv("FS2_GEMM_BIG");
    const char* e2 = getenv("FS2_GEMM_BIG_CFG");
    const int mode = e1 ? atoi(e1) : 1, cfg = e2 ? atoi(e2) : 44;
    if (mode == 0) return false;
    if (g.dtype != FS2_BF16 || g.a_kmajor || g.b_kmajor || g.accumulate || g.conv > 1) return false;
    if ((long)g.batch1 * g.batch2 * g.split_k != 1) return false;
    if (g.N % 8 != 0 || (g.colstats != nullptr && g.N > BIG_COLSTAT_N)) return false;
    if (g.conv == 1 && (g.pad < 0 || g.pad > g.taps)) return false;
    {   // the epilogue addresses C / mask / residual with 32-bit byte offsets (rows up to M + 255 enter the arithmetic)
        const long rows = (long)g.M + 512;
        if (rows * g.ldc * 4 >= 0x7FFFFFF0L) return false;
        if (g.relu_mask != nullptr && rows * g.ldm * 2 >= 0x7FFFFFF0L) return false;
        if (g.residual != nullptr && rows * g.ldr * 4 >= 0x7FFFFFF0L) return false;
        if (rows * g.lda * 2 >= 0x7FFFFFF0L || ((long)g.N + 512) * g.ldb * 2 >= 0x7FFFFFF0L) return false;
    }
    const int bm = cfg == 24 ? 128 : 256, bn = cfg == 42 ? 128 : 256;
    if (mode == 1) {
        const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
        if (tiles < 128 || g.N < 192) return false;
    }
    const bool f32 = g.c_dtype == FS2_F32;
    if (cfg == 42) *rc = f32 ? launch_big1<float, 4, 2>(g, st) : launch_big1<bf16_t, 4, 2>(g, st);
    else if (cfg == 24) *rc = f32 ? launch_big1<float, 2, 4>(g, st) : launch_big1<bf16_t, 2, 4>(g, st);
    else *rc = f32 ? launch_big1<float, 4, 4>(g, st) : launch_big1<bf16_t, 4, 4>(g, st);
    return true;
}

// Weights-stationary streaming GEMM for gfx950: C[M][N] = A[M][256] W[N][256]^T (+ fused epilogue), bf16, for the TALL products of
// the model whose reduction is one model width (d_model = 256): the fused q/v/k projection (N = 768), the attention output
// projection and its data gradient (N = 256), the first FeedForward layer of the decoder (k = 1: N = 1024) and the data gradient of
// the second (N = 1024, ReLU mask + bias-gradient column sums).  M = batch x frames ~ 44 k rows.
//
// Why a third GEMM kernel ("K = 256"): these products move 5-10 bytes per MFMA-FLOP less than the machine
// balance -- they are HBM streams (A read once, C written once; QKV: 23 MB in, 68 MB out = 14.5 us at 6.3 TB/s against 7 us of
// MFMA work).  A tiled GEMM runs them as ~700 tiles of 4 k-stages each: every tile re-stages its 128 KiB weight tile through the
// CU's LDS-DMA path (QKV: 156 MB staged for 23 MB of operands), pays a pipeline fill and an epilogue per tile, and all 256
// workgroups alternate between a load/compute phase and a store burst in lock step (measured: 31-33 us, the stores alone 8 us).
// Here a workgroup keeps its 256 output columns for its whole life:
//   * 8 waves = 2 row halves x 4 column groups; wave (wr, wc) holds W[64 wc .. 64 wc + 63][0..255] in 128 REGISTERS as MFMA
//     operand fragments (loaded once, straight from global memory / L2) -- the weights never touch LDS again;
//   * the activations stream through LDS in SLABS of 64 rows x 512 B (four 64-row x 128-byte quarter images, 16-byte chunk c of
//     row r at c ^ ((r >> 1) & 7): conflict-free ds_read_b128), filled by LDS-DMA in whole 128-byte lines, a ring of 4 slabs
//     (128 KiB): three slabs of lookahead, waits are counted (`s_waitcnt vmcnt(N)`, N = everything this wave has issued since the
//     awaited pieces; gfx9 retires vector-memory operations in issue order), one barrier per slab;
//   * per slab a wave multiplies its 32 rows x 64 columns x 256 (64 MFMA 16x16x32) and stores them at once (register-direct
//     epilogue, a lane holds 16 consecutive columns of a row: 16-byte stores): loads and stores are interleaved at 64-row
//     granularity on every CU all the time instead of chip-wide bursts;
//   * per-column statistics (bias-gradient sums, BatchNorm sums) stay in registers over ALL slabs of the workgroup and are reduced
//     across lanes once, at the end; the bias sits in 16 registers;
//   * workgroups of one XCD that share a row range take different column tiles at the same time (A is fetched from HBM once and
//     hit in that XCD's L2 by the other column tiles); placement is a speed assumption only.
// Same MFMA, same k order per accumulator as the other kernels: results are bit-identical to gemm_ring.hip / gemm.hip.
#include "fs2_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr unsigned OOB = 0x80000000u;
constexpr int WS_K = 256, WS_ROWS = 64, WS_BN = 256, WS_NT = 512, WS_NW = 8;
constexpr int WS_SLAB = WS_ROWS * WS_K * 2;        // 32 KiB
constexpr int WS_RING = 4;
constexpr int WS_SMEM = WS_RING * WS_SLAB;         // 128 KiB

constexpr int EPI_MASK = 1, EPI_RES_F32 = 2, EPI_RES_BF16 = 4, EPI_STATS = 8, EPI_SUMSQ = 16;

__device__ __forceinline__ int fA(int r) { return (r >> 1) & 7; }

// wait until at most 8 + NST * nepi of this wave's vector-memory operations are outstanding (the pieces of two younger slabs and the stores of
// `nepi` <= 3 epilogues), and for its LDS reads; then the workgroup barrier.  s_waitcnt takes an immediate: four cases, the steady state
// (nepi = 3) first -- a switch over every count compiled into a tree of taken branches on the path between a barrier and the work behind it
template <int NST>
__device__ __forceinline__ void ws_wait_barrier(int nepi) {
    constexpr int N3 = 8 + 3 * NST > 63 ? 63 : 8 + 3 * NST, N2 = 8 + 2 * NST > 63 ? 63 : 8 + 2 * NST, N1 = 8 + NST;
    if (__builtin_expect(nepi >= 3, 1)) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N3) : "memory"); return; }
    if (nepi == 2) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N2) : "memory"); return; }
    if (nepi == 1) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N1) : "memory"); return; }
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

extern thread_local int g_last_tile;     // gemm.hip

// grid: 8 * per_xcd workgroups; XCD group x = blockIdx & 7, slot = blockIdx >> 3 = (row range w_local) * tilesN + column tile
template <typename TC, int EPI>
__global__ __launch_bounds__(512, 2) void fs2_gemm_ws_kernel(const FS2Gemm p, const int tilesN, const int ranges_per_xcd, const int slabs_per_range) {
    constexpr int ESC = (int)sizeof(TC);
    constexpr bool HAS_MASK = (EPI & EPI_MASK) != 0, RES_F32 = (EPI & EPI_RES_F32) != 0, RES_BF16 = (EPI & EPI_RES_BF16) != 0;
    constexpr bool STATS = (EPI & EPI_STATS) != 0, SUMSQ = (EPI & EPI_SUMSQ) != 0;
    constexpr int NSTORES = 2 * (ESC == 4 ? 4 : 2);                  // 16-byte stores of one slab, per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, i16 = lane & 15;

    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wl = slot / tilesN, nt = slot - wl * tilesN;
    const int range = x * ranges_per_xcd + wl;                       // row range of this workgroup (shared by the column tiles of its XCD)
    const int nslabs_all = (p.M + WS_ROWS - 1) / WS_ROWS;
    const int s_beg = range * slabs_per_range;
    const int ns = min(slabs_per_range, nslabs_all - s_beg);         // slabs of this workgroup
    if (wl >= ranges_per_xcd || ns <= 0) return;
    const int n0 = nt * WS_BN;
    const int lda = (int)p.lda, ldb = (int)p.ldb;

    // ---- LDS-DMA of the activation slabs: this wave's pieces cover slab rows 8 wave .. 8 wave + 7, one piece per 128-byte k-quarter
    // (timing-only switches, FS2_WS_DBG: 1 = the output descriptor has zero records (stores dropped), 2 = the activation descriptor
    //  (zeros staged): prices one buffer's traffic with the instruction stream unchanged -- CDNA4 guide section 7; results are wrong)
    const int dbg = p.tile_order;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (dbg & 2) ? 0 : 0x7FFFFFF0, 0x00020000);
    const int drow = 8 * wave + (lane >> 3);
    const int dk8 = ((lane & 7) ^ fA(drow)) * 8;
    int li = 0, rp_i = 0;                                            // next slab to request, ring position
    auto issue = [&]() __attribute__((always_inline)) {
        // (always issued: past the workgroup's last slab every lane is out of range -- zeros land in a ring position nobody reads; a
        //  conditional issue would leave the compiler two paths to merge when it counts vmcnt for the mask / residual loads)
        const int m = (s_beg + li) * WS_ROWS + drow;
        const unsigned voff = (li < ns && m < p.M) ? (unsigned)((m * lda + dk8) * 2) : OOB;
        unsigned char* base = smem + rp_i + 1024 * wave;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(base + 8192 * kq), 16, (int)voff, 128 * kq, 0, 0);
        ++li;
        rp_i = (rp_i + WS_SLAB) & (WS_SMEM - 1);
    };
    // ---- the weights of this wave's 64 columns as MFMA A-operand fragments: tile jt takes the weight rows 16 (i>>2) + 4 jt + (i&3),
    //      i = 0..15 (so that a lane ends up with 16 consecutive output columns), k-step ks the columns 32 ks + 8 g .. + 7.
    //      The 256 x 256 weight tile (128 KiB = the whole ring) is staged ONCE through LDS by LDS-DMA in whole 128-byte lines, in the
    //      slab image format (column group j = "slab" j), and every wave reads its 32 fragments from there.  Fetching the fragments
    //      straight from global memory (16 rows x 64 B per wave-instruction, both row halves each their own copy: 256 KiB per
    //      workgroup through the texture addresser) cost ~7 us of the kernel's ~10 us of fixed time.
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 64 * j + drow;
        const unsigned voff = n < p.N ? (unsigned)((n * ldb + dk8) * 2) : OOB;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(smem + WS_SLAB * j + 8192 * kq + 1024 * wave), 16, (int)voff, 128 * kq, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    bf16x8 wf[8][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int rl = 16 * (i16 >> 2) + 4 * jt + (i16 & 3);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wf[ks][jt] = *reinterpret_cast<const bf16x8*>(smem + WS_SLAB * wc + 8192 * (ks >> 1) + rl * 128 + ((((ks & 1) * 4 + g) ^ fA(rl)) << 4));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave holds its weights: the ring is free for the activations
#pragma unroll
    for (int i = 0; i < WS_RING - 1; ++i) issue();

    // bias of the lane's 16 columns
    const int nb = n0 + wc * 64 + 16 * g;
    const bool ok_lo = nb < p.N, ok_hi = nb + 8 < p.N;               // N is a multiple of 8
    float bias[16];
    {
        const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(rsBias, (nb + 4 * j) * 4, 0, 0);      // absent bias / columns >= N: zeros
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[4 * j + r] = __uint_as_float(b4[r]);
        }
    }
    float cs[STATS ? 16 : 1], cq[SUMSQ ? 16 : 1];
    if constexpr (STATS) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { cs[c] = 0.f; if constexpr (SUMSQ) cq[c] = 0.f; }
    }
    const float alpha = p.alpha;
    const bool relu = p.relu != 0;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (dbg & 1) ? 0 : 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)p.relu_mask, 0, p.relu_mask ? 0x7FFFFFF0 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? 0x7FFFFFF0 : 0, 0x00020000);

    // fragment read address (lane part): row wr*32 + 16 it + i16 of quarter image kq, chunk (ks&1)*4 + g
    int rdA[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = wr * 32 + i16;
        rdA[h] = r * 128 + (((h * 4 + g) ^ fA(r)) << 4);           // (fA(r + 16) = fA(r): row tile it adds 2048 bytes)
    }
    // all of the prologue's register loads are waited for HERE, with the builtin (the compiler then knows the weight fragments have
    // landed and places no vmcnt for them inside the loop); vmcnt(0) also lands the first slabs: once per kernel
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // ---- Ping-pong of the two row halves.  A slab costs a wave ~1000 cycles of MFMA issue and ~1000 cycles of epilogue VALU work
    // (scale, bias, ReLU / mask, pack, stores).  With both waves of a SIMD in the same phase the matrix pipe idles through every
    // epilogue (measured with loads AND stores switched off: 2.3 us per slab against 1.0 us of MFMA time).  The waves of row half 1
    // therefore run ONE HALF-STEP BEHIND those of row half 0 (they pass one extra barrier before their first slab, row half 0 one
    // after its last): at every moment each SIMD has one wave in its MFMA phase and one in its epilogue.  Two barriers per slab:
    //   barrier 2s+1: half 0 enters MFMA(s), half 1 epilogue(s-1); before it EVERY wave has waited for its LDS-DMA pieces of slab s,
    //                 after it every wave requests slab s+3 into the ring position of slab s-1 (half 1 finished reading it);
    //   barrier 2s+2: half 0 enters epilogue(s), half 1 MFMA(s).
    // Counted waits: at barrier 2s+1 a wave has issued, since its pieces of slab s, the pieces of two younger slabs and the stores of
    // the epilogues it ran since (row half 0: slabs s-3 .. s-1, row half 1: s-4 .. s-2, as far as they exist).
    f32x4 acc[2][4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    u32x4 mraw[HAS_MASK ? 2 : 1][2], rraw[(RES_F32 || RES_BF16) ? 2 : 1][RES_F32 ? 4 : 2];
    auto plain_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto new_slab_barrier = [&](int sl) __attribute__((always_inline)) {
        if (sl >= ns) { plain_barrier(); return; }
        int nepi = wr == 0 ? sl : sl - 1;
        nepi = nepi < 0 ? 0 : (nepi > WS_RING - 1 ? WS_RING - 1 : nepi);
        static_assert(WS_RING == 4, "the wait counts assume two younger slabs");
        ws_wait_barrier<NSTORES>(nepi);
        issue();
    };
    auto mfma_phase = [&](int sl) __attribute__((always_inline)) {
        const int m0 = (s_beg + sl) * WS_ROWS + wr * 32 + i16;
        // mask / residual rows of this slab, half a step ahead of the epilogue that uses them
        if constexpr (HAS_MASK) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const bool row_ok = m0 + 16 * it < p.M;
                const unsigned offM = (unsigned)(((m0 + 16 * it) * (int)p.ldm + nb) * 2);
                mraw[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && ok_lo) ? offM : OOB, 0, 0);
                mraw[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && ok_hi) ? offM + 16 : OOB, 0, 0);
            }
        }
        if constexpr (RES_F32) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const bool row_ok = m0 + 16 * it < p.M;
                const unsigned offR = (unsigned)(((m0 + 16 * it) * (int)p.ldr + nb) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rraw[it][j] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (row_ok && (j < 2 ? ok_lo : ok_hi)) ? offR + 16 * j : OOB, 0, 0);
            }
        }
        if constexpr (RES_BF16) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const bool row_ok = m0 + 16 * it < p.M;
                const unsigned offR = (unsigned)(((m0 + 16 * it) * (int)p.ldr + nb) * 2);
                rraw[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (row_ok && ok_lo) ? offR : OOB, 0, 0);
                rraw[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (row_ok && ok_hi) ? offR + 16 : OOB, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* lb = smem + (sl & (WS_RING - 1)) * WS_SLAB;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bf16x8 fa[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) fa[it] = *reinterpret_cast<const bf16x8*>(lb + 8192 * (ks >> 1) + rdA[ks & 1] + 2048 * it);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)      // D[n][m]; the first k-step starts from zero (no separate clearing of 32 registers)
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][jt], fa[it], ks == 0 ? zero4 : acc[it][jt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // epilogue of the slab: lane holds C[m0 + 16 it][nb + 4 jt + r] in acc[it][jt][r]
    auto epilogue = [&](int sl) __attribute__((always_inline)) {
        const int m0 = (s_beg + sl) * WS_ROWS + wr * 32 + i16;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool row_ok = m0 + 16 * it < p.M;
            const bool oka = row_ok && ok_lo, okb = row_ok && ok_hi;
            const unsigned offC = (unsigned)(((m0 + 16 * it) * (int)p.ldc + nb) * ESC);
            float v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float t = acc[it][c >> 2][c & 3] * alpha + bias[c];
                if (relu) t = fmaxf(t, 0.f);
                if constexpr (HAS_MASK) {
                    const unsigned w = mraw[it][c >> 3][(c >> 1) & 3];
                    const float mk = __uint_as_float((c & 1) ? (w & 0xFFFF0000u) : (w << 16));
                    t = mk > 0.f ? t : 0.f;
                }
                if constexpr (RES_F32) t += __uint_as_float(rraw[it][c >> 2][c & 3]);
                if constexpr (RES_BF16) {
                    const unsigned w = rraw[it][c >> 3][(c >> 1) & 3];
                    t += __uint_as_float((c & 1) ? (w & 0xFFFF0000u) : (w << 16));
                }
                v[c] = t;
            }
            if constexpr (ESC == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]), __float_as_uint(v[4 * j + 2]), __float_as_uint(v[4 * j + 3])},
                                                           rsC, (j < 2 ? oka : okb) ? offC + 16 * j : OOB, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { bf16x8 h; u32x4 u; } o;
#pragma unroll
                    for (int c = 0; c < 8; ++c) o.h[c] = (bf16_t)v[8 * j + c];
                    __builtin_amdgcn_raw_buffer_store_b128(o.u, rsC, (j == 0 ? oka : okb) ? offC + 16 * j : OOB, 0, 0);
                    if constexpr (STATS) {      // statistics of the values as stored
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[8 * j + c] = (float)o.h[c];
                    }
                }
            }
            if constexpr (STATS) {
                if (row_ok) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) { cs[c] += v[c]; if constexpr (SUMSQ) cq[c] += v[c] * v[c]; }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (wr == 1) new_slab_barrier(0);               // row half 1: one half-step behind
    for (int sl = 0; sl < ns; ++sl) {
        if (wr == 0) new_slab_barrier(sl); else plain_barrier();
        mfma_phase(sl);
        if (wr == 0) plain_barrier(); else new_slab_barrier(sl + 1);
        epilogue(sl);
    }
    if (wr == 0) plain_barrier();

    if constexpr (STATS) {
        // the 16 lanes of a DPP row share the column group and hold 16 different rows: four DPP steps leave the row total in every lane
        // of the row; lane c of the row adds column c's total -- once per kernel (the workgroup never changed its columns)
        float mine_s = 0.f, mine_q = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float sv = cs[c];
            sv += dpp_mov<0xB1>(sv); sv += dpp_mov<0x4E>(sv); sv += dpp_mov<0x124>(sv); sv += dpp_mov<0x128>(sv);
            mine_s = (i16 == c) ? sv : mine_s;
            if constexpr (SUMSQ) {
                float qv = cq[c];
                qv += dpp_mov<0xB1>(qv); qv += dpp_mov<0x4E>(qv); qv += dpp_mov<0x124>(qv); qv += dpp_mov<0x128>(qv);
                mine_q = (i16 == c) ? qv : mine_q;
            }
        }
        // two row halves of the workgroup share a column: through LDS (the ring is idle now), then one global atomic per column
        float* cacc = reinterpret_cast<float*>(smem);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < 2 * WS_BN) cacc[tid] = 0.f;
        __syncthreads();
        const int cl = wc * 64 + 16 * g + i16;
        atomicAdd(cacc + cl, mine_s);
        if constexpr (SUMSQ) atomicAdd(cacc + WS_BN + cl, mine_q);
        __syncthreads();
        if (tid < WS_BN && n0 + tid < p.N) {
            atomicAdd(p.colstats + n0 + tid, cacc[tid]);
            if constexpr (SUMSQ) atomicAdd(p.colstats + p.N + n0 + tid, cacc[WS_BN + tid]);
        }
    }
}

namespace {

template <typename TC, int EPI>
int launch_ws2(const FS2Gemm& g, hipStream_t st) {
    const int tilesN = (g.N + WS_BN - 1) / WS_BN;
    const int nslabs = (g.M + WS_ROWS - 1) / WS_ROWS;
    int ranges_per_xcd = 32 / tilesN;                                      // one workgroup per CU: 32 per XCD
    if (ranges_per_xcd < 1) ranges_per_xcd = 1;
    const int nranges = 8 * ranges_per_xcd;
    const int slabs_per_range = (nslabs + nranges - 1) / nranges;
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_ws_kernel<TC, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM) != hipSuccess) {
            fs2_set_error("fs2_gemm: cannot raise the dynamic LDS limit of the weights-stationary kernel");
            return FS2_ELAUNCH;
        }
    }
    const int grid = 8 * ranges_per_xcd * tilesN;
    hipLaunchKernelGGL((fs2_gemm_ws_kernel<TC, EPI>), dim3(grid), dim3(WS_NT), WS_SMEM, st, g, tilesN, ranges_per_xcd, slabs_per_range);
    FS2_CHECK_LAUNCH("fs2_gemm(ws)");
    return FS2_OK;
}

int ws_epi_code(const FS2Gemm& g) {
    const int res = g.residual == nullptr ? 0 : (g.res_dtype == FS2_F32 ? EPI_RES_F32 : EPI_RES_BF16);
    return (g.relu_mask ? EPI_MASK : 0) | res | (g.colstats ? (g.colstats_mode == 0 ? EPI_STATS | EPI_SUMSQ : EPI_STATS) : 0);
}

template <typename TC>
int launch_ws1(const FS2Gemm& g, hipStream_t st) {
    switch (ws_epi_code(g)) {
        case 0: return launch_ws2<TC, 0>(g, st);
        case EPI_MASK: return launch_ws2<TC, EPI_MASK>(g, st);
        case EPI_STATS: return launch_ws2<TC, EPI_STATS>(g, st);
        case EPI_MASK | EPI_STATS: return launch_ws2<TC, EPI_MASK | EPI_STATS>(g, st);
        case EPI_RES_F32: return launch_ws2<TC, EPI_RES_F32>(g, st);
        case EPI_RES_BF16: return launch_ws2<TC, EPI_RES_BF16>(g, st);
        default: break;
    }
    fs2_set_error("fs2_gemm(ws): epilogue combination not compiled");
    return FS2_EINVAL;
}

}  // namespace

// false: not eligible / not chosen; true: launched on the weights-stationary kernel, *rc holds the result
bool fs2_gemm_ws_try(const FS2Gemm& g, hipStream_t st, int* rc) {
    // FS2_GEMM_WS: 0 never, 1 (default) where the shape heuristic says so, 2 wherever eligible (tests, A/B measurements)
    const char* e1 = getenv("FS2_GEMM_WS");
    const int mode = e1 ? atoi(e1) : 1;
    if (mode == 0) return false;
    const bool pointwise = g.conv == 0 || (g.conv == 1 && g.taps == 1 && g.pad == 0);      // (a k = 1 convolution is a linear layer: decoder FFN)
    if (g.dtype != FS2_BF16 || g.a_kmajor || g.b_kmajor || !pointwise || g.accumulate || g.K != WS_K) return false;
    if ((long)g.batch1 * g.batch2 * g.split_k != 1) return false;
    if (g.N % 8 != 0) return false;
    const int epi = ws_epi_code(g);
    if (!(epi == 0 || epi == EPI_MASK || epi == EPI_STATS || epi == (EPI_MASK | EPI_STATS) || epi == EPI_RES_F32 || epi == EPI_RES_BF16)) return false;
    {   // 32-bit byte offsets (rows up to M + 63 enter the arithmetic)
        const long rows = (long)g.M + 128;
        if (rows * g.ldc * 4 >= 0x7FFFFFF0L || rows * g.lda * 2 >= 0x7FFFFFF0L || ((long)g.N + 256) * g.ldb * 2 >= 0x7FFFFFF0L) return false;
        if (g.relu_mask != nullptr && rows * g.ldm * 2 >= 0x7FFFFFF0L) return false;
        if (g.residual != nullptr && rows * g.ldr * 4 >= 0x7FFFFFF0L) return false;
    }
    const int tilesN = (g.N + WS_BN - 1) / WS_BN;
    if (tilesN > 32) return false;
    // every workgroup stages its 128 KiB of weights once (~8 us of fixed time with the first slabs): worth it from ~6 slabs per
    // workgroup on -- the decoder-side products with N >= 512 (44 k rows); N = 256 (2.7 slabs per workgroup) stays on the tiled kernel
    if (mode == 1 && (long)g.M * tilesN < 6L * 256 * WS_ROWS) return false;
    g_last_tile = 131;          // (measurement aid: the weights-stationary kernel)
    FS2Gemm g2 = g;
    {
        const char* e2 = getenv("FS2_WS_DBG");
        g2.tile_order = e2 ? atoi(e2) : 0;
    }
    *rc = g.c_dtype == FS2_F32 ? launch_ws1<float>(g2, st) : launch_ws1<bf16_t>(g2, st);
    return true;
}

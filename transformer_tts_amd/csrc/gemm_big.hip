// Large-tile MFMA GEMM for gfx950, round-2 form: the tall row-major x row-major products of the decoder side of the model
// (M = batch * frames ~ 44k rows; linear layers and implicit-GEMM Conv1d, forward and data gradient).
// Since round 3 the bf16 products run on gemm_ring.hip (same geometry, counted waits, bias in LDS, spill-free epilogues, sliced
// split-K); this file keeps the ONE-BYTE-OPERAND instances (fp8 e4m3 x e4m3, bf8 e5m2 x fp8 e4m3: hp.fp8, BASELINE configs[4]).
//
// Why a second kernel: the 128x128 / 4-wave stage loop of gemm.hip tops out at ~775 TFLOP/s with
// L2-resident operands; a 256x256 block tile worked on by 16 waves of 64x64 (<= 128 VGPRs, four waves per SIMD, one
// workgroup per CU) halves the staging traffic per MFMA and reaches ~1400 in the same micro-benchmark.
//
// Structure
//   * block tile BM x 256 with BM = 4 * WTM, WTM = 64 or 48 (BM = 256 or 192: chosen per launch so that the number of
//     tiles fills the 256 CUs best); 4 x 4 waves, each wave a WTM x 64 output tile = (WTM/16) x 4 MFMA 16x16x32 tiles;
//   * K in stages of 64 bf16 (128 B per row); two LDS stage buffers [A: BM rows][B: 256 rows] x 128 B, 16-B chunk c of
//     row r stored at c ^ f(r) (conflict-free ds_read_b128);
//   * staging by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  One wave-instruction
//     writes 1 KiB = 8 rows linearly, so the swizzle is applied on the per-lane SOURCE address (lane -> logical chunk
//     (lane&7) ^ f(row) of row lane>>3).  Rows outside the matrix, conv halo rows and the K tail use an out-of-range
//     buffer offset: the DMA writes zeros;
//   * operands are passed to the MFMA swapped (A-operand = weight rows, B-operand = activation rows) and MFMA tile jt of
//     a wave takes the weight rows 16*(i>>2) + 4*jt + (i&3), i = 0..15, of the wave's 64: lane (row i16, group g) then
//     holds the SIXTEEN CONSECUTIVE OUTPUT COLUMNS 16g .. 16g+15 of its output row in acc[it][0..3][0..3].  The epilogue
//     goes straight from the accumulators to 16-byte global stores, four lanes covering a full 128-byte line (bf16) --
//     no LDS round trip, no barriers, and the LDS stays free for the next tile's first stage, which is already in
//     flight while the epilogue runs (persistent workgroups walk a flattened stream of stages over all their tiles);
//   * tile walk: workgroups b and b+8 share an XCD (round-robin dispatch).  XCD x owns the row slabs mt = x (mod 8);
//     the column tiles of a slab are taken by different CUs of that XCD at the same time, so an A slab is fetched
//     from HBM once and hit in that XCD's L2 by the other column tiles (placement is a speed assumption only);
//   * fused epilogue: alpha, bias, ReLU, ReLU mask, residual (fp32 / bf16), fp32 / bf16 output, per-column statistics
//     (BatchNorm sums or bias-gradient column sums: DPP row reduction over the 16 lanes that share a column group,
//     LDS accumulators, one global atomic per column per workgroup).
#include "fs2_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr unsigned OOB = 0x80000000u;
constexpr int BIG_COLSTAT_N = 2048;      // columns whose statistics a workgroup accumulates in LDS (2 floats each)
constexpr int NW = 16, NT = 1024, BN = 256;

// epilogue variants compiled in (template bits): the operands a launch does not have cost nothing
constexpr int EPI_MASK = 1, EPI_RES_F32 = 2, EPI_RES_BF16 = 4, EPI_STATS = 8, EPI_SUMSQ = 16;   // STATS: column sums (+ SUMSQ: and sums of squares)

template <int WTM> struct BG {
    static constexpr int MT = WTM / 16, BM = 4 * WTM;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int AQ = BM / 8, BQ = BN / 8;               // LDS-DMA wave-instructions per stage (1 KiB = 8 rows each)
    static constexpr int AI = (AQ + NW - 1) / NW, BI = BQ / NW;   // ... per wave
    static constexpr int SMEM = 2 * STAGE;
};

// 16-B chunk swizzles: A rows are read 16 consecutive rows per fragment; weight rows 16*(i>>2) + 4*jt + (i&3)
__device__ __forceinline__ int fA(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int fB(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }

}  // namespace

extern thread_local int g_last_tile;     // gemm.hip

template <typename TC, int WTM, int EPI, bool STAMP, int OPK>
__global__ __launch_bounds__(1024, 4) void fs2_gemm_big_kernel(const FS2Gemm p, const int tilesM, const int tilesN, unsigned long long* dbg) {
    typedef BG<WTM> G;
    constexpr int MT = G::MT, BM = G::BM, AI = G::AI, BI = G::BI;
    constexpr int ESC = (int)sizeof(TC);
    // operands: OPK 0 = bf16 x bf16; 1 = fp8 e4m3 x e4m3; 2 = bf8 e5m2 (A: a gradient) x fp8 e4m3 (B: weights).  One byte per
    // element doubles the k extent of a 128-byte stage row (128 instead of 64) and of a 16-byte chunk (16 instead of 8).
    constexpr int ES = OPK == 0 ? 2 : 1, EPC = 16 / ES, BK = 128 / ES;
    constexpr bool HAS_MASK = (EPI & EPI_MASK) != 0, RES_F32 = (EPI & EPI_RES_F32) != 0, RES_BF16 = (EPI & EPI_RES_BF16) != 0;
    constexpr bool STATS = (EPI & EPI_STATS) != 0, SUMSQ = (EPI & EPI_SUMSQ) != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, i16 = lane & 15;

    // ---- work of this block: items j = slot, slot + nslots, ... of its XCD group's list (slab-major, column tile fastest)
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int slabs = tilesM > x ? (tilesM - x + 7) >> 3 : 0;
    const int items = slabs * tilesN;
    const int nmine = items > slot ? (items - slot + nslots - 1) / nslots : 0;
    if (nmine == 0) return;
    const bool conv = p.conv == 1;
    const int nkt = (p.K + BK - 1) / BK;
    const int ntot = (conv ? p.taps : 1) * nkt;
    const int nst = nmine * ntot;
    const int pad = conv ? p.pad : 0;
    const int lda = (int)p.lda, ldb = (int)p.ldb;

    float* cacc = reinterpret_cast<float*>(smem + G::SMEM);
    if constexpr (STATS) {
        for (int n = tid; n < 2 * BIG_COLSTAT_N; n += NT) cacc[n] = 0.f;      // visible after the prologue's barrier
    }

    // conv: the descriptor base is moved back by `pad` rows so that the scalar stage offset (tap*lda + kb) is never negative
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.A) - (int64_t)pad * lda * ES), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7FFFFFF0, 0x00020000);

    // ---- LDS-DMA source coordinates of this lane: instruction i of this wave covers tile rows 8*(i*NW + wave) .. +7;
    //      the lane fetches logical chunk (lane&7) ^ f(row) of row lane>>3 of those (swizzle on the source side)
    auto dma_row = [&](int i) { return 8 * (i * NW + wave) + (lane >> 3); };
    auto a_k8 = [&](int i) { return ((lane & 7) ^ fA(dma_row(i))) * EPC; };
    auto b_k8 = [&](int i) { return ((lane & 7) ^ fB(dma_row(i))) * EPC; };

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
            voffA[i] = (m < p.M) ? (unsigned)((m * lda + a_k8(i)) * ES) : OOB;
            tA[i] = conv ? (m % p.seq_len) - pad : 0;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int n = n0 + dma_row(i);
            voffB[i] = (n < p.N) ? (unsigned)((n * ldb + b_k8(i)) * ES) : OOB;
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        const int kb = lkb, tap = ltap;
        const int sA = (tap * lda + kb) * ES;
        const int sB = (tap * p.K + kb) * ES;
        unsigned char* base = smem + buf * G::STAGE + 1024 * wave;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            if (i * NW + wave < G::AQ) {          // wave-uniform (BM = 192: 24 instructions for 16 waves)
                bool ok = (voffA[i] != OOB) && (kb + a_k8(i) < p.K);
                if (conv) ok = ok && ((unsigned)(tA[i] + tap) < (unsigned)p.seq_len);
                // (voffset must be an int expression: an unsigned one makes the host-side instantiation of the kernel
                //  template fail silently -- no stub, undefined symbol when the library is loaded)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(base + 1024 * NW * i), 16, (int)(ok ? voffA[i] : OOB), sA, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const bool ok = (voffB[i] != OOB) && (kb + b_k8(i) < p.K);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(base + G::A_BYTES + 1024 * NW * i), 16, (int)(ok ? voffB[i] : OOB), sB, 0, 0);
        }
        lkb += BK;
        if (lkb >= p.K) { lkb = 0; ++ltap; }
        if (++lst == ntot) {
            lst = 0; ltap = 0; lkb = 0;
            lj += nslots;
            if (--lleft > 0) prep_item(lj);
        }
    };

    // ---- fragment read addresses (lane part; A tile it adds it*2048, weight tile jt adds jt*512, stage buffer b adds b*STAGE)
    // bf16: k-step ks (32 k = 64 B) of a row is chunks 4ks .. 4ks+3, lane group g takes chunk 4ks + g (8 elements);
    // fp8: k-step ks (32 k = 32 B) is chunks 2ks, 2ks+1, lane group g takes the 8 bytes (g&1) of chunk 2ks + (g>>1)
    constexpr int NKS = ES == 2 ? 2 : 4;
    int rdA[NKS], rdB[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int ra = wr * WTM + i16;
        const int rb = wc * 64 + 16 * (i16 >> 2) + (i16 & 3);
        const int ch = ES == 2 ? ks * 4 + g : 2 * ks + (g >> 1);
        const int sub = ES == 2 ? 0 : 8 * (g & 1);
        rdA[ks] = ra * 128 + ((ch ^ fA(ra)) << 4) + sub;
        rdB[ks] = G::A_BYTES + rb * 128 + ((ch ^ fB(rb)) << 4) + sub;      // fB does not depend on jt (bits 2,3 of the row)
    }

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- compute cursor
    int cj = slot, cst = 0;
    bool pending = false;
    int pj = 0;

    unsigned long long t_issue = 0, t_epi = 0, t_mma = 0, t_wait = 0, t_bar = 0, t_prev = 0;
    auto stamp = [&](unsigned long long& sum) __attribute__((always_inline)) {
        if constexpr (STAMP) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            sum += t - t_prev;
            t_prev = t;
        }
    };

    // fp8 operands: the per-tensor de-quantisation factors live on the device (written by fs2_quantize_fp8); read once per launch
    const float alpha = p.alpha * (p.scale_a != nullptr ? *p.scale_a : 1.f) * (p.scale_b != nullptr ? *p.scale_b : 1.f);
    prep_item(lj);
    issue(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (STAMP) { unsigned long long junk = 0; stamp(junk); }

    for (int s = 0; ; ++s) {
        const int buf = s & 1;
        // Epilogue operands that do not depend on the row loop are fetched BEFORE the next stage's DMA is issued: vmcnt
        // retires in order, so a load issued after the DMA could only be consumed once the DMA (HBM latency) has landed.
        const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)p.relu_mask, 0, p.relu_mask ? 0x7FFFFFF0 : 0, 0x00020000);
        u32x4 braw[4];
        u32x4 mraw[HAS_MASK ? MT : 1][2];
        int pq = 0, pnt = 0;
        if (pending) {
            pq = pj / tilesN; pnt = pj - pq * tilesN;
            const int mb = (x + 8 * pq) * BM + wr * WTM + i16;
            const int nb = pnt * BN + wc * 64 + 16 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j)       // absent bias / columns >= N: zero records -> zeros
                braw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsBias, (nb + 4 * j) * 4, 0, 0);
            if constexpr (HAS_MASK) {
                const unsigned offM = (unsigned)((mb * (int)p.ldm + nb) * 2);
#pragma unroll
                for (int it = 0; it < 2; ++it) {      // the first two row tiles' masks ahead of the DMA, the rest behind it (registers)
                    const bool row_ok = mb + 16 * it < p.M;
                    const int so = 16 * it * (int)p.ldm * 2;
                    mraw[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && nb < p.N) ? offM : OOB, so, 0);
                    mraw[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && nb + 8 < p.N) ? offM + 16 : OOB, so, 0);
                }
            }
        }
        // stage s+1 -> the buffer every wave finished reading at the last barrier.  The scheduling barrier keeps the DMA instructions
        // in front of everything else of the stage (hipcc otherwise threads them through the epilogue / fragment reads: dominant
        // kernel 31.4 -> 30.2 us per launch).  Letting half of the waves issue AFTER their MFMAs, which helps the weight-gradient
        // kernel (gemm_big_km.hip), costs 1 % here.
        if (s + 1 < nst) issue(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HAS_MASK) {
            if (pending) {
                const int mb = (x + 8 * pq) * BM + wr * WTM + i16;
                const int nb = pnt * BN + wc * 64 + 16 * g;
                const unsigned offM = (unsigned)((mb * (int)p.ldm + nb) * 2);
#pragma unroll
                for (int it = 2; it < MT; ++it) {
                    const bool row_ok = mb + 16 * it < p.M;
                    const int so = 16 * it * (int)p.ldm * 2;
                    mraw[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && nb < p.N) ? offM : OOB, so, 0);
                    mraw[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && nb + 8 < p.N) ? offM + 16 : OOB, so, 0);
                }
            }
        }
        stamp(t_issue);
        if (pending) {
            // ---- epilogue of item pj: lane holds C[m0 + wr*WTM + it*16 + i16][n0 + wc*64 + 16g + 4jt + r] in acc[it][jt][r]
            pending = false;
            const int mb = (x + 8 * pq) * BM + wr * WTM + i16;
            const int nb = pnt * BN + wc * 64 + 16 * g;
            const bool ok_lo = nb < p.N, ok_hi = nb + 8 < p.N;       // N is a multiple of 8
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFF0, 0x00020000);
            const unsigned offC = (unsigned)((mb * (int)p.ldc + nb) * ESC);
            float bias[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) bias[4 * j + r] = __uint_as_float(braw[j][r]);
            float cs[STATS ? 16 : 1], cq[SUMSQ ? 16 : 1];
            if constexpr (STATS) {
#pragma unroll
                for (int c = 0; c < 16; ++c) { cs[c] = 0.f; if constexpr (SUMSQ) cq[c] = 0.f; }
            }
#pragma unroll
            for (int it = 0; it < MT; ++it) {
                const bool row_ok = mb + 16 * it < p.M;
                const bool oka = row_ok && ok_lo, okb = row_ok && ok_hi;
                u32x4 rraw[4];
                if constexpr (RES_F32) {
                    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, 0x7FFFFFF0, 0x00020000);
                    const unsigned offR = (unsigned)((mb * (int)p.ldr + nb) * 4);
                    const int so = 16 * it * (int)p.ldr * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        rraw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (j < 2 ? oka : okb) ? offR + 16 * j : OOB, so, 0);
                }
                if constexpr (RES_BF16) {
                    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, 0x7FFFFFF0, 0x00020000);
                    const unsigned offR = (unsigned)((mb * (int)p.ldr + nb) * 2);
                    const int so = 16 * it * (int)p.ldr * 2;
                    rraw[0] = __builtin_amdgcn_raw_buffer_load_b128(rsR, oka ? offR : OOB, so, 0);
                    rraw[1] = __builtin_amdgcn_raw_buffer_load_b128(rsR, okb ? offR + 16 : OOB, so, 0);
                }
                float v[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    float t = acc[it][c >> 2][c & 3] * alpha + bias[c];
                    if (p.relu) t = fmaxf(t, 0.f);
                    if constexpr (HAS_MASK) {
                        const unsigned w = mraw[it][c >> 3][(c >> 1) & 3];
                        const float mk = __uint_as_float((c & 1) ? (w & 0xFFFF0000u) : (w << 16));
                        t = mk > 0.f ? t : 0.f;
                    }
                    if constexpr (RES_F32) t += __uint_as_float(rraw[c >> 2][c & 3]);
                    if constexpr (RES_BF16) {
                        const unsigned w = rraw[c >> 3][(c >> 1) & 3];
                        t += __uint_as_float((c & 1) ? (w & 0xFFFF0000u) : (w << 16));
                    }
                    v[c] = t;
                }
                // The row advance goes into the per-lane offset, NOT into the scalar soffset operand: hipcc (ROCm 7.2) pads the
                // "16-byte store followed by a VALU write of its data registers" hazard only when soffset is not a register,
                // and with an SGPR soffset the very next instruction did overwrite v[data] -- on the MI355X a few stores per
                // launch then wrote the clobbering value (the row index) instead of the result.
                const unsigned so = (unsigned)(16 * it * (int)p.ldc * ESC);
                if constexpr (ESC == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]), __float_as_uint(v[4 * j + 2]), __float_as_uint(v[4 * j + 3])},
                                                               rsC, (j < 2 ? oka : okb) ? offC + so + 16 * j : OOB, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        union { bf16x8 h; u32x4 u; } o;
#pragma unroll
                        for (int c = 0; c < 8; ++c) o.h[c] = (bf16_t)v[8 * j + c];
                        __builtin_amdgcn_raw_buffer_store_b128(o.u, rsC, (j == 0 ? oka : okb) ? offC + so + 16 * j : OOB, 0, 0);
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
            if constexpr (STATS) {
                // the 16 lanes of a DPP row share g (the column group) and hold 16 different rows: four DPP steps leave
                // the row total in every lane of the row; lane c of the row then adds column c's total
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
                const int n = nb + i16;
                if (n < p.N) {
                    atomicAdd(cacc + n, mine_s);
                    if constexpr (SUMSQ) atomicAdd(cacc + BIG_COLSTAT_N + n, mine_q);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            stamp(t_epi);
        }
        if (s == nst) break;
        const unsigned char* lb = smem + buf * G::STAGE;
        if constexpr (ES == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[MT], fb[4];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(lb + rdA[ks] + i * 2048);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(lb + rdB[ks] + j * 512);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D[n][m]
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                long fa[MT], fb[4];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const long*>(lb + rdA[ks] + i * 2048);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const long*>(lb + rdB[ks] + j * 512);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // swapped operands: MFMA A = weights (always e4m3), MFMA B = activations (e4m3) or gradients (e5m2)
                        if constexpr (OPK == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j], fa[i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (++cst == ntot) { pending = true; pj = cj; cst = 0; cj += nslots; }
        __builtin_amdgcn_sched_barrier(0);         // (and the stage's MFMAs in front of the wait: measured together with the one above)
        stamp(t_mma);
        // own DMA landed, own fragment reads retired; then every wave's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp(t_wait);
        asm volatile("s_barrier" ::: "memory");
        stamp(t_bar);
    }

    if constexpr (STATS) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave's LDS adds are done
        for (int n = tid; n < p.N; n += NT) {
            const float a = cacc[n];
            if (a != 0.f) atomicAdd(p.colstats + n, a);
            if constexpr (SUMSQ) {
                const float q = cacc[BIG_COLSTAT_N + n];
                if (q != 0.f) atomicAdd(p.colstats + p.N + n, q);
            }
        }
    }
    if constexpr (STAMP) {
        if (dbg != nullptr && lane == 0 && (wave == 0 || wave == 15)) {
            unsigned long long* o = dbg + (blockIdx.x * 2 + (wave == 0 ? 0 : 1)) * 8;
            o[0] = t_issue; o[1] = t_epi; o[2] = t_mma; o[3] = t_wait; o[4] = t_bar; o[5] = (unsigned long long)nst; o[6] = (unsigned long long)nmine;
        }
    }
}

namespace {

template <typename TC, int WTM, int EPI, bool STAMP, int OPK>
int launch_big2(const FS2Gemm& g, hipStream_t st) {
    typedef BG<WTM> G;
    const int tilesM = (g.M + G::BM - 1) / G::BM, tilesN = (g.N + BN - 1) / BN;
    const int lds = G::SMEM + ((EPI & EPI_STATS) ? 2 * BIG_COLSTAT_N * 4 : 0);
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_big_kernel<TC, WTM, EPI, STAMP, OPK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM + 2 * BIG_COLSTAT_N * 4) != hipSuccess) {
            fs2_set_error("fs2_gemm: cannot raise the dynamic LDS limit of the large-tile kernel");
            return FS2_ELAUNCH;
        }
    }
    const long per_xcd = (long)((tilesM + 7) / 8) * tilesN;
    const int grid = 8 * (int)(per_xcd < 32 ? per_xcd : 32);
    hipLaunchKernelGGL((fs2_gemm_big_kernel<TC, WTM, EPI, STAMP, OPK>), dim3(grid), dim3(NT), lds, st, g, tilesM, tilesN, (unsigned long long*)nullptr);
    FS2_CHECK_LAUNCH("fs2_gemm(big)");
    return FS2_OK;
}

template <typename TC, int WTM, int OPK>
int launch_big1(const FS2Gemm& g, hipStream_t st) {
    const int res = g.residual == nullptr ? 0 : (g.res_dtype == FS2_F32 ? EPI_RES_F32 : EPI_RES_BF16);
    const int epi = (g.relu_mask ? EPI_MASK : 0) | res | (g.colstats ? (g.colstats_mode == 0 ? EPI_STATS | EPI_SUMSQ : EPI_STATS) : 0);
    switch (epi) {      // the combinations the model uses; anything else stays on the 128-tile kernel (checked by the caller)
        case 0: return launch_big2<TC, WTM, 0, false, OPK>(g, st);
        case EPI_MASK: return launch_big2<TC, WTM, EPI_MASK, false, OPK>(g, st);
        case EPI_STATS: return launch_big2<TC, WTM, EPI_STATS, false, OPK>(g, st);
        case EPI_STATS | EPI_SUMSQ: return launch_big2<TC, WTM, EPI_STATS | EPI_SUMSQ, false, OPK>(g, st);
        case EPI_MASK | EPI_STATS: return launch_big2<TC, WTM, EPI_MASK | EPI_STATS, false, OPK>(g, st);
        case EPI_RES_F32: return launch_big2<TC, WTM, EPI_RES_F32, false, OPK>(g, st);
        case EPI_RES_BF16: return launch_big2<TC, WTM, EPI_RES_BF16, false, OPK>(g, st);
        case EPI_MASK | EPI_RES_F32: return launch_big2<TC, WTM, EPI_MASK | EPI_RES_F32, false, OPK>(g, st);
        default: break;
    }
    fs2_set_error("fs2_gemm(big): epilogue combination %d not compiled", epi);
    return FS2_EINVAL;
}

bool epi_compiled(const FS2Gemm& g) {
    const int res = g.residual == nullptr ? 0 : (g.res_dtype == FS2_F32 ? EPI_RES_F32 : EPI_RES_BF16);
    const int epi = (g.relu_mask ? EPI_MASK : 0) | res | (g.colstats ? (g.colstats_mode == 0 ? EPI_STATS | EPI_SUMSQ : EPI_STATS) : 0);
    return epi == (EPI_STATS | EPI_SUMSQ) || epi == 0 || epi == EPI_MASK || epi == EPI_STATS || epi == (EPI_MASK | EPI_STATS) || epi == EPI_RES_F32 ||
           epi == EPI_RES_BF16 || epi == (EPI_MASK | EPI_RES_F32);
}

}  // namespace

// false: not eligible; true: the product was launched on the large-tile kernel and *rc holds the result (fp8 operands only)
bool fs2_gemm_big_try(const FS2Gemm& g, hipStream_t st, int* rc) {
    const bool fp8 = g.dtype == FS2_FP8 || g.dtype == FS2_BF8_FP8;
    if (!fp8 || g.a_kmajor || g.b_kmajor || g.accumulate || g.conv > 1) return false;
    if (g.K % 16 != 0 || g.lda % 16 != 0 || g.ldb % 16 != 0) return false;
    if ((long)g.batch1 * g.batch2 * g.split_k != 1) return false;
    if (g.N % 8 != 0 || (g.colstats != nullptr && g.N > BIG_COLSTAT_N)) return false;
    if (g.conv == 1 && (g.pad < 0 || g.pad > g.taps)) return false;
    if (!epi_compiled(g)) return false;
    {   // the epilogue addresses C / mask / residual with 32-bit byte offsets (rows up to M + 255 enter the arithmetic)
        const long rows = (long)g.M + 512;
        if (rows * g.ldc * 4 >= 0x7FFFFFF0L) return false;
        if (g.relu_mask != nullptr && rows * g.ldm * 2 >= 0x7FFFFFF0L) return false;
        if (g.residual != nullptr && rows * g.ldr * 4 >= 0x7FFFFFF0L) return false;
        if (rows * g.lda * 2 >= 0x7FFFFFF0L || ((long)g.N + 512) * g.ldb * 2 >= 0x7FFFFFF0L) return false;
    }
    const bool f32 = g.c_dtype == FS2_F32;
    g_last_tile = 192;          // one-byte operands: the 192-row tile only (keeps the number of kernel instances down)
    if (g.dtype == FS2_FP8) *rc = f32 ? launch_big1<float, 48, 1>(g, st) : launch_big1<bf16_t, 48, 1>(g, st);
    else *rc = f32 ? launch_big1<float, 48, 2>(g, st) : launch_big1<bf16_t, 48, 2>(g, st);
    return true;
}

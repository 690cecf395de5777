// Row-major x row-major bf16 MFMA GEMM for gfx950, 16 waves: the tall products of the model (linear layers and implicit-GEMM
// Conv1d, forward and data gradient; M = batch * frames rows against N <= 2048 output channels).  Round 3 form of the
// large-tile kernel of round 2 (retired in round 4).
//
// Geometry (unchanged, it fits the model): BM x 256 block tile, BM = 128 / 192 / 256 chosen per launch, 16 waves of
// (BM/4) x 64, <= 128 VGPRs (four waves per SIMD), one persistent workgroup per CU walking a flattened stream of k-slots over all
// its work items; operands swapped on the MFMA (weights as the A operand, weight rows of a wave permuted over its four column
// tiles) so that a lane holds 16 consecutive output columns of one row: 16-byte stores straight from the accumulators.
//   * K is consumed in SLOTS of 64 bf16 = 128 B per row; one LDS-DMA wave-instruction (buffer_load_dwordx4 ... lds) covers
//     8 rows x 128 B: whole cache lines.  (A first round-3 version used 32-deep slots = 16 rows x 64 B per instruction to fit a
//     5-deep ring: bit-identical and 15-30 % SLOWER -- half-line pieces double the texture-addresser work per byte, exactly the
//     "fragment-shaped loads" the CDNA4 guide warns about.  Measured, dropped.)
//   * the LDS holds a RING of S slots: S = 3 for the 128-row tile (48 KiB each), 2 for the 192 / 256-row tiles; 16-byte chunk c of
//     row r sits at c ^ f(r) (conflict-free ds_read_b128), the swizzle applied on the per-lane SOURCE address of the DMA;
//   * waits are COUNTED: `s_waitcnt vmcnt(N)` with N = the vector-memory operations this wave has issued since the awaited pieces
//     (gfx9 retires loads, stores and LDS-DMA in issue order).  The 16-byte stores of an epilogue are therefore not waited for at
//     the next slot barrier (the round-2 kernel's vmcnt(0) drained them there);
//   * bias lives in LDS (ds_read_b128 in the epilogue): a plain / bias / ReLU epilogue contains no vector-memory load, so nothing
//     in the slot stream waits on the in-order queue; mask / residual rows (two of the model's eight product classes) are buffer
//     loads one row tile ahead inside the epilogue (all row tiles up front cost 24-48 registers: the round-2 instances with a mask
//     spilled up to 208 B per lane; every instance of this kernel the model uses has zero scratch: tools/check_resources.py);
//   * sliced split-K for products with few output tiles and a long reduction (the 6144 x 256 x (9 x 1024) encoder convolutions):
//     every split stores its fp32 partial tile with plain 16-byte stores into its own slice of a workspace; fs2_splitk_reduce sums
//     the slices and applies bias / ReLU / residual / cast (no float atomics: ~1.3 TB/s chip-wide against ~6 TB/s of plain stores).
// Results are bit-identical to the 128-tile kernel of gemm.hip (same MFMA, same k order per accumulator).
#pragma once
#include "fs2_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr unsigned OOB = 0x80000000u;
constexpr int NW = 16, NT = 1024, BN = 256;               // waves, threads, tile columns (k per slot: 128 B = 64 bf16 or 128 fp8)
constexpr int LDS_MAX = 160 * 1024;

constexpr int EPI_MASK = 1, EPI_RES_F32 = 2, EPI_RES_BF16 = 4, EPI_STATS = 8, EPI_SUMSQ = 16, EPI_Q8 = 32;

template <int WTM> struct RG {
    static constexpr int MT = WTM / 16, BM = 4 * WTM;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SLOT = A_BYTES + B_BYTES;
    static constexpr int AQ = BM / 8, BQ = BN / 8;                // LDS-DMA pieces per slot (1 KiB = 8 rows x 128 B each)
    static constexpr int AI = (AQ + NW - 1) / NW, BI = BQ / NW;   // ... per wave (piece i of wave w covers rows 8*(i*NW + w) .. +7)
    static constexpr int SMAX = WTM == 32 ? 3 : 2;
};

template <int N> struct IC { static constexpr int value = N; };

// Division by the launch's invariants (work-item decoding, k-split bounds, the frame index of a Conv1d row) without the ~25-instruction
// sequences an integer division compiles into: the host passes, for every divisor d, M = ceil(2^(31+l) / d) and the shift 31 + l with
// l = ceil(log2 d); (n * M) >> shift = n / d exactly for 0 <= n < 2^31 (the error n * (M d - 2^(31+l)) / (d 2^(31+l)) is below 2^-l <= 1/d).
// The kernel's prologue (every wave runs it before the first LDS-DMA request leaves: ~600 instructions, 8 divisions) is on the
// critical path of every launch.
struct RingMagic { unsigned m_nslots, s_nslots, m_splits, s_splits, m_tn, s_tn, m_tns, s_tns, m_nkt, s_nkt, m_seq, s_seq; };
__device__ __forceinline__ int fdiv(int n, unsigned M, unsigned sh) { return (int)(((unsigned long long)(unsigned)n * M) >> sh); }

// 16-B chunk swizzles: A rows are read 16 consecutive rows per fragment; weight rows 16*(i>>2) + 4*jt + (i&3)
__device__ __forceinline__ int fA(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int fB(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }

// wait until at most n of this wave's vector-memory operations are outstanding, and for its LDS reads; then the workgroup barrier.
// s_waitcnt takes an immediate: n is rounded DOWN to one of six counts (waiting for a few operations more than necessary is always
// right) -- a switch over every count compiled into a tree of ~10 taken branches and ~25 scalar instructions on the path of the hot
// counts, and every scalar instruction between a slot barrier and the first MFMA behind it costs its full latency: all waves of the
// workgroup run this section at the same moment, nothing overlaps it (measured: 64 extra s_add per slot = +0.34 us per slot,
// profiles/r04_ring_probe.txt)
__device__ __forceinline__ void wait_barrier(int n) {
#define FS2_W(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // (the steady-state counts first: 0 with a 2-deep ring, 3 or 4 = the pieces of one younger slot with a 3-deep ring)
    if (__builtin_expect(n < 3, 1)) { FS2_W(0); return; }
    if (__builtin_expect(n == 3, 1)) { FS2_W(3); return; }
    if (n < 6) { FS2_W(4); return; }
    if (n < 8) { FS2_W(6); return; }
    if (n < 12) { FS2_W(8); return; }
    FS2_W(12);
#undef FS2_W
}

}  // namespace

extern thread_local int g_last_tile;     // gemm.hip
extern thread_local int g_last_splits;

// ES: bytes per operand element -- 2: bf16 (v_mfma_f32_16x16x32_bf16, two k-steps per slot); 1: OCP fp8 (e4m3 weights; e4m3 or
// e5m2 activations / gradients by p.dtype), one block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 per slot with unit block scales (the
// per-tensor scales multiply alpha): twice the bf16 MFMA rate, half the staged and LDS-read bytes per multiply-add.
template <typename TC, int WTM, int EPI, int ES = 2>
__global__ __launch_bounds__(1024, 4) void fs2_gemm_ring_kernel(const FS2Gemm p, const int tilesM, const int tilesN, const int splits,
                                                               const int S, const int stats_off, const int bias_off, const RingMagic mg) {
    typedef RG<WTM> G;
    constexpr int MT = G::MT, BM = G::BM, SLOT = G::SLOT;
    constexpr int SK = 128 / ES, CE = 16 / ES;                      // k elements per slot / per 16-byte chunk
    constexpr int ESC = (int)sizeof(TC);
    constexpr bool HAS_MASK = (EPI & EPI_MASK) != 0, RES_F32 = (EPI & EPI_RES_F32) != 0, RES_BF16 = (EPI & EPI_RES_BF16) != 0;
    constexpr bool STATS = (EPI & EPI_STATS) != 0, SUMSQ = (EPI & EPI_SUMSQ) != 0;
    constexpr bool Q8 = (EPI & EPI_Q8) != 0;          // the epilogue also writes an fp8 copy of C (FS2Gemm.q8; bf16 C only)
    constexpr bool EPI_LOADS = HAS_MASK || RES_F32 || RES_BF16;      // the epilogue issues vector-memory loads of its own
    constexpr int NSTORES = MT * ((ESC == 4 ? 4 : 2) + (Q8 ? 1 : 0));  // 16-byte stores of one epilogue, per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, i16 = lane & 15;

    // ---- work items of this workgroup: XCD group x = blockIdx & 7 owns the row slabs mt = x (mod 8); its items are numbered
    //      (slab, k-split, column tile) with the column tile fastest, so the workgroups of one XCD that run at the same time share
    //      a slab's k-range of A in that XCD's L2; workgroup `slot` of the group takes items slot, slot + nslots, ...
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int slabs = tilesM > x ? (tilesM - x + 7) >> 3 : 0;
    const int items = slabs * tilesN * splits;
    const int nmine = items > slot ? fdiv(items - slot + nslots - 1, mg.m_nslots, mg.s_nslots) : 0;
    if (nmine == 0) return;
    const bool conv = p.conv == 1;
    const int nkt = (p.K + SK - 1) / SK;
    const int ntot = (conv ? p.taps : 1) * nkt;                 // slots of a whole reduction
    const int per = fdiv(ntot + splits - 1, mg.m_splits, mg.s_splits);      // slots per split (the host made every split non-empty)
    const int pad = conv ? p.pad : 0;
    const int lda = (int)p.lda, ldb = (int)p.ldb;
    const int ring_bytes = S * SLOT;
    const bool plain = !conv && (p.K % SK == 0);                // no per-slot validity arithmetic
    int nst = 0;                                                // slots this workgroup runs through
    if (splits == 1) nst = nmine * ntot;
    else
        for (int j = slot; j < items; j += nslots) {
            const int jt = fdiv(j, mg.m_tn, mg.s_tn);
            const int sp = jt - fdiv(jt, mg.m_splits, mg.s_splits) * splits;
            nst += min(ntot, sp * per + per) - sp * per;
        }

    // ---- LDS carve: [ring S x SLOT][column statistics][bias]
    float* cacc = reinterpret_cast<float*>(smem + stats_off);
    const float* biasl = reinterpret_cast<const float*>(smem + bias_off);
    const int ncols = tilesN * BN;
    const bool has_bias = p.bias != nullptr;                    // without a bias the LDS area is 16 zero floats, read by every lane

    // conv: the descriptor base is moved back by `pad` rows so that the scalar slot offset (tap*lda + kb) is never negative
    // (timing-only switches, FS2_RING_DBG -> p.tile_order: 1 = the output descriptor has zero records (stores dropped), 2 = the activation
    //  descriptor, 4 = the weight descriptor (zeros staged): prices one buffer's traffic with the instruction stream unchanged; results wrong)
    const int dbg = ES == 2 ? p.tile_order : 0;        // (bf16 instances only: the fp8 192-row instances have no register to spare)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.A) - (int64_t)pad * lda * ES), 0, (dbg & 2) ? 0 : 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (dbg & 4) ? 0 : 0x7FFFFFF0, 0x00020000);

    // ---- LDS-DMA source coordinates of this lane: piece i of this wave covers tile rows 8*(i*NW + wave) .. +7; the lane fetches
    //      logical chunk (lane&7) ^ f(row) of row lane>>3 of those (swizzle on the source side)
    auto dma_row = [&](int i) { return 8 * (i * NW + wave) + (lane >> 3); };
    auto a_k8 = [&](int i) { return ((lane & 7) ^ fA(dma_row(i))) * CE; };
    auto b_k8 = [&](int i) { return ((lane & 7) ^ fB(dma_row(i))) * CE; };
    int pw = G::BI;                                                 // LDS-DMA instructions of this wave per slot (wave-uniform)
#pragma unroll
    for (int i = 0; i < G::AI; ++i) pw += (i * NW + wave < G::AQ) ? 1 : 0;

    // ---- load cursor
    int lj = slot, lsl = 0, lend = 0, ltap = 0, lkb = 0, lleft = nmine, rp_i = 0;
    unsigned voffA[G::AI], voffB[G::BI];
    int tA[G::AI];
    auto prep_item = [&](int j) {
        const int q = fdiv(j, mg.m_tns, mg.s_tns), rem = j - q * (tilesN * splits);
        const int sp = fdiv(rem, mg.m_tn, mg.s_tn), nt = rem - sp * tilesN;
        const int m0 = (x + 8 * q) * BM, n0 = nt * BN;
#pragma unroll
        for (int i = 0; i < G::AI; ++i) {
            const int m = m0 + dma_row(i);
            voffA[i] = (m < p.M) ? (unsigned)((m * lda + a_k8(i)) * ES) : OOB;
            tA[i] = conv ? m - fdiv(m, mg.m_seq, mg.s_seq) * p.seq_len - pad : 0;
        }
#pragma unroll
        for (int i = 0; i < G::BI; ++i) {
            const int n = n0 + dma_row(i);
            voffB[i] = (n < p.N) ? (unsigned)((n * ldb + b_k8(i)) * ES) : OOB;
        }
        const int s0 = sp * per;
        lend = min(ntot, s0 + per) - s0;
        ltap = fdiv(s0, mg.m_nkt, mg.s_nkt);
        lkb = (s0 - ltap * nkt) * SK;
    };
    // (as few instructions as possible: this runs between a slot barrier and the MFMAs behind it.  The per-lane offsets already say OOB
    //  for rows outside the matrix; only the last k-slot of a row whose K is not a multiple of the slot, and the taps of a convolution
    //  at the ends of an utterance, need a per-lane decision: two instantiations, chosen by a workgroup-uniform branch)
    // MODE 1: plain (no convolution, K a multiple of the slot); 2: convolution with K a multiple of the slot (the taps' frame test is
    // the only per-lane decision: three vector instructions per activation piece); 0: general (K tail as well)
    auto issue_pieces = [&](auto MODEC) __attribute__((always_inline)) {
        constexpr int MODE = decltype(MODEC)::value;
        const int kb = lkb, tap = ltap;
        const int sA = MODE == 1 ? kb * ES : (tap * lda + kb) * ES;
        const int sB = MODE == 1 ? kb * ES : (tap * p.K + kb) * ES;
        unsigned char* base = smem + rp_i + 1024 * wave;
        const bool ktail = MODE == 0 && kb + SK > p.K;             // (workgroup-uniform)
#pragma unroll
        for (int i = 0; i < G::AI; ++i) {
            if (i * NW + wave < G::AQ) {          // wave-uniform (BM = 192: 24 pieces for 16 waves)
                unsigned v = voffA[i];
                if constexpr (MODE == 2) v = ((unsigned)(tA[i] + tap) < (unsigned)p.seq_len) ? v : OOB;
                if constexpr (MODE == 0) {
                    if (conv) v = ((unsigned)(tA[i] + tap) < (unsigned)p.seq_len) ? v : OOB;
                    if (ktail) v = (kb + a_k8(i) < p.K) ? v : OOB;
                }
                // (voffset must be an int expression: an unsigned one makes the host-side instantiation of the kernel template fail
                //  silently -- no stub, undefined symbol when the library is loaded)
                // (FS2_RING_NT_A: measurement build -- the activation rows, which one workgroup reads once, carry the nontemporal hint)
#ifdef FS2_RING_NT_A
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(base + 1024 * NW * i), 16, (int)v, sA, 0, 2);
#else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(base + 1024 * NW * i), 16, (int)v, sA, 0, 0);
#endif
            }
        }
#pragma unroll
        for (int i = 0; i < G::BI; ++i) {
            unsigned v = voffB[i];
            if constexpr (MODE == 0) {
                if (ktail) v = (kb + b_k8(i) < p.K) ? v : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(base + G::A_BYTES + 1024 * NW * i), 16, (int)v, sB, 0, 0);
        }
    };
    const int issue_mode = plain ? 1 : (conv && p.K % SK == 0) ? 2 : 0;
    auto issue_dma = [&]() __attribute__((always_inline)) {
        if (issue_mode == 1) issue_pieces(IC<1>{});
        else if (issue_mode == 2) issue_pieces(IC<2>{});
        else issue_pieces(IC<0>{});
    };
    auto advance_cursor = [&]() __attribute__((always_inline)) {
        lkb += SK;
        if (lkb >= p.K) { lkb = 0; ++ltap; }
        if (++lsl == lend) {
            lsl = 0;
            lj += nslots;
            if (--lleft > 0) prep_item(lj);
        }
        rp_i += SLOT;
        if (rp_i == ring_bytes) rp_i = 0;
    };
    auto issue = [&]() __attribute__((always_inline)) { issue_dma(); advance_cursor(); };

    // ---- fragment read addresses (lane part; A row tile `it` adds it*2048, weight tile jt adds jt*512, ring position rp_c):
    //      bf16: k-step ks (32 k = 64 B) of a row is chunks 4ks .. 4ks+3, lane group g takes chunk 4ks + g (8 elements);
    //      fp8: the 32 k-bytes of lane group g in the 128-deep MFMA are chunks g and g + 4 (any assignment of the slot's k to the
    //      lane groups is right as long as both operands use it; this one keeps ds_read_b128 conflict-free under the same swizzle)
    int rdA[2], rdB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ra = wr * WTM + i16;
        const int rb = wc * 64 + 16 * (i16 >> 2) + (i16 & 3);
        const int ch = ks * 4 + g;
        rdA[ks] = ra * 128 + ((ch ^ fA(ra)) << 4);
        rdB[ks] = G::A_BYTES + rb * 128 + ((ch ^ fB(rb)) << 4);      // fB does not depend on jt (bits 2,3 of the row)
    }

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- compute cursor
    int cj = slot, csl = 0, cend = 0, rp_c = 0;
    int cm0 = 0, cn0 = 0, csp = 0;
    auto decode_c = [&](int j) {
        const int q = fdiv(j, mg.m_tns, mg.s_tns), rem = j - q * (tilesN * splits);
        csp = fdiv(rem, mg.m_tn, mg.s_tn);
        const int nt = rem - csp * tilesN;
        cm0 = (x + 8 * q) * BM; cn0 = nt * BN;
        const int s0 = csp * per;
        cend = min(ntot, s0 + per) - s0;
    };
    decode_c(cj);

    const float alpha = ES == 1 ? p.alpha * (p.scale_a != nullptr ? *p.scale_a : 1.f) * (p.scale_b != nullptr ? *p.scale_b : 1.f) : p.alpha;
    const bool act_e5m2 = p.dtype == FS2_BF8_FP8;                   // (fp8 form) activations / gradients in e5m2, weights always e4m3
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (dbg & 1) ? 0 : 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)p.relu_mask, 0, p.relu_mask ? 0x7FFFFFF0 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? 0x7FFFFFF0 : 0, 0x00020000);
    // fp8 copy of C: speculative scale (the amax of this tensor one step ago), amax of the values as stored for the repair launch
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(Q8 ? p.q8 : nullptr, 0, Q8 ? 0x7FFFFFF0 : 0, 0x00020000);
    typedef __attribute__((ext_vector_type(2))) unsigned short us16x2;
    float qscale = 1.f, qfmax = 448.f;
    us16x2 qmax_pk = {0, 0};
    if constexpr (Q8) {
        float inv;
        qscale = fs2_pow2_scale(p.q8_prev[0], p.q8_bf8 ? 15 : 8, &inv);
        qfmax = p.q8_bf8 ? 57344.f : 448.f;
    }

    // ---- epilogue of the item (cm0, cn0, csp): lane holds C[cm0 + wr*WTM + it*16 + i16][cn0 + wc*64 + 16g + 4jt + r] in acc[it][jt][r]
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int mb = cm0 + wr * WTM + i16;
        const int nb = cn0 + wc * 64 + 16 * g;
        const bool ok_lo = nb < p.N, ok_hi = nb + 8 < p.N;       // N is a multiple of 8
        // split-K: slice csp of the fp32 workspace (slice stride sC1 elements)
        const unsigned offC = (unsigned)((mb * (int)p.ldc + nb) * ESC) + (unsigned)csp * (unsigned)((int)p.sC1 * ESC);
        float cs[STATS ? 16 : 1], cq[SUMSQ ? 16 : 1];
        if constexpr (STATS) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { cs[c] = 0.f; if constexpr (SUMSQ) cq[c] = 0.f; }
        }
        // operand rows of the first row tile; the rows of tile it+1 are requested before tile it is finished (one tile of lookahead:
        // all MT tiles up front cost 24-48 registers and spilled in the round-2 kernel)
        u32x4 mraw[2], rraw[RES_F32 ? 4 : 2];
        auto fetch = [&](int it) __attribute__((always_inline)) {
            const bool row_ok = mb + 16 * it < p.M;
            if constexpr (HAS_MASK) {
                const unsigned offM = (unsigned)(((mb + 16 * it) * (int)p.ldm + nb) * 2);
                mraw[0] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && ok_lo) ? offM : OOB, 0, 0);
                mraw[1] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (row_ok && ok_hi) ? offM + 16 : OOB, 0, 0);
            }
            if constexpr (RES_F32) {
                const unsigned offR = (unsigned)(((mb + 16 * it) * (int)p.ldr + nb) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rraw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (row_ok && (j < 2 ? ok_lo : ok_hi)) ? offR + 16 * j : OOB, 0, 0);
            }
            if constexpr (RES_BF16) {
                const unsigned offR = (unsigned)(((mb + 16 * it) * (int)p.ldr + nb) * 2);
                rraw[0] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (row_ok && ok_lo) ? offR : OOB, 0, 0);
                rraw[1] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (row_ok && ok_hi) ? offR + 16 : OOB, 0, 0);
            }
        };
        // (one tile of lookahead where the registers allow it: the 256-row tile and the mask + fp32-residual form request a tile's rows
        //  right before they use them instead -- with the lookahead those instances spilled 7-19 registers)
        constexpr bool LOOK = WTM < 64 && !(HAS_MASK && RES_F32);
        if constexpr (EPI_LOADS && LOOK) fetch(0);
#pragma unroll
        for (int it = 0; it < MT; ++it) {
            const bool row_ok = mb + 16 * it < p.M;
            const bool oka = row_ok && ok_lo, okb = row_ok && ok_hi;
            u32x4 mcur[2], rcur[RES_F32 ? 4 : 2];
            if constexpr (EPI_LOADS) {
                if constexpr (!LOOK) fetch(it);
                mcur[0] = mraw[0]; mcur[1] = mraw[1];
#pragma unroll
                for (int j = 0; j < (RES_F32 ? 4 : 2); ++j) rcur[j] = rraw[j];
                if constexpr (LOOK) { if (it + 1 < MT) fetch(it + 1); }
            }
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(biasl + (has_bias ? nb + 4 * j : 0));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * j + r;
                    float t = acc[it][j][r] * alpha + b4[r];
                    if (p.relu) t = fmaxf(t, 0.f);
                    if constexpr (HAS_MASK) {
                        const unsigned w = mcur[c >> 3][(c >> 1) & 3];
                        const float mk = __uint_as_float((c & 1) ? (w & 0xFFFF0000u) : (w << 16));
                        t = mk > 0.f ? t : 0.f;
                    }
                    if constexpr (RES_F32) t += __uint_as_float(rcur[c >> 2][c & 3]);
                    if constexpr (RES_BF16) {
                        const unsigned w = rcur[c >> 3][(c >> 1) & 3];
                        t += __uint_as_float((c & 1) ? (w & 0xFFFF0000u) : (w << 16));
                    }
                    v[c] = t;
                }
            }
            // (the row advance goes into the per-lane offset, not into an SGPR soffset: hipcc pads the "16-byte store, then VALU
            //  write of its data registers" hazard only for an immediate soffset -- DESIGN.md, "A store hazard found on the way")
            const unsigned so = (unsigned)(16 * it * (int)p.ldc * ESC);
            if constexpr (ESC == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]), __float_as_uint(v[4 * j + 2]), __float_as_uint(v[4 * j + 3])},
                                                           rsC, (j < 2 ? oka : okb) ? offC + so + 16 * j : OOB, 0, 0);
            } else {
                u32x4 q8w;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { bf16x8 h; u32x4 u; } o;
#pragma unroll
                    for (int c = 0; c < 8; ++c) o.h[c] = (bf16_t)v[8 * j + c];
                    __builtin_amdgcn_raw_buffer_store_b128(o.u, rsC, (j == 0 ? oka : okb) ? offC + so + 16 * j : OOB, 0, 0);
                    if constexpr (STATS || Q8) {      // statistics / fp8 codes of the values as stored
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[8 * j + c] = (float)o.h[c];
                    }
                    if constexpr (Q8) {
                        // amax of the stored bf16 values on their packed bit patterns (non-negative bf16 order like their 15-bit codes)
                        if (j == 0 ? oka : okb) {
#pragma unroll
                            for (int k4 = 0; k4 < 4; ++k4) {
                                union { unsigned u; us16x2 h; } m;
                                m.u = o.u[k4] & 0x7FFF7FFFu;
                                qmax_pk = __builtin_elementwise_max(qmax_pk, m.h);
                            }
                        }
                        int w0 = 0, w1 = 0;
                        auto cv = [&](int c) { return fminf(fmaxf(v[8 * j + c] * qscale, -qfmax), qfmax); };
                        if (p.q8_bf8) {
                            w0 = __builtin_amdgcn_cvt_pk_bf8_f32(cv(0), cv(1), w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(cv(2), cv(3), w0, true);
                            w1 = __builtin_amdgcn_cvt_pk_bf8_f32(cv(4), cv(5), w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(cv(6), cv(7), w1, true);
                        } else {
                            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cv(0), cv(1), w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cv(2), cv(3), w0, true);
                            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cv(4), cv(5), w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cv(6), cv(7), w1, true);
                        }
                        q8w[2 * j] = (unsigned)w0; q8w[2 * j + 1] = (unsigned)w1;
                    }
                }
                if constexpr (Q8) {       // 16 codes = the lane's 16 consecutive columns of this row (N is a multiple of 16 here)
                    const unsigned offQ = (unsigned)((mb + 16 * it) * (int)p.ldc + nb);
                    __builtin_amdgcn_raw_buffer_store_b128(q8w, rsQ, oka ? offQ : OOB, 0, 0);
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
            // the 16 lanes of a DPP row share g (the column group) and hold 16 different rows: four DPP steps leave the row total
            // in every lane of the row; lane c of the row then adds column c's total to the workgroup's LDS accumulator
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
                if constexpr (SUMSQ) atomicAdd(cacc + ncols + n, mine_q);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- prologue: S - 1 slots in flight
    prep_item(lj);
    const int ahead = S - 1 < nst ? S - 1 : nst;
    for (int i = 0; i < ahead; ++i) issue();
    // bias and statistics accumulators go to LDS BEHIND the first slots' requests: the bias fetch overlaps the DMA round trip instead of
    // preceding it (one memory latency per launch, not two); their first reader is an epilogue, i.e. behind the first slot barrier, which
    // every thread reaches with its own LDS writes retired (lgkmcnt(0))
    if constexpr (STATS) {
        for (int n = tid; n < (SUMSQ ? 2 : 1) * ncols; n += NT) cacc[n] = 0.f;
    }
    for (int n = tid; n < (has_bias ? ncols : 16); n += NT)
        reinterpret_cast<float*>(smem + bias_off)[n] = (has_bias && n < p.N) ? p.bias[n] : 0.f;

    // Stores issued by the last epilogue count towards vmcnt like the LDS-DMA pieces; `since_epi` iterations ago this wave issued
    // NSTORES of them.  They are younger than the piece awaited at iteration t when the epilogue ran in iterations t-W .. t-1, with
    // W = S - 1 when the slot issue of an item-end iteration precedes its epilogue, S - 2 when it follows it (EPI_LOADS).
    int since_epi = 1 << 20;
    constexpr int W_OFF = EPI_LOADS ? 2 : 1;
    // Order inside a slot (bf16): barrier -> LDS reads of the first k-step -> LDS-DMA requests of slot t+S-1 -> MFMAs -> bookkeeping.
    // Every wave of the workgroup leaves the barrier at the same moment and nothing overlaps what a wave does before its first MFMA:
    // each scalar instruction there costs the whole workgroup its latency (measured: 64 extra s_add per slot = +0.34 us per slot; slot
    // time = matrix time + ~0.4 us at every tile height before this order, profiles/r04_ring_probe.txt).  So the wait count of the NEXT
    // barrier, the load cursor and the ring pointers are updated BEHIND the MFMAs (the matrix pipes are still draining then), and the
    // DMA requests sit in the shadow of the fragment reads' latency.
    auto next_wait = [&](int t1) __attribute__((always_inline)) {
        const int younger = (S - 2 < nst - 1 - t1) ? S - 2 : nst - 1 - t1;        // slots issued after slot t1 by the time it is awaited
        return younger * pw + (since_epi <= S - W_OFF ? NSTORES : 0);
    };
    int n_wait = next_wait(0);

    constexpr bool LEGACY = ES == 2 && MT == 3 && HAS_MASK && STATS;     // (the one instance whose registers do not allow the new order)
    for (int t = 0; t < nst; ++t) {
        if constexpr (LEGACY) n_wait = next_wait(t);
        wait_barrier(n_wait);
        const bool tile_end = (csl + 1 == cend);
        const bool more = t + S - 1 < nst;
        // slot t+S-1 -> the ring position every wave finished reading before the barrier it has just passed
        const bool do_issue = !LEGACY && more && !(EPI_LOADS && tile_end);
        if constexpr (ES == 1) {
            if (do_issue) issue_dma();
            __builtin_amdgcn_sched_barrier(0);
            typedef __attribute__((ext_vector_type(8))) int i32x8;
            const unsigned char* lb = smem + rp_c;
            union F8 { struct { u32x4 lo, hi; } s; i32x8 v; };
            // the MT activation fragments stay (8 registers each), the four weight fragments stream through one register set
            F8 fa[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                fa[i].s.lo = *reinterpret_cast<const u32x4*>(lb + rdA[0] + i * 2048);
                fa[i].s.hi = *reinterpret_cast<const u32x4*>(lb + rdA[1] + i * 2048);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                F8 fb;
                fb.s.lo = *reinterpret_cast<const u32x4*>(lb + rdB[0] + j * 512);
                fb.s.hi = *reinterpret_cast<const u32x4*>(lb + rdB[1] + j * 512);
                // (weights as the A operand: cbsz = 0, e4m3; blgp = the activation format; block scales 2^0 = E8M0 127)
                if (act_e5m2) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb.v, fa[i].v, acc[i][j], 0, 1, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                } else {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb.v, fa[i].v, acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
        } else {
            // (the instance with ReLU mask + column sums at 192 rows has no register to spare: requests first, plain fragment order)
            constexpr bool TIGHT = MT >= 4 || (MT == 3 && HAS_MASK && STATS);
            const unsigned char* lb = smem + rp_c;
            bf16x8 fa[MT], fb[4];
            if constexpr (TIGHT && MT == 3) {
                if (more && !(EPI_LOADS && tile_end)) issue();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(lb + rdA[0] + i * 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(lb + rdB[0] + j * 512);
            if constexpr (!(TIGHT && MT == 3)) {
                __builtin_amdgcn_sched_barrier(0);
                if (do_issue) issue_dma();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!TIGHT) {
                // the weight fragments of the second k-step are requested before the first k-step's MFMAs, its activation fragments
                // into the registers of the first k-step's as those retire: the second k-step starts without an LDS round trip in
                // front of it (the 256-row tile has no registers for the second weight set: plain order there)
                bf16x8 fb2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb2[j] = *reinterpret_cast<const bf16x8*>(lb + rdB[1] + j * 512);
                __builtin_amdgcn_sched_barrier(0);        // (hipcc otherwise sinks these reads to just in front of their first use)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D[n][m]: weights as the A operand
                    fa[i] = *reinterpret_cast<const bf16x8*>(lb + rdA[1] + i * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb2[j], fa[i], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(lb + rdA[1] + i * 2048);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(lb + rdB[1] + j * 512);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LEGACY) {
            if (do_issue) advance_cursor();
        }
        rp_c += SLOT;
        if (rp_c == ring_bytes) rp_c = 0;
        ++since_epi;
        if (tile_end) {
            epilogue();
            since_epi = 1;
            csl = 0;
            cj += nslots;
            if (t + 1 < nst) decode_c(cj);
            if constexpr (EPI_LOADS) {
                if (more) issue();
            }
        } else {
            ++csl;
        }
        if constexpr (!LEGACY) n_wait = next_wait(t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (Q8) {       // one atomic per workgroup on the tensor's amax word (16 floats of LDS behind the bias area)
        float* q8wm = reinterpret_cast<float*>(smem + bias_off + (has_bias ? ncols * 4 : 64));
        const unsigned short mb16 = qmax_pk[0] > qmax_pk[1] ? qmax_pk[0] : qmax_pk[1];
        float qamax = __uint_as_float((unsigned)mb16 << 16);
        if (!(qamax == qamax)) qamax = 0.f;                      // (a NaN in the tensor does not define its range)
        qamax = wave_max(qamax);
        if (lane == 0) q8wm[wave] = qamax;
        __syncthreads();
        if (tid == 0) {
            float b = q8wm[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) b = fmaxf(b, q8wm[w]);
            if (b > 0.f) atomicMax(reinterpret_cast<unsigned*>(p.q8_state), __float_as_uint(b));
        }
    }
    if constexpr (STATS) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave's LDS adds are done
        for (int n = tid; n < p.N; n += NT) {
            const float a = cacc[n];
            if (a != 0.f) atomicAdd(p.colstats + n, a);
            if constexpr (SUMSQ) {
                const float q = cacc[ncols + n];
                if (q != 0.f) atomicAdd(p.colstats + p.N + n, q);
            }
        }
    }
}

namespace {

template <typename TC, int WTM, int EPI, int ES = 2>
int launch_ring2(const FS2Gemm& g, int splits, hipStream_t st) {
    typedef RG<WTM> G;
    const int tilesM = (g.M + G::BM - 1) / G::BM, tilesN = (g.N + BN - 1) / BN;
    const int ncols = tilesN * BN;
    const int stats_bytes = (EPI & EPI_STATS) ? ((EPI & EPI_SUMSQ) ? 2 : 1) * ncols * 4 : 0;
    const int bias_bytes = g.bias != nullptr ? ncols * 4 : 64;
    int S = (LDS_MAX - stats_bytes - bias_bytes - 64) / G::SLOT;
    if (S > G::SMAX) S = G::SMAX;
    {   // FS2_RING_S: ring depth override (measurements)
        static const int s_env = getenv("FS2_RING_S") ? atoi(getenv("FS2_RING_S")) : 0;
        if (s_env >= 2 && s_env < S) S = s_env;
    }
    if (S < 2) { fs2_set_error("fs2_gemm(ring): N=%d leaves no room for the slot ring", g.N); return FS2_EINVAL; }
    const int stats_off = S * G::SLOT, bias_off = stats_off + stats_bytes;
    const int lds = bias_off + bias_bytes + ((EPI & EPI_Q8) ? 64 : 0);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[16] = {};            // > 64 KiB of dynamic LDS must be allowed once per kernel and device
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_ring_kernel<TC, WTM, EPI, ES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_MAX) != hipSuccess) {
            fs2_set_error("fs2_gemm: cannot raise the dynamic LDS limit of the ring kernel");
            return FS2_ELAUNCH;
        }
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    const long per_xcd = (long)((tilesM + 7) / 8) * tilesN * splits;
    const int grid = 8 * (int)(per_xcd < 32 ? per_xcd : 32);
    auto magic = [](unsigned d, unsigned& M, unsigned& sh) {
        if (d < 1) d = 1;
        int l = 0;
        while ((1u << l) < d) ++l;
        M = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
        sh = 31 + l;
    };
    RingMagic mg;
    const int SKh = 128 / ES;
    magic((unsigned)(grid >> 3), mg.m_nslots, mg.s_nslots);
    magic((unsigned)splits, mg.m_splits, mg.s_splits);
    magic((unsigned)tilesN, mg.m_tn, mg.s_tn);
    magic((unsigned)(tilesN * splits), mg.m_tns, mg.s_tns);
    magic((unsigned)((g.K + SKh - 1) / SKh), mg.m_nkt, mg.s_nkt);
    magic((unsigned)(g.conv == 1 ? g.seq_len : 1), mg.m_seq, mg.s_seq);
    hipLaunchKernelGGL((fs2_gemm_ring_kernel<TC, WTM, EPI, ES>), dim3(grid), dim3(NT), lds, st, g, tilesM, tilesN, splits, S, stats_off, bias_off, mg);
    FS2_CHECK_LAUNCH("fs2_gemm(ring)");
    return FS2_OK;
}

int epi_code(const FS2Gemm& g) {
    const int res = g.residual == nullptr ? 0 : (g.res_dtype == FS2_F32 ? EPI_RES_F32 : EPI_RES_BF16);
    return (g.relu_mask ? EPI_MASK : 0) | res | (g.colstats ? (g.colstats_mode == 0 ? EPI_STATS | EPI_SUMSQ : EPI_STATS) : 0) | (g.q8 ? EPI_Q8 : 0);
}

bool epi_compiled(int epi) {
    if (epi == EPI_Q8 || epi == (EPI_MASK | EPI_STATS | EPI_Q8)) return true;       // (fp8 operands + bf16 C only: checked by the caller)
    return epi == 0 || epi == EPI_MASK || epi == EPI_STATS || epi == (EPI_STATS | EPI_SUMSQ) || epi == (EPI_MASK | EPI_STATS) || epi == EPI_RES_F32 ||
           epi == EPI_RES_BF16 || epi == (EPI_MASK | EPI_RES_F32);
}

template <typename TC, int WTM, int ES = 2>
int launch_ring1(const FS2Gemm& g, int splits, hipStream_t st) {
    switch (epi_code(g)) {      // the combinations the model uses; anything else stays on the older kernels (checked by the caller)
        case 0: return launch_ring2<TC, WTM, 0, ES>(g, splits, st);
        // (fp8 form: the mask / statistics epilogues exist for the 128-row tile only -- beside the 32-byte fragments of the 192-row tile
        //  they spill 8-44 B per lane; gemm_ring_f8.hip routes them)
        case EPI_MASK:
            if constexpr (ES == 2 || WTM == 32) return launch_ring2<TC, WTM, EPI_MASK, ES>(g, splits, st);
            break;
        case EPI_STATS:
            if constexpr (ES == 2 || WTM == 32) return launch_ring2<TC, WTM, EPI_STATS, ES>(g, splits, st);
            break;
        case EPI_STATS | EPI_SUMSQ:
            if constexpr (WTM < 64 && (ES == 2 || WTM == 32)) return launch_ring2<TC, WTM, EPI_STATS | EPI_SUMSQ, ES>(g, splits, st);       // (256-row tile: 10-19 spilled registers,
            break;                                                                                            //  fs2_gemm_ring_try picks 192 rows)
        case EPI_MASK | EPI_STATS:
            if constexpr (WTM < 64 && (ES == 2 || WTM == 32)) return launch_ring2<TC, WTM, EPI_MASK | EPI_STATS, ES>(g, splits, st);
            break;
        case EPI_RES_F32: return launch_ring2<TC, WTM, EPI_RES_F32, ES>(g, splits, st);
        case EPI_RES_BF16: return launch_ring2<TC, WTM, EPI_RES_BF16, ES>(g, splits, st);
        case EPI_MASK | EPI_RES_F32:
            if constexpr (WTM < 64 && (ES == 2 || WTM == 32)) return launch_ring2<TC, WTM, EPI_MASK | EPI_RES_F32, ES>(g, splits, st);
            break;
        case EPI_Q8:
            if constexpr (ES == 1 && sizeof(TC) == 2 && WTM == 32) return launch_ring2<TC, WTM, EPI_Q8, ES>(g, splits, st);     // (192 rows: 16 B of scratch)
            break;
        case EPI_MASK | EPI_STATS | EPI_Q8:
            if constexpr (ES == 1 && sizeof(TC) == 2 && WTM == 32) return launch_ring2<TC, WTM, EPI_MASK | EPI_STATS | EPI_Q8, ES>(g, splits, st);
            break;
        default: break;
    }
    fs2_set_error("fs2_gemm(ring): epilogue combination not compiled");
    return FS2_EINVAL;
}

}  // namespace


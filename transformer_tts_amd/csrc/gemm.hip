// MFMA GEMM family for gfx950: linear / implicit-GEMM Conv1d / batched attention products, forward,
// dgrad and wgrad, in bf16 (v_mfma_f32_16x16x32_bf16) or exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Block = 256 threads = 4 waves (2x2).  Block tile TM x TM (128, or 64 for problems with few tiles),
// wave tile TM/2 x TM/2 = (TM/32)^2 MFMA tiles of 16x16.  K is consumed in stages of 128 BYTES per row
// (64 bf16 / 32 f32): both dtypes share one LDS geometry and one 16-byte fragment read per lane per
// 64-byte k-step; only the MFMA issue differs (1 x 16x16x32 bf16, or 4 x 16x16x4 f32 whose k-slots are
// the 4 floats of the same 16 bytes).
//
// PERSISTENT blocks: the grid is min(work items, 2 x 256 CUs); a block walks work items (output tile x
// batch x k-split) with stride gridDim.x and issues the global loads of the NEXT item's first stage before
// the epilogue of the current one, so neither the load latency at the head of a tile nor the store tail is
// exposed -- most products of this model have only 2-16 k-stages per tile.
//
// LDS images (128*TM bytes per operand per stage, double buffered; the epilogue reuses them):
//   row-major operand  [TM rows][128 B], 16-B chunk c of row r stored at c ^ ((r>>1)&7)
//                      -> the 16 rows of a ds_read_b128 lane group fall on 16 distinct 16-B slots;
//   k-major bf16       [64 k][256 B], chunk c of row r at c ^ (((r&3)<<2)|((r>>2)&3)) and read with
//                      ds_read_b64_tr_b16 (hardware transpose; image (b) of the CDNA4 guide, T10);
//   k-major f32        [32 k][512 B], plain, read with ds_read_b32.          (k-major: TM = 128 only)
// Staging is global -> registers -> LDS through a ring of D register sets (D = 2 for the 128 tile, 4 for the
// 64 tile): the block walks ONE flattened stream of stages over all its work items, the loads of stage s+D are
// issued before the MFMAs of stage s (also across tile boundaries), and hipcc counts vmcnt for the set that is
// moved to LDS.  Loads are raw buffer loads: rows outside the matrix / conv halo rows get an out-of-range
// offset and read as zeros (no branches around loads -- a predicated `if (ok) v = *p` serialises them).
#include <type_traits>
#include "fs2_common.h"
#include <stdlib.h>

bool fs2_gemm_ws_try(const FS2Gemm& g, hipStream_t st, int* rc);      // gemm_ws.hip
bool fs2_gemm_ring_try(const FS2Gemm& g, hipStream_t st, int* rc);    // gemm_ring.hip
bool fs2_gemm_big_km_try(const FS2Gemm& g, hipStream_t st, int* rc);  // gemm_big_km.hip

thread_local int g_last_tile = 0;     // rows of the block tile of the last fs2_gemm launch of this thread (measurement aid)
thread_local int g_last_splits = 1;   // slices written by the last sliced split-K launch (accumulate = 2) of this thread
extern "C" int fs2_gemm_last_tile(void) { return g_last_tile; }
extern "C" int fs2_gemm_last_splits(void) { return g_last_splits; }

namespace {

template <typename T, bool KM> struct Tile {
    static constexpr int EPC = 16 / (int)sizeof(T);                 // elements per 16-B chunk
    static constexpr int BK = 128 / (int)sizeof(T);                 // k elements per stage
    static constexpr int CPR = KM ? (128 * (int)sizeof(T)) / 16 : 8;  // chunks per LDS row
    __device__ static __forceinline__ int lds_off(int row, int ch) {
        if constexpr (!KM) return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
        else if constexpr (sizeof(T) == 2) return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
        else return row * 512 + (ch << 4);
    }
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

// fragment of a 16-wide tile (rows/cols o0..o0+15 of the operand) for k-step ks (64 bytes of k)
template <typename T, bool KM>
__device__ __forceinline__ typename Frag<T>::type read_frag(const unsigned char* lds, int o0, int ks, int lane) {
    typedef Tile<T, KM> TL;
    const int g = lane >> 4, i16 = lane & 15;
    if constexpr (!KM) {
        const int row = o0 + i16;
        return *reinterpret_cast<const typename Frag<T>::type*>(lds + TL::lds_off(row, ks * 4 + g));
    } else if constexpr (sizeof(T) == 2) {
        // k rows ks*32 + 8g + {0..3} and {4..7}; lane 4q+p of the 16-lane group addresses row q, columns 4p..4p+3
        const int q = i16 >> 2, pp = i16 & 3;
        const int ch = (o0 >> 3) + (pp >> 1);
        const int r0 = ks * 32 + 8 * g + q;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + TL::lds_off(r0, ch) + 8 * (pp & 1)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + TL::lds_off(r0 + 4, ch) + 8 * (pp & 1)));
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
        u.s.lo = lo; u.s.hi = hi;
        return u.v;
    } else {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = *reinterpret_cast<const float*>(lds + (ks * 16 + 4 * g + e) * 512 + (o0 + i16) * 4);
        return v;
    }
}

template <typename T>
__device__ __forceinline__ void mma(f32x4& acc, const typename Frag<T>::type& a, const typename Frag<T>::type& b) {
    if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
}

constexpr int COLSTAT_LDS_N = 1024;     // columns whose statistics a block accumulates in LDS (2 floats each)

template <int TM> struct Geo {
    static constexpr int WT = TM / 2;                         // wave tile
    static constexpr int NI = WT / 16;                        // MFMA tiles per wave per dim
    static constexpr int STAGE = 128 * TM;                    // bytes per operand per stage
    static constexpr int NCH = STAGE / 16 / 256;              // 16-B chunks per thread per operand per stage
    static constexpr int EPI_LD = WT + 4;                     // floats per epilogue row (+4 pad: conflict-free)
    static constexpr int EPI_BYTES = 4 * WT * EPI_LD * 4;
    static constexpr int SMEM = (4 * STAGE > EPI_BYTES) ? 4 * STAGE : EPI_BYTES;
};

// one work item: output tile (m0,n0) of batch (b1,b2), k-stages [it0,it1)
struct Work {
    int m0, n0, split, b1, b2, it0, it1;
    int64_t aoff, boff, coff;
};

template <int N> struct IC { static constexpr int value = N; };

template <typename T, typename TC, bool AKM, bool BKM, int TM, bool DIRECT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const FS2Gemm p, const int total_work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Tile<T, AKM> TA;
    typedef Tile<T, BKM> TB;
    typedef Geo<TM> G;
    constexpr int EPC = TA::EPC, BK = TA::BK, NI = G::NI, NCH = G::NCH, WT = G::WT, EPI_LD = G::EPI_LD;
    constexpr int D = (TM == 128) ? 2 : 4;     // stages in flight in registers (prefetch depth)
    static_assert(TM == 128 || (!AKM && !BKM), "k-major operands use the 128 tile only");
    static_assert(!DIRECT, "a register-direct epilogue was tried (partial-line stores, accumulators in scratch) and dropped");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // Column statistics (BatchNorm sums / bias gradients) are accumulated per block in LDS and flushed with one
    // global atomic per touched column when the block retires: a global float atomic issued from the epilogue sits
    // on the in-order vmcnt queue in front of the next stages' loads, and ~700 wave-tiles add into each column.
    float* cacc = reinterpret_cast<float*>(smem + G::SMEM);
    const bool lds_stats = p.colstats != nullptr && p.N <= COLSTAT_LDS_N;
    if (lds_stats) {
        for (int n = tid; n < 2 * COLSTAT_LDS_N; n += 256) cacc[n] = 0.f;     // visible after the prologue's barrier
    }

    const int tilesN = (p.N + TM - 1) / TM, tilesM = (p.M + TM - 1) / TM;
    const int tiles = tilesM * tilesN;
    const int Kb = p.Kb > 0 ? p.Kb : p.K;
    const int nkt = (p.K + BK - 1) / BK;                       // k tiles per tap
    const int ntot = (p.conv == 1 ? p.taps : 1) * nkt;         // stages in the whole reduction
    const int per = (ntot + p.split_k - 1) / p.split_k;        // the host guarantees no empty split
    const int seq = p.seq_len;

    // Work items are numbered tile-fastest (n, then m -- or m, then n: tile_order 2), then k-split, then batch; a block
    // owns the CONTIGUOUS range [wbeg, wend): neighbouring tiles share operand panels in L2, and stepping to the next
    // item is a few scalar increments instead of six integer divisions.  m-fastest: the grid is a multiple of
    // 8 * tilesN, so block L and block L + grid/tilesN walk the same row slabs of A, in step, on the same XCD
    // (blockIdx % 8): the slab is fetched from HBM once and hit in that XCD's L2 by the other column panels.
    const bool mfast = p.tile_order == 2;
    const int wbeg = (int)((int64_t)blockIdx.x * total_work / gridDim.x);
    const int wend = (int)((int64_t)(blockIdx.x + 1) * total_work / gridDim.x);
    auto set_batch = [&](Work& k) {
        k.aoff = k.b1 * p.sA1 + k.b2 * p.sA2; k.boff = k.b1 * p.sB1 + k.b2 * p.sB2; k.coff = k.b1 * p.sC1 + k.b2 * p.sC2;
        k.it0 = k.split * per; k.it1 = min(ntot, k.it0 + per);
    };
    auto first_work = [&](int w) {
        Work k;
        const int lin = w % tiles;
        int z = w / tiles;
        k.split = z % p.split_k; z /= p.split_k;
        k.b2 = z % p.batch2;
        k.b1 = z / p.batch2;
        if (mfast) { k.n0 = (lin / tilesM) * TM; k.m0 = (lin % tilesM) * TM; }
        else { k.m0 = (lin / tilesN) * TM; k.n0 = (lin % tilesN) * TM; }
        set_batch(k);
        return k;
    };
    auto next_work = [&](Work& k) {
        bool wrap;
        if (mfast) {
            k.m0 += TM;
            wrap = false;
            if (k.m0 >= p.M) { k.m0 = 0; k.n0 += TM; if (k.n0 >= p.N) { k.n0 = 0; wrap = true; } }
        } else {
            k.n0 += TM;
            wrap = false;
            if (k.n0 >= p.N) { k.n0 = 0; k.m0 += TM; if (k.m0 >= p.M) { k.m0 = 0; wrap = true; } }
        }
        if (wrap) {
            if (++k.split == p.split_k) { k.split = 0; if (++k.b2 == p.batch2) { k.b2 = 0; ++k.b1; } }
            set_batch(k);
        }
    };

    // ---- per-thread, tile-independent staging coordinates and LDS addresses (hoisted out of every loop)
    int a_row[NCH], a_ch[NCH], b_row[NCH], b_ch[NCH], wrA[NCH], wrB[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i;
        a_row[i] = c / TA::CPR; a_ch[i] = c % TA::CPR;
        b_row[i] = c / TB::CPR; b_ch[i] = c % TB::CPR;
        wrA[i] = TA::lds_off(a_row[i], a_ch[i]);
        int brow = b_row[i];
        wrB[i] = TB::lds_off(brow, b_ch[i]);
    }
    int rdA[2], rdB[2];     // fragment read addresses of row-major operands: one lane offset per k-step, tile i adds i*2048
    {
        const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            rdA[ks] = AKM ? 0 : TA::lds_off(wr * WT + i16, ks * 4 + g);
            rdB[ks] = BKM ? 0 : TB::lds_off(wc * WT + i16, ks * 4 + g);
        }
    }

    // Staging: branch-free raw buffer loads through a (base, 2 GiB) descriptor; a lane whose chunk lies outside the
    // matrix / the sequence (conv halo) uses the out-of-range voffset 0x80000000 and the hardware returns zeros.
    // D stages are kept in flight in D register sets; the compiler counts vmcnt for the set being stored to LDS.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ES = (int)sizeof(T);
    const int lda = (int)p.lda, ldb = (int)p.ldb;
    u32x4 ra[D][NCH], rb[D][NCH];

    // ---- LOAD cursor: the next stage to fetch, possibly several stages / one work item ahead of the compute cursor
    int lw = wbeg;
    if (lw >= wend) return;
    Work lk = first_work(lw);
    int lit = lk.it0;
    int ltap, lkb;                     // tap and k offset of the next stage to load (advanced incrementally)
    unsigned offA[NCH], offB[NCH];     // voffset of the chunk at k = 0, tap shift 0 of work lk (OOB if the row is outside)
    int tA[NCH];                       // conv: position of the row inside its sequence
    __amdgpu_buffer_rsrc_t rsA, rsB;   // buffer descriptors of work lk (rebuilt only when the batch changes)
    const bool plain = (p.conv == 0) && (p.K % BK == 0) && (Kb == p.K);   // no per-stage validity: zero VALU per load
    auto prep_work = [&](const Work& k) {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const T*>(p.A) + k.aoff), 0, 0x7FFFFFF0, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const T*>(p.B) + k.boff), 0, 0x7FFFFFF0, 0x00020000);
        ltap = (p.conv == 1) ? k.it0 / nkt : 0;
        lkb = (k.it0 - ltap * nkt) * BK;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if constexpr (!AKM) {
                const int m = k.m0 + a_row[i];
                offA[i] = (m < p.M) ? (unsigned)((m * lda + a_ch[i] * EPC) * ES) : OOB;
                tA[i] = (p.conv == 1) ? (m % seq) : 0;
            } else {
                const int m = k.m0 + a_ch[i] * EPC;
                offA[i] = (m < p.M) ? (unsigned)((a_row[i] * lda + m) * ES) : OOB;
                tA[i] = 0;
            }
            if constexpr (!BKM) {
                const int n = k.n0 + b_row[i];
                offB[i] = (n < p.N) ? (unsigned)((n * ldb + b_ch[i] * EPC) * ES) : OOB;
            } else {
                const int n = k.n0 + b_ch[i] * EPC;
                offB[i] = (n < p.N) ? (unsigned)((b_row[i] * ldb + n) * ES) : OOB;
            }
        }
    };
    auto issue_load = [&](auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        // Always issued, also past the block's last stage (every offset is OOB then: zeros, no memory traffic): a
        // conditionally issued load makes hipcc wait vmcnt(0) wherever a ring slot is consumed, because on the path
        // without the issue nothing younger is outstanding -- and vmcnt(0) drains the whole ring at every stage.
        const int kb = lkb, tap = ltap;
        if (plain) {
            // wave-uniform, non-negative byte advance of this stage -> the scalar offset operand; the per-lane offsets
            // (row validity already folded in as OOB) are untouched: no VALU at all
            const int sA = AKM ? kb * lda * ES : kb * ES;
            const int sB = BKM ? kb * ldb * ES : kb * ES;
#pragma unroll
            for (int i = 0; i < NCH; ++i) ra[SLOT][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[i], sA, 0);
#pragma unroll
            for (int i = 0; i < NCH; ++i) rb[SLOT][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB[i], sB, 0);
        } else {
            const int shiftA = (p.conv == 1) ? tap - p.pad : 0;
            const int shiftB = (p.conv == 2) ? (lk.b2 - p.pad) : 0;
            // byte advance of this stage; may be negative for conv taps left of the centre -> added per lane
            const int sA = AKM ? kb * lda * ES : (shiftA * lda + kb) * ES;
            const int sB = BKM ? (kb + shiftB) * ldb * ES : (tap * p.K + kb) * ES;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                bool ok = offA[i] != OOB;
                if constexpr (!AKM) {
                    ok = ok && (kb + a_ch[i] * EPC < p.K);
                    if (p.conv == 1) { const int tt = tA[i] + shiftA; ok = ok && (tt >= 0) && (tt < seq); }
                } else {
                    ok = ok && (kb + a_row[i] < p.K);
                }
                ra[SLOT][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? offA[i] + (unsigned)sA : OOB, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                bool ok = offB[i] != OOB;
                if constexpr (!BKM) {
                    ok = ok && (kb + b_ch[i] * EPC < p.K);
                } else {
                    const int kk = kb + b_row[i];
                    ok = ok && (kk < Kb);
                    if (p.conv == 2) { const int tt = (kk % seq) + shiftB; ok = ok && (tt >= 0) && (tt < seq); }
                }
                rb[SLOT][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? offB[i] + (unsigned)sB : OOB, 0, 0);
            }
        }
        // advance the load cursor (no divisions)
        lkb += BK;
        if (lkb >= p.K) { lkb = 0; ++ltap; }
        if (++lit == lk.it1) {
            if (++lw < wend) { next_work(lk); prep_work(lk); lit = lk.it0; }
            else {
#pragma unroll
                for (int i = 0; i < NCH; ++i) { offA[i] = OOB; offB[i] = OOB; }
                lit = -0x40000000;      // never reaches it1 again
            }
        }
    };
    auto store_stage = [&](auto slot_c, auto buf_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int BUF = decltype(buf_c)::value;      // compile-time: the LDS addresses are lane offset + immediate
        unsigned char* la = smem + BUF * 2 * G::STAGE;
        unsigned char* lb = la + G::STAGE;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *reinterpret_cast<u32x4*>(la + wrA[i]) = ra[SLOT][i];
            *reinterpret_cast<u32x4*>(lb + wrB[i]) = rb[SLOT][i];
        }
    };

    // number of stages this block will run through (all its work items)
    int nst = 0;
    {
        int sp = (wbeg / tiles) % p.split_k, lin = wbeg % tiles;
        for (int w = wbeg; w < wend; ++w) {
            nst += min(ntot, sp * per + per) - sp * per;
            if (++lin == tiles) { lin = 0; if (++sp == p.split_k) sp = 0; }
        }
    }
    prep_work(lk);

    // ---- COMPUTE cursor
    int cw = wbeg;
    Work ck = lk;
    int cit = ck.it0;
    f32x4 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto epilogue = [&]() __attribute__((always_inline)) {
        const Work& cur = ck;
        TC* __restrict__ C = reinterpret_cast<TC*>(p.C) + cur.coff;
        {
            // ---- staged epilogue (k-major B, or fp32 atomic accumulate): accumulators -> wave-private LDS rows.
            // The stage that follows is still in registers, so the LDS image is free between the two barriers.
            float* ew = reinterpret_cast<float*>(smem) + wave * WT * EPI_LD;
            lds_barrier();            // every wave has finished reading the stage buffers
            {
                const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ew[(i * 16 + g * 4 + r) * EPI_LD + j * 16 + i16] = acc[i][j][r];
            }
            lds_barrier();
            bool stored = false;
            if constexpr (sizeof(TC) == 4) {
                if (p.accumulate) {
                    // one output row per wave-instruction, consecutive lanes on consecutive floats: every atomic
                    // instruction covers contiguous bytes (the full-rate shape)
                    for (int c0 = 0; c0 < WT; c0 += 64) {
                        const int cl = c0 + lane;
                        const int n = cur.n0 + wc * WT + cl;
                        if (cl < WT && n < p.N) {
                            float bias1 = 0.f;
                            if (p.bias != nullptr && cur.split == 0) bias1 = p.bias[n];
                            for (int r0 = 0; r0 < WT; ++r0) {
                                const int rl = (r0 + 4 * cur.split) & (WT - 1);   // splits of one tile walk its rows out of phase
                                const int m = cur.m0 + wr * WT + rl;
                                if (m >= p.M) continue;
                                atomicAdd(reinterpret_cast<float*>(C) + (int64_t)m * p.ldc + n, ew[rl * EPI_LD + cl] * p.alpha + bias1);
                            }
                        }
                    }
                    stored = true;
                }
            }
            if (!stored) {
                constexpr int CG = WT / 4;                      // 4-column groups per wave-tile row
                constexpr int RPP = 64 / CG;                    // rows per pass
                constexpr int NP = WT / RPP;                    // passes
                constexpr int PC = (AKM || BKM || NP < 8) ? 4 : 8;   // passes per chunk: their mask / residual loads fly together
                const int cg = lane % CG;
                const int ncol = cur.n0 + wc * WT + cg * 4;
                const bool col_ok = ncol < p.N;
                float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias != nullptr && col_ok) bias4 = *reinterpret_cast<const float4*>(p.bias + ncol);
                float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool has_mask = p.relu_mask != nullptr, has_res = p.residual != nullptr;
                const bool res_f32 = p.res_dtype == FS2_F32;
#pragma unroll
                for (int pc = 0; pc < NP; pc += PC) {
                    // The relu-mask and residual rows come from HBM: issue the loads of PC passes back to back (loaded
                    // inside the pass loop each one waited out its own latency: +90 us on the 44400x1024 dgrad).
                    typedef typename std::conditional<sizeof(T) == 2, uint2, uint4>::type MaskRaw;
                    MaskRaw mraw[PC];
                    uint4 rraw[PC];
#pragma unroll
                    for (int q = 0; q < PC; ++q) {
                        const int rl = (pc + q) * RPP + lane / CG;
                        const int m = cur.m0 + wr * WT + rl;
                        const bool ok = col_ok && m < p.M;
                        mraw[q] = MaskRaw{};
                        rraw[q] = uint4{0u, 0u, 0u, 0u};
                        if (has_mask && ok)
                            mraw[q] = *reinterpret_cast<const MaskRaw*>(reinterpret_cast<const T*>(p.relu_mask) + cur.coff + (int64_t)m * p.ldm + ncol);
                        if (has_res && ok) {
                            if (res_f32) rraw[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(p.residual) + cur.coff + (int64_t)m * p.ldr + ncol);
                            else {
                                const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.residual) + cur.coff + (int64_t)m * p.ldr + ncol);
                                rraw[q].x = h.x; rraw[q].y = h.y;
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < PC; ++q) {
                        const int rl = (pc + q) * RPP + lane / CG;
                        const int m = cur.m0 + wr * WT + rl;
                        float4 v = *reinterpret_cast<const float4*>(ew + rl * EPI_LD + cg * 4);
                        if (!(col_ok && m < p.M)) continue;
                        v.x = v.x * p.alpha + bias4.x; v.y = v.y * p.alpha + bias4.y;
                        v.z = v.z * p.alpha + bias4.z; v.w = v.w * p.alpha + bias4.w;
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if (has_mask) {
                            float4 mk;
                            if constexpr (sizeof(T) == 2) {
                                mk.x = __uint_as_float(mraw[q].x << 16); mk.y = __uint_as_float(mraw[q].x & 0xFFFF0000u);
                                mk.z = __uint_as_float(mraw[q].y << 16); mk.w = __uint_as_float(mraw[q].y & 0xFFFF0000u);
                            } else {
                                mk.x = __uint_as_float(mraw[q].x); mk.y = __uint_as_float(mraw[q].y);
                                mk.z = __uint_as_float(mraw[q].z); mk.w = __uint_as_float(mraw[q].w);
                            }
                            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                            v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                        }
                        if (has_res) {
                            if (res_f32) {
                                v.x += __uint_as_float(rraw[q].x); v.y += __uint_as_float(rraw[q].y);
                                v.z += __uint_as_float(rraw[q].z); v.w += __uint_as_float(rraw[q].w);
                            } else {
                                v.x += __uint_as_float(rraw[q].x << 16); v.y += __uint_as_float(rraw[q].x & 0xFFFF0000u);
                                v.z += __uint_as_float(rraw[q].y << 16); v.w += __uint_as_float(rraw[q].y & 0xFFFF0000u);
                            }
                        }
                        store4<TC>(C + (int64_t)m * p.ldc + ncol, v);
                        if (p.colstats != nullptr) {
                            if constexpr (sizeof(TC) != 4) {   // statistics of the values as stored
                                v.x = (float)(TC)v.x; v.y = (float)(TC)v.y; v.z = (float)(TC)v.z; v.w = (float)(TC)v.w;
                            }
                            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                            cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
                        }
                    }
                }
                if (p.colstats != nullptr) {
                    float* f[2] = {&cs.x, &cq.x};
#pragma unroll
                    for (int wq = 0; wq < 2; ++wq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float sv = f[wq][e];
#pragma unroll
                            for (int o = CG; o < 64; o <<= 1) sv += __shfl_xor(sv, o, 64);
                            f[wq][e] = sv;
                        }
                    if (lane < CG && col_ok) {
                        if (lds_stats) {        // (two explicit branches: a select of an LDS and a global pointer is a flat pointer)
                            atomicAdd(cacc + ncol + 0, cs.x); atomicAdd(cacc + ncol + 1, cs.y);
                            atomicAdd(cacc + ncol + 2, cs.z); atomicAdd(cacc + ncol + 3, cs.w);
                            if (p.colstats_mode == 0) {
                                atomicAdd(cacc + COLSTAT_LDS_N + ncol + 0, cq.x); atomicAdd(cacc + COLSTAT_LDS_N + ncol + 1, cq.y);
                                atomicAdd(cacc + COLSTAT_LDS_N + ncol + 2, cq.z); atomicAdd(cacc + COLSTAT_LDS_N + ncol + 3, cq.w);
                            }
                        } else {
                            atomicAdd(p.colstats + ncol + 0, cs.x); atomicAdd(p.colstats + ncol + 1, cs.y);
                            atomicAdd(p.colstats + ncol + 2, cs.z); atomicAdd(p.colstats + ncol + 3, cs.w);
                            if (p.colstats_mode == 0) {
                                atomicAdd(p.colstats + p.N + ncol + 0, cq.x); atomicAdd(p.colstats + p.N + ncol + 1, cq.y);
                                atomicAdd(p.colstats + p.N + ncol + 2, cq.z); atomicAdd(p.colstats + p.N + ncol + 3, cq.w);
                            }
                        }
                    }
                }
            }
            lds_barrier();            // the epilogue image is free again before the next stage is stored
        }
    };

    // one step of the flattened stage stream: fetch stage s+D, compute stage s, finish the tile if it was its last
    // stage, move stage s+1 from registers to LDS
    auto step = [&](auto slot_c, auto next_c, int s) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;   // = s % D: the register set that held stage s (free now)
        constexpr int BUF = SLOT & 1;                   // = s % 2 (D is even): compile-time LDS buffer
        const bool tile_end = (cit + 1 == ck.it1);
        // Stage s+1 goes from registers to the other LDS buffer FIRST (its loads were issued a whole step ago), so the
        // LDS writes run under this step's MFMAs instead of sitting between them and the barrier.  Exception: a step
        // that ends a tile needs the whole LDS image for its epilogue and stores afterwards.
        if (!tile_end && s + 1 < nst) store_stage(next_c, IC<(BUF ^ 1)>{});
        issue_load(slot_c);             // stage s+D (all-OOB once the block's stages are exhausted)
        const unsigned char* la = smem + BUF * 2 * G::STAGE;
        const unsigned char* lb = la + G::STAGE;
        if constexpr (!AKM && !BKM) {
            // all fragment reads of the stage in flight before the first MFMA: LDS latency is exposed once, not twice
            typename Frag<T>::type fa[2][NI], fb[2][NI];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[ks][i] = *reinterpret_cast<const typename Frag<T>::type*>(la + rdA[ks] + i * 2048);
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[ks][j] = *reinterpret_cast<const typename Frag<T>::type*>(lb + rdB[ks] + j * 2048);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) mma<T>(acc[i][j], fa[ks][i], fb[ks][j]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typename Frag<T>::type fa[NI], fb[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if constexpr (!AKM) fa[i] = *reinterpret_cast<const typename Frag<T>::type*>(la + rdA[ks] + i * 2048);
                    else fa[i] = read_frag<T, AKM>(la, wr * WT + i * 16, ks, lane);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr (!BKM) fb[j] = *reinterpret_cast<const typename Frag<T>::type*>(lb + rdB[ks] + j * 2048);
                    else fb[j] = read_frag<T, BKM>(lb, wc * WT + j * 16, ks, lane);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) mma<T>(acc[i][j], fa[i], fb[j]);
            }
        }
        ++cit;
        if (tile_end) {                 // last stage of this work item
            epilogue();
            if (++cw < wend) { next_work(ck); cit = ck.it0; }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s + 1 < nst) store_stage(next_c, IC<(BUF ^ 1)>{});
        }
        lds_barrier();
    };

    // prologue: D stages in flight, stage 0 to LDS
    issue_load(IC<0>{});
    if constexpr (D > 1) issue_load(IC<1>{});
    if constexpr (D > 2) { issue_load(IC<2>{}); issue_load(IC<3>{}); }
    store_stage(IC<0>{}, IC<0>{});
    lds_barrier();

    int s = 0;
    if constexpr (D == 2) {
        for (; s + 2 <= nst; s += 2) { step(IC<0>{}, IC<1>{}, s); step(IC<1>{}, IC<0>{}, s + 1); }
        if (s < nst) step(IC<0>{}, IC<1>{}, s);
    } else {
        for (; s + 4 <= nst; s += 4) {
            step(IC<0>{}, IC<1>{}, s); step(IC<1>{}, IC<2>{}, s + 1); step(IC<2>{}, IC<3>{}, s + 2); step(IC<3>{}, IC<0>{}, s + 3);
        }
        if (s < nst) { step(IC<0>{}, IC<1>{}, s); ++s; }
        if (s < nst) { step(IC<1>{}, IC<2>{}, s); ++s; }
        if (s < nst) { step(IC<2>{}, IC<3>{}, s); ++s; }
    }
    if (lds_stats) {        // the last step ended with a barrier: every wave's LDS adds are done
        for (int n = tid; n < p.N; n += 256) {
            const float a = cacc[n];
            if (a != 0.f) atomicAdd(p.colstats + n, a);
            if (p.colstats_mode == 0) {
                const float q = cacc[COLSTAT_LDS_N + n];
                if (q != 0.f) atomicAdd(p.colstats + p.N + n, q);
            }
        }
    }
}

template <typename T, typename TC, bool AKM, bool BKM, int TM, bool DIRECT>
int launch1(const FS2Gemm& g, int total, hipStream_t st) {
    const int slots = (TM == 128) ? 512 : 768;      // resident blocks: 2 (128-tile) / 3 (64-tile) per CU x 256 CUs
    int grid = total < slots ? total : slots;
    if (g.tile_order == 2) {                        // column panels aligned to multiples of 8 blocks (one XCD per slab group)
        const int unit = 8 * ((g.N + TM - 1) / TM);
        grid = (slots / unit) * unit;
    }
    const int lds = Geo<TM>::SMEM + (g.colstats != nullptr && g.N <= COLSTAT_LDS_N ? 2 * COLSTAT_LDS_N * 4 : 0);
    hipLaunchKernelGGL((gemm_kernel<T, TC, AKM, BKM, TM, DIRECT>), dim3(grid), dim3(256), lds, st, g, total);
    FS2_CHECK_LAUNCH("fs2_gemm");
    return FS2_OK;
}

template <typename T, typename TC>
int launch(const FS2Gemm& g, hipStream_t st) {
    const long zdim = (long)g.batch1 * g.batch2 * g.split_k;
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * zdim;
    g_last_tile = 128;
    if (!g.a_kmajor && !g.b_kmajor) {
        if (t128 < 384) {       // too few 128-tiles to fill 256 CUs: quarter-size tiles
            const long t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * zdim;
            g_last_tile = 64;
            return launch1<T, TC, false, false, 64, false>(g, (int)t64, st);
        }
        return launch1<T, TC, false, false, 128, false>(g, (int)t128, st);
    }
    if (!g.a_kmajor && g.b_kmajor) return launch1<T, TC, false, true, 128, false>(g, (int)t128, st);
    if (g.a_kmajor && g.b_kmajor) return launch1<T, TC, true, true, 128, false>(g, (int)t128, st);
    fs2_set_error("fs2_gemm: a_kmajor=1 with b_kmajor=0 is not provided");
    return FS2_EINVAL;
}

}  // namespace

extern "C" int fs2_gemm(const FS2Gemm* gp, void* stream) {
    FS2_REQUIRE(gp != nullptr, "fs2_gemm: null descriptor");
    FS2Gemm g = *gp;
    if (g.dtype == FS2_FP8 || g.dtype == FS2_BF8_FP8) {     // one-byte operands: the 16-wave LDS-DMA kernel only
        FS2_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.A && g.B && g.C, "fs2_gemm: empty problem / null operand");
        FS2_REQUIRE(fs2_aligned16(g.A) && fs2_aligned16(g.B) && fs2_aligned16(g.C), "fs2_gemm: operands must be 16-byte aligned");
        FS2_REQUIRE(g.c_dtype == FS2_F32 || g.c_dtype == FS2_BF16, "fs2_gemm: fp8 operands write fp32 or bf16");
        FS2_REQUIRE(g.ldc % 8 == 0 && g.N % 8 == 0, "fs2_gemm: fp8 operands need N and ldc in multiples of 8");
        if (g.split_k < 1) g.split_k = 1;
        if (g.batch1 < 1) g.batch1 = 1;
        if (g.batch2 < 1) g.batch2 = 1;
        if (g.conv == 0) { g.taps = 1; g.pad = 0; if (g.seq_len <= 0) g.seq_len = 1; }
        if (g.conv == 1) FS2_REQUIRE(g.taps >= 1 && g.seq_len > 0, "fs2_gemm: conv=1 needs taps, seq_len");
        if (g.bias) FS2_REQUIRE(fs2_aligned16(g.bias), "fs2_gemm: bias must be 16-byte aligned");
        int rc = FS2_OK;
        if (fs2_gemm_ring_try(g, (hipStream_t)stream, &rc)) return rc;      // ring kernel, block-scaled 128-deep fp8 MFMA (g_last_tile 130 / 192)
        FS2_REQUIRE(g.q8 == nullptr, "fs2_gemm: FS2Gemm.q8 needs the fp8 ring kernel (bf16 C, contiguous rows, N %% 16 == 0)");
        fs2_set_error("fs2_gemm: fp8 operands need a row-major un-batched product with K, lda, ldb multiples of 16, no accumulate / "
                      "split-K, a compiled epilogue combination and 32-bit addressable operands");
        return FS2_EINVAL;
    }
    FS2_REQUIRE(g.dtype == FS2_F32 || g.dtype == FS2_BF16, "fs2_gemm: bad dtype %d", g.dtype);
    FS2_REQUIRE(g.q8 == nullptr, "fs2_gemm: FS2Gemm.q8 (fp8 copy of C) exists for fp8 operands only");
    FS2_REQUIRE(g.c_dtype == FS2_F32 || g.c_dtype == g.dtype, "fs2_gemm: c_dtype must be f32 or the operand dtype");
    FS2_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "fs2_gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    if (g.split_k < 1) g.split_k = 1;
    if (g.batch1 < 1) g.batch1 = 1;
    if (g.batch2 < 1) g.batch2 = 1;
    const int epc = g.dtype == FS2_BF16 ? 8 : 4;
    FS2_REQUIRE(g.A && g.B && g.C, "fs2_gemm: null operand");
    FS2_REQUIRE(fs2_aligned16(g.A) && fs2_aligned16(g.B) && fs2_aligned16(g.C), "fs2_gemm: operands must be 16-byte aligned");
    FS2_REQUIRE(g.lda % epc == 0 && g.ldb % epc == 0, "fs2_gemm: lda/ldb must be multiples of %d elements", epc);
    FS2_REQUIRE(g.sA1 % epc == 0 && g.sA2 % epc == 0 && g.sB1 % epc == 0 && g.sB2 % epc == 0,
                "fs2_gemm: batch strides of A/B must be multiples of %d elements", epc);
    FS2_REQUIRE(g.N % 4 == 0 || g.ldc >= ((g.N + 3) / 4) * 4, "fs2_gemm: N=%d needs ldc >= N rounded up to 4", g.N);
    FS2_REQUIRE(g.ldc % 4 == 0 && g.sC1 % 4 == 0 && g.sC2 % 4 == 0, "fs2_gemm: ldc and C batch strides must be multiples of 4");
    if (g.c_dtype == FS2_BF16) FS2_REQUIRE(g.ldc % 8 == 0 && g.sC1 % 8 == 0 && g.sC2 % 8 == 0, "fs2_gemm: bf16 C needs ldc and batch strides in multiples of 8 (16-byte stores)");
    if (!g.a_kmajor) FS2_REQUIRE(g.K % epc == 0, "fs2_gemm: K=%d must be a multiple of %d for a row-major A", g.K, epc);
    if (g.a_kmajor) FS2_REQUIRE(((g.M + epc - 1) / epc) * epc <= g.lda, "fs2_gemm: k-major A needs lda >= M rounded up to %d", epc);
    if (g.b_kmajor) FS2_REQUIRE(((g.N + epc - 1) / epc) * epc <= g.ldb, "fs2_gemm: k-major B needs ldb >= N rounded up to %d", epc);
    FS2_REQUIRE(g.conv >= 0 && g.conv <= 2, "fs2_gemm: bad conv mode");
    if (g.conv == 1) FS2_REQUIRE(!g.a_kmajor && !g.b_kmajor && g.taps >= 1 && g.seq_len > 0, "fs2_gemm: conv=1 needs row-major operands, taps, seq_len");
    if (g.conv == 2) FS2_REQUIRE(g.a_kmajor && g.b_kmajor && g.seq_len > 0 && g.batch2 >= 1, "fs2_gemm: conv=2 needs k-major operands and seq_len");
    if (g.conv == 0) { g.taps = 1; g.pad = 0; if (g.seq_len <= 0) g.seq_len = 1; }
    FS2_REQUIRE(!(g.split_k > 1 && !g.accumulate), "fs2_gemm: split_k > 1 requires accumulate");
    FS2_REQUIRE(!(g.accumulate && g.c_dtype != FS2_F32), "fs2_gemm: accumulate requires fp32 C");
    FS2_REQUIRE(!(g.accumulate && (g.relu || g.relu_mask || g.residual || g.colstats)), "fs2_gemm: accumulate excludes relu/mask/residual/colstats");
    FS2_REQUIRE(g.accumulate >= 0 && g.accumulate <= 2, "fs2_gemm: accumulate must be 0, 1 or 2");
    if (g.accumulate == 2) {      // sliced split-K: the ring kernel only (gemm_ring.hip)
        FS2_REQUIRE(g.dtype == FS2_BF16 && !g.a_kmajor && !g.b_kmajor && g.conv <= 1 && (long)g.batch1 * g.batch2 == 1 && !g.bias && g.alpha == 1.f &&
                    g.N % 8 == 0 && g.N <= 2048 && g.sC1 >= (int64_t)g.M * g.ldc && g.sC1 % 4 == 0,
                    "fs2_gemm: accumulate = 2 (sliced split-K) needs un-batched row-major bf16 operands, no bias / alpha, N %% 8 == 0, N <= 2048 and sC1 >= M * ldc");
        if (g.conv == 0) { g.taps = 1; g.pad = 0; if (g.seq_len <= 0) g.seq_len = 1; }
        int rc = FS2_OK;
        if (fs2_gemm_ring_try(g, (hipStream_t)stream, &rc)) return rc;
        fs2_set_error("fs2_gemm: accumulate = 2: operand sizes beyond the ring kernel's 32-bit addressing");
        return FS2_EINVAL;
    }
    if (g.residual) FS2_REQUIRE(g.ldr % 4 == 0, "fs2_gemm: ldr must be a multiple of 4");
    if (g.relu_mask) FS2_REQUIRE(g.ldm % 4 == 0, "fs2_gemm: ldm must be a multiple of 4");
    if (g.bias) FS2_REQUIRE(fs2_aligned16(g.bias), "fs2_gemm: bias must be 16-byte aligned");
    // columns are stored in groups of 4: a ragged N writes zeros (products of zero-filled B rows) into
    // the pad columns [N, roundup4(N)) and cannot be combined with per-column operands
    if (g.N % 4 != 0)
        FS2_REQUIRE(!g.b_kmajor && !g.bias && !g.residual && !g.relu_mask && !g.colstats && !g.accumulate,
                    "fs2_gemm: N=%d not a multiple of 4 only for plain row-major-B products", g.N);
    {   // staging uses 32-bit byte offsets from the (batch-adjusted) operand base
        const long es = g.dtype == FS2_BF16 ? 2 : 4;
        const long ra = g.a_kmajor ? g.K : g.M, rb = g.b_kmajor ? (g.Kb > 0 ? g.Kb : g.K) : g.N;
        FS2_REQUIRE((ra + 16) * g.lda * es < 0x7FFFFFF0L && (rb + 16) * g.ldb * es < 0x7FFFFFF0L,
                    "fs2_gemm: an operand slice exceeds 2 GiB");
    }
    // no empty k-split: shrink split_k so that every split owns at least one stage
    {
        const int bk = g.dtype == FS2_BF16 ? 64 : 32;
        const int ntot = (g.conv == 1 ? g.taps : 1) * ((g.K + bk - 1) / bk);
        if (g.split_k > ntot) g.split_k = ntot;
        const int per = (ntot + g.split_k - 1) / g.split_k;
        g.split_k = (ntot + per - 1) / per;
    }
    const long total = (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * g.batch1 * g.batch2 * g.split_k;
    FS2_REQUIRE(total < (1L << 30), "fs2_gemm: too many work items");
    hipStream_t st = (hipStream_t)stream;
    {   // tall row-major bf16 products: the 16-wave LDS-DMA kernel (gemm_ring.hip)
        int rc = FS2_OK;
        if (fs2_gemm_ws_try(g, st, &rc)) return rc;       // tall products with K = 256: weights-stationary stream (g_last_tile 131)
        if (fs2_gemm_ring_try(g, st, &rc)) return rc;     // (sets g_last_tile to 130 / 192 / 256)
        if (fs2_gemm_big_km_try(g, st, &rc)) return rc;   // decoder-side weight gradients (g_last_tile 129)
    }
    // tile walk: tall row-major products with 2..8 column tiles of 128 go m-fastest on an XCD-aligned grid (see the
    // kernel's work numbering); everything else n-fastest.  FS2_GEMM_MFAST=0 disables the choice (measurements).
    if (g.tile_order != 1 && g.tile_order != 2) {
        static const bool allow = [] { const char* e = getenv("FS2_GEMM_MFAST"); return e == nullptr || e[0] != '0'; }();
        const long zdim = (long)g.batch1 * g.batch2 * g.split_k;
        const int tn = (g.N + 127) / 128, tm = (g.M + 127) / 128;
        const bool tall = !g.a_kmajor && !g.b_kmajor && zdim == 1 && (long)tn * tm >= 384 && tn >= 2 && tn <= 8 && tm >= 8 * tn;
        g.tile_order = (allow && tall) ? 2 : 1;
    }
    if (g.tile_order == 2)
        FS2_REQUIRE(!g.a_kmajor && !g.b_kmajor && (long)g.batch1 * g.batch2 * g.split_k == 1 && (long)((g.N + 127) / 128) * ((g.M + 127) / 128) >= 384 &&
                    (g.N + 127) / 128 <= 64, "fs2_gemm: tile_order 2 is for un-batched row-major products with at least 384 tiles of 128");
    if (g.dtype == FS2_BF16) {
        if (g.c_dtype == FS2_F32) return launch<bf16_t, float>(g, st);
        return launch<bf16_t, bf16_t>(g, st);
    }
    return launch<float, float>(g, st);
}

// Weight-gradient GEMM for gfx950, 16-wave form: C[Mo][No] (fp32) += alpha * sum_k A[k][m] * B[k'][n] with BOTH operands
// k-major (the reduction index -- batch x time -- is the row index of the two row-major activation tensors), the decoder-side
// weight gradients of the model: reduction length ~44 k rows, outputs of 256 x 256 ... 1024 x 256 (x taps).
//
// These products are bound by the cross-workgroup reduction: with 256 CUs and a handful of output tiles the reduction has to be
// split ~16-64 ways, and every workgroup then adds its fp32 partial tile to memory with float atomics (~1.3 TB/s chip-wide,
// i.e. one 256-byte wave-instruction per ~50 ns per CU).  The design therefore keeps the partial tile per CU SMALL
// (128 x 128 = 64 KiB, 12.8 us of atomics) and gets its MFMA work per staged byte from splitting K INSIDE the workgroup:
//   * block = 16 waves = 4 k-groups x (2 x 2 waves of 64 x 64): a stage is 128 reduction rows of a 128(m) x 128(n) output
//     tile; k-group g multiplies rows 32g .. 32g+31 of the stage (one MFMA k-step), every wave 4 x 4 MFMA 16x16x32 tiles;
//   * LDS-DMA staging (buffer_load_dwordx4 ... lds) into two 64 KiB stage buffers [A: 128 k][128 m] [B: 128 k][128 n], rows
//     of 256 B, 16-B chunk c of row r at c ^ (((r&3)<<2) | ((r>>2)&3)) (image (b) of the CDNA4 guide, T10) read with
//     ds_read_b64_tr_b16; the swizzle is applied on the per-lane source address (one wave-instruction writes 4 rows linearly);
//     reduction rows past K / outside the sequence (conv taps) and columns past the matrix use an out-of-range offset (zeros);
//   * at the end of a work item the four k-groups sum their accumulators with a two-round tree through LDS (plain 16-byte
//     accesses; LDS float atomics ran this at ~0.35 lanes per clock) and all 16 waves flush the 128 x 128 total with
//     row-contiguous float atomics (256 B per wave-instruction: the full-rate shape);
//   * persistent workgroups, one per CU; work items (output tile x tap x batch x k-split) numbered so that neighbouring items
//     share their reduction range (operand slabs hit in L2) and dealt to the 8 XCD groups in contiguous chunks.
#include "fs2_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int N> struct KIC { static constexpr int value = N; };

constexpr unsigned OOB = 0x80000000u;
constexpr int NT = 1024, NW = 16, TM = 128, TN = 128, BK = 128;
constexpr int OP_BYTES = BK * 256, STAGE = 2 * OP_BYTES, SMEM = 2 * STAGE;      // 32 KiB per operand per stage

__device__ __forceinline__ int km_f(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int km_off(int row, int ch) { return row * 256 + ((ch ^ km_f(row)) << 4); }

// fragment of the 16 columns o0 .. o0+15 for the 32 reduction rows r0k .. r0k+31 of an image
__device__ __forceinline__ bf16x8 km_frag(const unsigned char* img, int o0, int r0k, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int ch = (o0 >> 3) + (pp >> 1), r0 = r0k + 8 * g + q;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + km_off(r0, ch) + 8 * (pp & 1)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + km_off(r0 + 4, ch) + 8 * (pp & 1)));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

// one-byte operands: image rows of 128 B, 16-byte chunk c of row r at c ^ ((r >> 1) & 7); ds_read_b64_tr_b8 hands lane i of a 16-lane group
// column i of an 8-row x 16-column byte block (lane 2q + p of the group supplies the address of row q, columns 8p .. 8p+7 -- measured,
// tools/probes/tr8_probe.hip): the 8 k-values lane (column i16, k-group g) needs for v_mfma_f32_16x16x32_bf8_fp8.  Conflict-free: the
// 32 lanes of a half read 16 rows x 16 B of one logical chunk = 16 different (row parity, swizzled chunk) pairs x 2 halves.
typedef __attribute__((ext_vector_type(2))) int i32x2_t;
typedef __attribute__((address_space(3))) i32x2_t lds_i32x2;
__device__ __forceinline__ int km_f8(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ long km_frag8(const unsigned char* img, int o0, int r0k, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 1, pp = i16 & 1;
    const int r0 = r0k + 8 * g + q;
    const i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(img + r0 * 128 + (((o0 >> 4) ^ km_f8(r0)) << 4) + 8 * pp));
    union { i32x2_t v; long l; } u;
    u.v = v;
    return u.l;
}

}  // namespace

extern thread_local int g_last_tile;     // gemm.hip

// vblock / vgrid: this workgroup's index in, and the size of, the (virtual) grid of ITS product -- blockIdx / gridDim for a launch of one
// product, a sub-range of the grid in a grouped launch (fs2_gemm_big_km_grouped_kernel: several products, one launch)
// ES: bytes per operand element -- 2: bf16; 1: fp8 (A = dY in e5m2, B = X in e4m3: the copies the data-gradient / forward products of the
// fp8 operand mode already hold; half the staged and LDS-read bytes per multiply-add on a kernel that is bound by exactly those)
// KG: k-groups inside the workgroup.  4 (fp8): 128 reduction rows per stage of 32 KiB, 4 x (2 x 2 waves of 64 x 64).  2 (bf16, round 3): 64
// rows per stage = 32 KiB, so the 128 KiB hold a 4-deep ring with THREE stages in flight (the bf16 form of round 2 -- four k-groups, 64 KiB
// stages, one in flight -- was retired in round 4) --
// the operands of a weight gradient were just written by other kernels, and how many bytes a CU keeps outstanding decides how fast they
// arrive; 2 x (2 x 4 waves of 64 x 32): 32 accumulators per lane, 6 fragments per 8 MFMAs (LDS reads x 1.5, under the MFMA time).
template <int ES, int KG = 4>
__device__ __forceinline__ void km_body(const FS2Gemm& p, const int tilesM, const int tilesN, const int splits, const int nitems, const int rot_step,
                                        const int stream_units, float* __restrict__ ws, const int vblock, const int vgrid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert((KG == 4 && ES == 1) || (KG == 2 && ES == 2), "bf16: two k-groups; fp8: four");
    constexpr int NJ = KG == 2 ? 2 : 4;              // 16-column blocks of a wave's output tile (64 x 64 or 64 x 32)
    constexpr int SROWS = KG == 2 ? 64 : BK;          // reduction rows per stage
    const int kg = KG == 2 ? wave >> 3 : wave >> 2, wr = KG == 2 ? (wave >> 2) & 1 : (wave >> 1) & 1, wc = KG == 2 ? wave & 3 : wave & 1;
    const int g = lane >> 4, i16 = lane & 15;

    // ---- items of this block.
    // Uniform k-split (stream_units == 0): XCD group x = blockIdx & 7 owns the contiguous range [x*per_x, (x+1)*per_x) of item
    //      numbers (tile fastest, then k-split: the workgroups of an XCD run the same reduction rows of different tiles), its
    //      workgroups (slot = blockIdx >> 3) take them round-robin.
    // Balanced stream (stream_units = U > 0; for tile counts that no integer split maps onto 256 CUs, e.g. the 144 tiles of an
    //      encoder convolution's weight gradient: 56 % of the chip for 48 stages): the nbase tiles x nstk stages form one list
    //      of units (tile-major), workgroup L takes units [L*U, (L+1)*U), U <= nstk: the tail [s0, nstk) of one tile and the head
    //      [0, r) of the next.  It runs the HEAD FIRST: at step tau every workgroup of the chip is then at reduction stage tau or
    //      tau + nstk - U -- two k-streams chip-wide, so the operand slabs are shared in L2 exactly as under a uniform split.
    const int x = vblock & 7, slot = vblock >> 3, nslots = vgrid >> 3;
    const int per_x = (nitems + 7) >> 3;
    const int ibeg = x * per_x, iend = min(nitems, ibeg + per_x);
    const int have = stream_units > 0 ? 0 : iend - ibeg;
    const int nmine = have > slot ? (have - slot + nslots - 1) / nslots : 0;
    if (stream_units == 0 && nmine == 0) return;

    const int tiles = tilesM * tilesN;
    const int taps = p.conv == 2 ? p.batch2 : 1;
    const int nb2 = p.conv == 2 ? 1 : p.batch2;
    const int Kb = p.Kb > 0 ? p.Kb : p.K;
    const int nstk = (p.K + BK - 1) / BK;                   // stages of the whole reduction
    const int per = (nstk + splits - 1) / splits;           // (the host made every split non-empty)
    const int lda = (int)p.lda, ldb = (int)p.ldb;
    const int seq = p.seq_len;

    struct Item { int m0, n0, tap, st0, st1, rot; int64_t aoff, boff, coff; };
    // balanced stream: the (at most two) parts of this workgroup, head of the second tile first
    int npart = 0, part_tile[2] = {0, 0}, part_s0[2] = {0, 0}, part_s1[2] = {0, 0};
    // (workgroups b and b + 8 share an XCD: give each XCD a contiguous range of the list -- with the tile order below that is the
    //  (column tile, tap) tiles of ONE row block of the output, which all read the same columns of A)
    const int Lw = x * nslots + slot;
    if (stream_units > 0) {
        const long total = (long)nitems * nstk;               // (nitems = tiles x taps x batch here)
        const long ubeg = (long)Lw * stream_units;
        if (ubeg >= total) return;
        const long uend = ubeg + stream_units < total ? ubeg + stream_units : total;
        const int t0 = (int)(ubeg / nstk), s0 = (int)(ubeg - (long)t0 * nstk);
        const int n0 = (int)(uend - ubeg);
        if (s0 + n0 <= nstk) { npart = 1; part_tile[0] = t0; part_s0[0] = s0; part_s1[0] = s0 + n0; }
        else {
            npart = 2;
            part_tile[0] = t0 + 1; part_s0[0] = 0; part_s1[0] = n0 - (nstk - s0);
            part_tile[1] = t0; part_s0[1] = s0; part_s1[1] = nstk;
        }
    }
    auto decode = [&](int j) {
        // item number -> (tile fastest, then k-split, then tap, then batch): neighbours share the reduction range
        Item it;
        int z, sp;
        if (stream_units > 0) {
            const int pi = j == 0 ? 0 : 1;
            z = part_tile[pi]; sp = vblock;
            it.st0 = part_s0[pi]; it.st1 = part_s1[pi];
        } else {
            z = ibeg + j;
            sp = (z / tiles) % splits;
            it.st0 = sp * per; it.st1 = min(nstk, it.st0 + per);
            z = (z % tiles) + (z / (tiles * splits)) * tiles;     // drop the split digit: (tile, tap, batch)
        }
        int tm, tn;
        if (stream_units > 0) {      // column tile fastest, then tap, then row block, then batch
            tn = z % tilesN; z /= tilesN;
            it.tap = z % taps; z /= taps;
            tm = z % tilesM; z /= tilesM;
        } else {
            const int tile = z % tiles; z /= tiles;
            it.tap = z % taps; z /= taps;
            tm = tile / tilesN; tn = tile % tilesN;
        }
        const int b2 = z % nb2, b1 = z / nb2;
        it.m0 = tm * TM; it.n0 = tn * TN;
        it.rot = (sp * rot_step) & (TM - 1);     // the splits of one tile flush its rows in different orders (below)
        const int c2 = p.conv == 2 ? it.tap : b2;
        it.aoff = b1 * p.sA1 + (p.conv == 2 ? 0 : b2 * p.sA2);
        it.boff = b1 * p.sB1 + (p.conv == 2 ? 0 : b2 * p.sB2);
        it.coff = b1 * p.sC1 + c2 * p.sC2;
        return it;
    };

    // ---- LDS-DMA: instruction i of this wave covers image rows 4*(i*NW + wave) .. +3 of A (i = 0,1) and of B (i = 0,1);
    //      the lane fetches logical chunk (lane&15) ^ f(row) of row lane>>4 of those
    //      (fp8: ONE instruction per operand, image rows 8*wave .. +7 of 128 B, chunk (lane&7) ^ f8(row) of row lane>>3)
    //      (two k-groups: ONE instruction per operand as well: the 64 rows of a stage = 16 waves x 4 rows)
    constexpr int NI = (ES == 2 && KG == 4) ? 2 : 1, OPB = SROWS * 128 * ES, SBYTES = 2 * OPB;      // (SBYTES: one stage buffer, 64 / 32 KiB)
    auto dma_row = [&](int i) { return ES == 2 ? 4 * (i * NW + wave) + (lane >> 4) : 8 * wave + (lane >> 3); };
    auto dma_col = [&](int i) { return ES == 2 ? ((lane & 15) ^ km_f(dma_row(i))) * 8 : ((lane & 7) ^ km_f8(dma_row(i))) * 16; };

    // Per work item: the per-lane byte offsets of the lane's 16-byte chunk in reduction row dma_row(i) of the item's operands (OOB outside the
    // matrix' columns), and the scalar byte offset of the next stage's first row -- a stage request is then an s_mov m0 + a buffer_load per
    // piece.  What a wave does between a stage barrier and its first MFMA is paid by the whole workgroup (every wave is there at the same
    // moment; measured on the ring kernel: profiles/r04_d_ring_slot_overhead.txt), and the per-stage form of this address arithmetic
    // was ~70 vector instructions (a 32 x 32 multiply, a modulo for the taps) for 8 MFMAs.  Row validity: only the last stage of a
    // reduction can cross K (workgroup-uniform test); Conv1d taps (conv = 2): the frame index of the lane's row is carried along, the
    // base address is moved back by `pad` rows so that the scalar offset (rows + tap) is never negative.
    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned voA[NI], voB[NI];
    int tB[NI];
    int sA = 0, sB = 0;
    Item ck;
    int lst = 0;                          // next stage to stage
    const bool conv2 = p.conv == 2;
    auto prep = [&](const Item& k) {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(p.A) + k.aoff * ES), 0, 0x7FFFFFF0, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(p.B) + (k.boff - (conv2 ? (int64_t)p.pad * ldb : 0)) * ES), 0,
                                                0x7FFFFFF0, 0x00020000);
        lst = k.st0 * (BK / SROWS);
        const int kb0 = lst * SROWS;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            voA[i] = k.m0 + dma_col(i) < p.M ? (unsigned)((dma_row(i) * lda + k.m0 + dma_col(i)) * ES) : OOB;
            voB[i] = k.n0 + dma_col(i) < p.N ? (unsigned)((dma_row(i) * ldb + k.n0 + dma_col(i)) * ES) : OOB;
            tB[i] = conv2 ? (kb0 + dma_row(i)) % seq : 0;
        }
        sA = kb0 * lda * ES;
        sB = (kb0 + (conv2 ? k.tap : 0)) * ldb * ES;
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        const int kb = lst * SROWS;
        unsigned char* base = smem + buf * SBYTES + 1024 * wave;
        if constexpr (ES == 1) {
            // (fp8 operands: the per-stage form of the arithmetic -- with the lean requests below the configs[4] step in fp8 measured 0.8 %
            //  SLOWER, 30.3 -> 30.55 ms, alternating builds on one box; the bf16 kernel gains 4-7 % from them)
            const int shiftB = conv2 ? ck.tap : 0;                  // (rsB starts `pad` rows before the matrix)
            const int kk = kb + dma_row(0);
            const bool okA = voA[0] != OOB && kk < p.K;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)base, 16, (int)(okA ? (unsigned)((kk * lda + ck.m0 + dma_col(0)) * ES) : OOB), 0, 0, 0);
            bool okB = voB[0] != OOB && kk < Kb;
            if (conv2) { const int tt = (kk % seq) + ck.tap - p.pad; okB = okB && tt >= 0 && tt < seq; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(base + OPB), 16,
                                                     (int)(okB ? (unsigned)(((kk + shiftB) * ldb + ck.n0 + dma_col(0)) * ES) : OOB), 0, 0, 0);
            ++lst;
            return;
        }
        const bool tailA = kb + SROWS > p.K, tailB = kb + SROWS > Kb;          // (workgroup-uniform)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            unsigned v = voA[i];
            if (tailA) v = kb + dma_row(i) < p.K ? v : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(base + 1024 * NW * i), 16, (int)v, sA, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            unsigned v = voB[i];
            if (tailB) v = kb + dma_row(i) < Kb ? v : OOB;
            if (conv2) {
                v = (unsigned)(tB[i] + ck.tap - p.pad) < (unsigned)seq ? v : OOB;
                tB[i] += SROWS;
                if (seq >= SROWS) tB[i] = tB[i] >= seq ? tB[i] - seq : tB[i];
                else tB[i] %= seq;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(base + OPB + 1024 * NW * i), 16, (int)v, sB, 0, 0);
        }
        sA += SROWS * lda * ES;
        sB += SROWS * ldb * ES;
        ++lst;
    };

    f32x4 acc[4][NJ];

    const int jbeg = stream_units > 0 ? 0 : slot, jend = stream_units > 0 ? npart : have, jstep = stream_units > 0 ? 1 : nslots;
    for (int j = jbeg; j < jend; j += jstep) {
        ck = decode(j);
        prep(ck);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nst = (ck.st1 - ck.st0) * (BK / SROWS);      // (the host plans in units of 128 rows; rows past K read as zeros)
        if constexpr (KG == 2) {
            // ---- bf16, two k-groups: 4-deep ring of 32 KiB stages, as the fp8 path below
            const int pre = nst < 3 ? nst : 3;
            for (int i = 0; i < pre; ++i) issue(i);
            // one stage; the ring position is a compile-time constant (the loop below is unrolled by the ring depth): the fragment addresses
            // are lane offsets + immediates instead of twelve vector adds per stage.  Order: barrier -> fragment reads -> the request of
            // stage s+3 (in the shadow of the reads' latency) -> MFMAs.
            auto stage = [&](auto BUFC, const int s) __attribute__((always_inline)) {
                constexpr int BUF = decltype(BUFC)::value;
                const int younger = nst - 1 - s;
                if (__builtin_expect(younger >= 2, 1)) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned char* la = smem + BUF * SBYTES;
                const unsigned char* lb = la + OPB;
                bf16x8 fa[4], fb[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = km_frag(la, wr * 64 + i * 16, 32 * kg, lane);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) fb[jj] = km_frag(lb, wc * 32 + jj * 16, 32 * kg, lane);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 3 < nst) issue((BUF + 3) & 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int s = 0; s < nst; s += 4) {
                stage(KIC<0>{}, s);
                if (s + 1 < nst) stage(KIC<1>{}, s + 1);
                if (s + 2 < nst) stage(KIC<2>{}, s + 2);
                if (s + 3 < nst) stage(KIC<3>{}, s + 3);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if constexpr (ES == 1) {
            // ---- fp8: stages of 32 KiB in a 4-deep ring, THREE in flight (the two-buffer form of the bf16 path would leave half of its
            //      bytes in flight: the kernel is bound by what a CU keeps outstanding on the L2 -> LDS path).  Counted waits: a wave issues
            //      two LDS-DMA instructions per stage and they retire in order, so stage s has landed once at most 2 x (stages issued
            //      after it) remain; the buffer stage s+3 goes into was read in iteration s-1, before the barrier every wave has just passed.
            const int pre = nst < 3 ? nst : 3;
            for (int i = 0; i < pre; ++i) issue(i);
            // (request first, ring position at run time: the reads-first / unrolled order of the bf16 stage above measured 1-3 % SLOWER here)
            for (int s = 0; s < nst; ++s) {
                const int younger = nst - 1 - s;
                if (__builtin_expect(younger >= 2, 1)) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (s + 3 < nst) issue((s + 3) & 3);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* la = smem + (s & 3) * SBYTES;
                const unsigned char* lb = la + OPB;
                long fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = km_frag8(la, wr * 64 + i * 16, 32 * kg, lane);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) fb[jj] = km_frag8(lb, wc * 64 + jj * 16, 32 * kg, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        float* red = reinterpret_cast<float*>(smem + STAGE);      // the total, row-major [128][128] fp32, in the second 64 KiB
        if constexpr (KG == 2) {
            // ---- two k-groups: group 1 parks its accumulators in the first 64 KiB (lane-linear 16-byte slots), group 0 adds its own
            //      and writes the total row-major
            float4* red4 = reinterpret_cast<float4*>(smem);
            const int wslot = (wave & 7) * 512 + lane;                // per wave position: 8 float4 per lane
            if (kg == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        red4[wslot + (i * 2 + jj) * 64] = make_float4(acc[i][jj][0], acc[i][jj][1], acc[i][jj][2], acc[i][jj][3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kg == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const float4 v = red4[wslot + (i * 2 + jj) * 64];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            red[(wr * 64 + i * 16 + g * 4 + r) * TN + wc * 32 + jj * 16 + i16] = acc[i][jj][r] + (&v.x)[r];
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
        // ---- the four k-groups hold partial sums of the same 128 x 128 tile: a two-round tree through LDS (both stage buffers
        //      are free: nothing is staged across work items), plain 16-byte accesses in accumulator order (LDS float atomics
        //      ran this reduction at ~0.35 lanes per clock: 90 us per tile)
        float4* red4 = reinterpret_cast<float4*>(smem);
        const int wslot = (wave & 3) * 1024 + lane;               // per (wr, wc) wave position: 16 float4 per lane, lane-linear
        if (kg >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    red4[(kg - 2) * 4096 + wslot + (i * 4 + jj) * 64] = make_float4(acc[i][jj][0], acc[i][jj][1], acc[i][jj][2], acc[i][jj][3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kg < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 v = red4[kg * 4096 + wslot + (i * 4 + jj) * 64];
                    acc[i][jj][0] += v.x; acc[i][jj][1] += v.y; acc[i][jj][2] += v.z; acc[i][jj][3] += v.w;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    red4[wslot + (i * 4 + jj) * 64] = make_float4(acc[i][jj][0], acc[i][jj][1], acc[i][jj][2], acc[i][jj][3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 v = red4[wslot + (i * 4 + jj) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(wr * 64 + i * 16 + g * 4 + r) * TN + wc * 64 + jj * 16 + i16] = acc[i][jj][r] + (&v.x)[r];
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (ws != nullptr) {
            // ---- sliced flush (fs2_wgrad_sliced): the 128 x 128 partial tile goes to slice `item` of the workspace with plain
            //      16-byte stores (64 KiB contiguous per workgroup, ~6 TB/s chip-wide against ~1.3 TB/s of float atomics);
            //      fs2_wgrad_reduce adds the slices of every tile into the gradient.  Uniform k-split: slice = item number ibeg + j.
            //      Balanced stream: slice = Lw + tile -- strictly increasing along the unit list (workgroup L + 1 starts in the tile
            //      workgroup L ended in, or a later one), so no two parts share a slice and the (<= nstk / U + 2) parts of tile z are the
            //      slices L + z of the workgroups L whose unit range meets [z * nstk, (z + 1) * nstk).
            const int64_t slice = stream_units > 0 ? (int64_t)Lw + part_tile[j == 0 ? 0 : 1] : (int64_t)(ibeg + j);
            float4* dst = reinterpret_cast<float4*>(ws + slice * (TM * TN));
            const float4* src = reinterpret_cast<const float4*>(red);
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[tid + NT * i] = src[tid + NT * i];
        } else {
        // ---- flush with row-contiguous float atomics (256 B per wave-instruction: the full-rate shape), 8 rows per wave
        float* __restrict__ C = reinterpret_cast<float*>(p.C) + ck.coff;
        // (fp8 operands: the two de-quantisation factors are device scalars)
        const float alpha_eff = ES == 1 ? p.alpha * (p.scale_a != nullptr ? *p.scale_a : 1.f) * (p.scale_b != nullptr ? *p.scale_b : 1.f) : p.alpha;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = (wave * 8 + rr + ck.rot) & (TM - 1);       // rotated per split: the 16-64 workgroups of one tile do not
                                                                       // pile their atomics onto the same addresses at the same moment (-5 %)
            const int m = ck.m0 + row;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = ck.n0 + h * 64 + lane;
                const float v = red[row * TN + h * 64 + lane];
                if (m < p.M && n < p.N) atomicAdd(C + (int64_t)m * p.ldc + n, v * alpha_eff);
            }
        }
        }
        // the LDS is free again before the next item's first stage is staged
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

template <int ES, int KG = 4>
__global__ __launch_bounds__(1024, 4) void fs2_gemm_big_km_kernel(const FS2Gemm p, const int tilesM, const int tilesN, const int splits,
                                                                  const int nitems, const int rot_step, const int stream_units,
                                                                  float* __restrict__ ws) {
    km_body<ES, KG>(p, tilesM, tilesN, splits, nitems, rot_step, stream_units, ws, (int)blockIdx.x, (int)gridDim.x);
}

// Several weight-gradient products in ONE launch (the products of one layer's backward): product d owns the workgroups
// [wg_begin[d], wg_begin[d+1]) (a multiple of 8 each: the XCD of a workgroup is blockIdx & 7 in both numberings), uniform k-split, partial
// tiles to ws[d].  One launch ramp, one first-stage latency and one tail for the group instead of one per product, and splits sized for
// the group's total work (~256 items in all instead of ~256 per product: a third of the partial tiles).
constexpr int KM_GROUP = 4;
struct KmGroupArgs {
    FS2Gemm g[KM_GROUP];
    float* ws[KM_GROUP];
    int tilesM[KM_GROUP], tilesN[KM_GROUP], splits[KM_GROUP], nitems[KM_GROUP], wg_begin[KM_GROUP + 1];
    int n, rot_step;
};
template <int ES, int KG = 4>
__global__ __launch_bounds__(1024, 4) void fs2_gemm_big_km_grouped_kernel(const KmGroupArgs a) {
    int d = 0;
    while (d + 1 < a.n && (int)blockIdx.x >= a.wg_begin[d + 1]) ++d;
    km_body<ES, KG>(a.g[d], a.tilesM[d], a.tilesN[d], a.splits[d], a.nitems[d], a.rot_step, 0, a.ws[d], (int)blockIdx.x - a.wg_begin[d],
            a.wg_begin[d + 1] - a.wg_begin[d]);
}

namespace {

struct KmPlan { int tilesM, tilesN, splits, stream_units, grid; long nitems, base; };

// work decomposition of a weight-gradient product, or false when the 16-wave kernel does not take it
bool km_plan(const FS2Gemm& g, int mode, KmPlan& pl, bool allow_f8 = false) {
    const bool f8 = allow_f8 && g.dtype == FS2_BF8_FP8;      // (A = dY in e5m2, B = X in e4m3; fs2_wgrad_sliced / fs2_wgrad_grouped only)
    if ((g.dtype != FS2_BF16 && !f8) || g.c_dtype != FS2_F32 || !g.a_kmajor || !g.b_kmajor || !g.accumulate) return false;
    if (g.conv != 0 && g.conv != 2) return false;
    if (g.bias || g.residual || g.relu_mask || g.colstats || g.relu) return false;
    const long rowsA = (long)g.K + 16, rowsB = (long)(g.Kb > 0 ? g.Kb : g.K) + 64;
    if (rowsA * g.lda * 2 >= 0x7FFFFFF0L || rowsB * g.ldb * 2 >= 0x7FFFFFF0L) return false;
    pl.tilesM = (g.M + TM - 1) / TM; pl.tilesN = (g.N + TN - 1) / TN;
    const long nb = (long)g.batch1 * g.batch2;
    const long base = (long)pl.tilesM * pl.tilesN * nb;       // output tiles (x taps x batch)
    const int nstk = (g.K + BK - 1) / BK;
    // uniform k-split: about one item per CU, at least 2 stages per split (never more items than CUs: a second round would double the time)
    int splits = (int)(256 / base);
    if (splits > nstk / 2) splits = nstk / 2 > 0 ? nstk / 2 : 1;
    if (splits < 1) splits = 1;
    const int per = (nstk + splits - 1) / splits;
    splits = (nstk + per - 1) / per;                          // no empty split
    long nitems = base * splits;
    // balanced stream (see the kernel): tile counts no integer split maps onto 256 CUs
    int stream_units = 0;
    const long total = base * nstk;
    const bool fits = base <= 256 && nitems >= 208;           // the uniform split fills >= 81 % of the chip in one round
    if (!fits && total >= 1024 && base < (1L << 20)) {
        stream_units = (int)((total + 255) / 256);
        if (stream_units > nstk) stream_units = 0;            // (a workgroup's range may span two tiles, not three)
    }
    static const int stream_env = getenv("FS2_KM_STREAM") ? atoi(getenv("FS2_KM_STREAM")) : 1;      // 0: never (A/B measurements)
    if (!stream_env) stream_units = 0;
    // every weight gradient of the model with a reduction of >= 2048 rows: the decoder side (44 k rows), the encoder convolutions (balanced
    // stream) and the encoder's linear layers (6144 rows: 96 items of 2 stages -- as fast as the 4-wave kernel, A/B in the whole step
    // 8.155 vs 8.16 ms, one kernel family less)
    if (mode == 1 && nstk < 16) return false;
    if (stream_units > 0) nitems = base;
    if (nitems >= (1L << 30)) return false;
    const long per_x = (nitems + 7) / 8;
    pl.grid = stream_units > 0 ? 8 * (int)(((total + stream_units - 1) / stream_units + 7) / 8) : 8 * (int)(per_x < 32 ? per_x : 32);
    pl.splits = splits; pl.stream_units = stream_units; pl.nitems = nitems; pl.base = base;
    return true;
}

int km_launch(const FS2Gemm& g, const KmPlan& pl, float* ws, hipStream_t st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[16] = {};            // per device (one process per GPU is the deployment; a process driving several still works)
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_big_km_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_big_km_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            fs2_set_error("fs2_gemm: cannot raise the dynamic LDS limit of the weight-gradient kernel");
            return FS2_ELAUNCH;
        }
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    g_last_tile = 129;          // (measurement aid: the 16-wave weight-gradient kernel)
    static const int rot_step = getenv("FS2_KM_ROT") ? atoi(getenv("FS2_KM_ROT")) : 4;
    if (g.dtype != FS2_BF16)
        hipLaunchKernelGGL((fs2_gemm_big_km_kernel<1, 4>), dim3(pl.grid), dim3(NT), SMEM, st, g, pl.tilesM, pl.tilesN, pl.splits, (int)pl.nitems, rot_step,
                           pl.stream_units, ws);
    else
        hipLaunchKernelGGL((fs2_gemm_big_km_kernel<2, 2>), dim3(pl.grid), dim3(NT), SMEM, st, g, pl.tilesM, pl.tilesN, pl.splits, (int)pl.nitems, rot_step,
                           pl.stream_units, ws);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { fs2_set_error("fs2_gemm(big km): launch failed: %s", hipGetErrorString(e_)); return FS2_ELAUNCH; }
    return FS2_OK;
}

// out[m][n] += alpha * sum_s ws[((y * splits + s) * tiles + tile) * 128 * 128 + ...]: one workgroup per (descriptor, batch item y, tile,
// block of 8 rows), one float4 per thread, the slices of the k-split read 8 loads deep.
// Balanced-stream products (splits = -U < 0, reserved = stages of the whole reduction | conv flag): tile z of the stream's tile order (column
// tile fastest, then tap, then row block, then batch) is the sum of the slices L + z of the workgroups L = z*nstk / U .. ((z+1)*nstk - 1) / U.
constexpr int KM_STREAM_CONV = 1 << 30;
struct ReduceArgs { FS2WgradPart parts[40]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_k(const ReduceArgs a) {
    int b = blockIdx.x, d = 0;
    while (d + 1 < a.n && b >= a.parts[d + 1].block_begin) ++d;
    const FS2WgradPart& p = a.parts[d];
    b -= p.block_begin;
    const int chunk = b & 15;
    b >>= 4;
    const int tiles = p.tilesM * p.tilesN;
    const int tid = threadIdx.x;
    const int row = chunk * 8 + (tid >> 5), col = (tid & 31) * 4;
    int tm, tn, c2, b1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.splits < 0) {
        const int U = -p.splits, nstk = p.reserved & (KM_STREAM_CONV - 1);
        int z = b;
        tn = z % p.tilesN; z /= p.tilesN;
        if (p.reserved & KM_STREAM_CONV) { c2 = z % p.n2; z /= p.n2; tm = z % p.tilesM; b1 = z / p.tilesM; }
        else { tm = z % p.tilesM; z /= p.tilesM; c2 = z % p.n2; b1 = z / p.n2; }
        if (tm * TM + row >= p.M || tn * TN + col >= p.N) return;
        const long u0 = (long)b * nstk;
        const int L0 = (int)(u0 / U), L1 = (int)((u0 + nstk - 1) / U);
        const float* __restrict__ src = p.ws + ((int64_t)L0 + b) * (TM * TN) + row * TN + col;
        for (int L = L0; L <= L1; ++L, src += TM * TN) {
            const float4 w = *reinterpret_cast<const float4*>(src);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
    } else {
        const int y = b / tiles, tile = b - y * tiles;
        tm = tile / p.tilesN; tn = tile % p.tilesN;
        if (tm * TM + row >= p.M || tn * TN + col >= p.N) return;
        c2 = y % p.n2; b1 = y / p.n2;
        const float* __restrict__ src = p.ws + ((int64_t)y * p.splits * tiles + tile) * (TM * TN) + row * TN + col;
        const int64_t sstride = (int64_t)tiles * (TM * TN);
        int s = 0;
        for (; s + 8 <= p.splits; s += 8) {
            float4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const float4*>(src + (s + u) * sstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { v.x += w[u].x; v.y += w[u].y; v.z += w[u].z; v.w += w[u].w; }
        }
        for (; s < p.splits; ++s) {
            const float4 w = *reinterpret_cast<const float4*>(src + s * sstride);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
    }
    const int m = tm * TM + row, n = tn * TN + col;
    const float alpha = p.alpha * (p.scale_a != nullptr ? *p.scale_a : 1.f) * (p.scale_b != nullptr ? *p.scale_b : 1.f);    // (fp8 operands)
    float* o = p.dst + (int64_t)b1 * p.sC1 + (int64_t)c2 * p.sC2 + (int64_t)m * p.ldc + n;
    if (n + 3 < p.N && ((uintptr_t)o & 15) == 0) {
        float4 c = *reinterpret_cast<float4*>(o);
        c.x += v.x * alpha; c.y += v.y * alpha; c.z += v.z * alpha; c.w += v.w * alpha;
        *reinterpret_cast<float4*>(o) = c;
    } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < 4 && n + e < p.N; ++e) o[e] += vv[e] * alpha;
    }
}

}  // namespace

// false: not eligible / not chosen; true: launched, *rc holds the result
bool fs2_gemm_big_km_try(const FS2Gemm& g, hipStream_t st, int* rc) {
    // FS2_GEMM_BIG_KM: 0 never, 1 (default) where the shape heuristic says so, 2 wherever eligible (tests, A/B measurements)
    const char* e1 = getenv("FS2_GEMM_BIG_KM");
    const int mode = e1 ? atoi(e1) : 1;
    if (mode == 0) return false;
    KmPlan pl;
    if (!km_plan(g, mode, pl)) return false;
    *rc = km_launch(g, pl, nullptr, st);
    return true;
}

namespace {
// normalised copy of a descriptor + the preconditions of the 16-wave weight-gradient kernel (as fs2_gemm's own checks)
bool wgrad_desc_ok(const FS2Gemm* gp, FS2Gemm& g) {
    g = *gp;
    if (g.split_k < 1) g.split_k = 1;
    if (g.batch1 < 1) g.batch1 = 1;
    if (g.batch2 < 1) g.batch2 = 1;
    if (g.conv == 0) { g.taps = 1; g.pad = 0; if (g.seq_len <= 0) g.seq_len = 1; }
    if (!(g.M > 0 && g.N > 0 && g.K > 0 && g.A && g.B && g.C) || !fs2_aligned16(g.A) || !fs2_aligned16(g.B)) return false;
    const int al = g.dtype == FS2_BF8_FP8 ? 16 : 8;       // 16-byte chunks
    if (g.lda % al != 0 || g.ldb % al != 0 || g.sA1 % al != 0 || g.sA2 % al != 0 || g.sB1 % al != 0 || g.sB2 % al != 0) return false;
    if (((g.M + al - 1) / al) * al > g.lda || ((g.N + al - 1) / al) * al > g.ldb) return false;
    return true;
}
void fill_part(FS2WgradPart* part, const FS2Gemm& g, const float* ws, int tilesM, int tilesN, int splits, long base) {
    const int taps = g.conv == 2 ? g.batch2 : 1, nb2 = g.conv == 2 ? 1 : g.batch2;
    part->ws = ws; part->dst = (float*)g.C; part->ldc = g.ldc; part->sC1 = g.sC1; part->sC2 = g.sC2;
    part->M = g.M; part->N = g.N; part->tilesM = tilesM; part->tilesN = tilesN; part->splits = splits;
    part->n2 = g.conv == 2 ? taps : nb2; part->nbatch = (int)(base / ((long)tilesM * tilesN)); part->alpha = g.alpha; part->block_begin = 0; part->reserved = 0;
    part->scale_a = g.dtype == FS2_BF8_FP8 ? g.scale_a : nullptr;
    part->scale_b = g.dtype == FS2_BF8_FP8 ? g.scale_b : nullptr;
}
}  // namespace

// Sliced weight gradient: the product of fs2_gemm(a_kmajor = b_kmajor = 1, accumulate = 1) with the partial tiles of the k-split stored
// with plain stores into `ws` instead of float atomics on C; fs2_wgrad_reduce adds them to C later.  Returns the number of floats of
// `ws` it used and fills `part` -- or 0 when the product does not run in that form (not eligible, more than 256 output tiles, workspace
// too small): the caller then uses fs2_gemm.  Negative: error.  part->splits > 0: uniform k-split; < 0: balanced stream (-units per workgroup).
extern "C" int64_t fs2_wgrad_sliced(const FS2Gemm* gp, float* ws, int64_t ws_floats, FS2WgradPart* part, void* stream) {
    if (gp == nullptr || ws == nullptr || part == nullptr) { fs2_set_error("fs2_wgrad_sliced: null argument"); return FS2_EINVAL; }
    FS2Gemm g;
    if (!wgrad_desc_ok(gp, g) || !fs2_aligned16(ws)) return 0;
    const char* e1 = getenv("FS2_GEMM_BIG_KM");
    const int mode = e1 ? atoi(e1) : 1;
    KmPlan pl;
    if (mode == 0 || !km_plan(g, mode, pl, true)) return 0;
    static const int stream_sliced = getenv("FS2_KM_STREAM_SLICED") ? atoi(getenv("FS2_KM_STREAM_SLICED")) : 1;      // 0: atomics (A/B measurements)
    if (pl.stream_units > 0 && stream_sliced) {
        // balanced stream: every workgroup stores its one or two partial tiles to slice (workgroup + tile) of the workspace
        const int nstk = (g.K + BK - 1) / BK;
        const long nwg = (pl.base * nstk + pl.stream_units - 1) / pl.stream_units;
        const int64_t need = (nwg + pl.base) * (int64_t)(TM * TN);
        if (need <= ws_floats) {
            const int rc = km_launch(g, pl, ws, (hipStream_t)stream);
            if (rc != FS2_OK) return rc;
            fill_part(part, g, ws, pl.tilesM, pl.tilesN, -pl.stream_units, pl.base);
            part->reserved = nstk | (g.conv == 2 ? KM_STREAM_CONV : 0);
            return need;
        }
    }
    if (pl.stream_units > 0 || pl.nitems > pl.grid) {
        // balanced stream without room in the workspace, or more items than workgroups (> 256 output tiles): float-atomic flush straight
        // into the gradient.  bf16 operands take that road through fs2_gemm; fp8 operands (which fs2_gemm does not accept k-major) are
        // launched here: the product is complete on return, nothing to reduce
        if (g.dtype != FS2_BF8_FP8) return 0;
        const int rc = km_launch(g, pl, nullptr, (hipStream_t)stream);
        if (rc != FS2_OK) return rc;
        part->splits = 0;
        return 1;
    }
    const int64_t need = pl.nitems * (int64_t)(TM * TN);
    if (need > ws_floats || pl.nitems > pl.grid) return 0;          // (one item per workgroup: every slice is written exactly once)
    const int rc = km_launch(g, pl, ws, (hipStream_t)stream);
    if (rc != FS2_OK) return rc;
    fill_part(part, g, ws, pl.tilesM, pl.tilesN, pl.splits, pl.base);
    return need;
}

// How fs2_wgrad_sliced / fs2_wgrad_grouped would run the product: 0 not at all (fs2_gemm's own kernels), 1 uniform k-split with partial
// tiles, 2 balanced stream (partial tiles when the workspace has room, else float atomics) or more than 256 output tiles (float atomics)
extern "C" int fs2_wgrad_plan(const FS2Gemm* gp) {
    FS2Gemm g;
    if (gp == nullptr || !wgrad_desc_ok(gp, g)) return 0;
    const char* e1 = getenv("FS2_GEMM_BIG_KM");
    const int mode = e1 ? atoi(e1) : 1;
    KmPlan pl;
    if (mode == 0 || !km_plan(g, mode, pl, true)) return 0;
    return (pl.stream_units > 0 || pl.nitems > pl.grid) ? 2 : 1;
}

// n (<= 4) products of fs2_wgrad_sliced's kind in ONE launch; parts[i] describes the partial tiles of product i for fs2_wgrad_reduce.
// Products the grouped launch does not take (not eligible, balanced-stream decomposition, the group already holds 224 output tiles) are
// left out: parts[i].splits == 0 marks them and the caller launches those on their own.  Returns the floats of `ws` used, 0 when
// nothing was launched (no product taken / workspace too small), negative on error.
extern "C" int64_t fs2_wgrad_grouped(const FS2Gemm* descs, int n, float* ws, int64_t ws_floats, FS2WgradPart* parts, void* stream) {
    if (descs == nullptr || ws == nullptr || parts == nullptr || n < 1) { fs2_set_error("fs2_wgrad_grouped: null argument"); return FS2_EINVAL; }
    if (n > KM_GROUP || !fs2_aligned16(ws)) return 0;
    const char* e1 = getenv("FS2_GEMM_BIG_KM");
    const int mode = e1 ? atoi(e1) : 1;
    if (mode == 0) return 0;
    KmGroupArgs a;
    KmPlan pl[KM_GROUP];
    int nstk[KM_GROUP], src[KM_GROUP];       // src[k]: index in descs of the k-th product the group takes
    long work = 0, tiles = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        parts[i].splits = 0;                 // "not taken" until proven otherwise
        if (!wgrad_desc_ok(descs + i, a.g[m]) || !km_plan(a.g[m], mode, pl[m], true) || pl[m].stream_units > 0) continue;
        if (m > 0 && a.g[m].dtype != a.g[0].dtype) continue;            // (one operand format per launch)
        if (tiles + pl[m].base > 224) continue;
        nstk[m] = (a.g[m].K + BK - 1) / BK;
        work += pl[m].base * nstk[m];
        tiles += pl[m].base;
        src[m++] = i;
    }
    if (m == 0) return 0;
    // stages per item: the group's work over ~256 workgroups, at least 2; grown until the 8-aligned workgroup ranges fit 256
    int per = (int)((work + 255) / 256);
    if (per < 2) per = 2;
    for (;; ++per) {
        int wgs = 0;
        for (int i = 0; i < m; ++i) {
            int sp = (nstk[i] + per - 1) / per;
            const int pi = (nstk[i] + sp - 1) / sp;
            sp = (nstk[i] + pi - 1) / pi;                             // no empty split
            a.splits[i] = sp;
            a.nitems[i] = (int)(pl[i].base * sp);
            a.wg_begin[i] = wgs;
            wgs += 8 * ((a.nitems[i] + 7) / 8);
        }
        a.wg_begin[m] = wgs;
        if (wgs <= 256) break;
        if (per > (1 << 20)) return 0;
    }
    int64_t need = 0;
    for (int i = 0; i < m; ++i) {
        a.ws[i] = ws + need;
        a.tilesM[i] = pl[i].tilesM; a.tilesN[i] = pl[i].tilesN;
        need += (int64_t)a.nitems[i] * (TM * TN);
    }
    if (need > ws_floats) return 0;
    // invariants the kernel relies on (cheap, always checked): 8-aligned monotonic workgroup ranges inside one 256-workgroup round, every
    // taken product with at least one item and no more items than workgroups, its partial tiles inside [ws, ws + ws_floats)
    for (int i = 0; i < m; ++i) {
        if (!(a.wg_begin[i] % 8 == 0 && a.wg_begin[i] < a.wg_begin[i + 1] && a.wg_begin[i + 1] <= 256 && a.nitems[i] >= 1 &&
              a.nitems[i] <= a.wg_begin[i + 1] - a.wg_begin[i] && a.ws[i] >= ws && a.ws[i] + (int64_t)a.nitems[i] * (TM * TN) <= ws + ws_floats &&
              a.g[i].A != nullptr && a.g[i].B != nullptr && a.g[i].C != nullptr)) {
            fs2_set_error("fs2_wgrad_grouped: internal plan error for product %d (wg [%d, %d), items %d)", i, a.wg_begin[i], a.wg_begin[i + 1], a.nitems[i]);
            return FS2_EINVAL;
        }
    }
    for (int i = m; i < KM_GROUP; ++i) { a.g[i] = a.g[0]; a.ws[i] = nullptr; a.tilesM[i] = a.tilesN[i] = a.splits[i] = a.nitems[i] = 0; a.wg_begin[i + 1] = a.wg_begin[m]; }
    a.n = m;
    a.rot_step = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[16] = {};
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_big_km_grouped_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&fs2_gemm_big_km_grouped_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            fs2_set_error("fs2_wgrad_grouped: cannot raise the dynamic LDS limit of the weight-gradient kernel");
            return FS2_ELAUNCH;
        }
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    g_last_tile = 129;
    if (a.g[0].dtype != FS2_BF16) hipLaunchKernelGGL((fs2_gemm_big_km_grouped_kernel<1, 4>), dim3(a.wg_begin[m]), dim3(NT), SMEM, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((fs2_gemm_big_km_grouped_kernel<2, 2>), dim3(a.wg_begin[m]), dim3(NT), SMEM, (hipStream_t)stream, a);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { fs2_set_error("fs2_wgrad_grouped: launch failed: %s", hipGetErrorString(e_)); return FS2_ELAUNCH; }
    for (int i = 0; i < m; ++i) fill_part(parts + src[i], a.g[i], a.ws[i], pl[i].tilesM, pl[i].tilesN, a.splits[i], pl[i].base);
    return need;
}

extern "C" int fs2_wgrad_reduce(const FS2WgradPart* parts, int n, void* stream) {
    FS2_REQUIRE(parts != nullptr && n >= 1, "fs2_wgrad_reduce: no parts");
    for (int i0 = 0; i0 < n; i0 += 40) {
        ReduceArgs a;
        a.n = n - i0 < 40 ? n - i0 : 40;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            a.parts[i] = parts[i0 + i];
            FS2_REQUIRE(a.parts[i].ws && a.parts[i].dst && a.parts[i].splits != 0 && a.parts[i].tilesM >= 1 && a.parts[i].tilesN >= 1 && a.parts[i].nbatch >= 1 &&
                            a.parts[i].n2 >= 1, "fs2_wgrad_reduce: bad part %d", i0 + i);
            if (a.parts[i].splits < 0) {          // balanced stream: U units per workgroup, never more than the stages of one tile
                const int nstk = a.parts[i].reserved & (KM_STREAM_CONV - 1);
                FS2_REQUIRE(nstk >= 1 && -a.parts[i].splits <= nstk, "fs2_wgrad_reduce: bad balanced-stream part %d", i0 + i);
            }
            a.parts[i].block_begin = blocks;
            blocks += 16 * a.parts[i].nbatch * a.parts[i].tilesM * a.parts[i].tilesN;
        }
        hipLaunchKernelGGL(wgrad_reduce_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        FS2_CHECK_LAUNCH("fs2_wgrad_reduce");
    }
    return FS2_OK;
}

// Fused attention-probability kernel for gfx950 (bf16): P = dropout(softmax(mask(Q K^T / sqrt(dk)))) without the
// score tensor ever reaching HBM.
//
// Reference: Models/modules.py:7-21 (attention()): scores = q k^T / sqrt(d_k); masked_fill(mask == 0, -1e4) on keys;
// softmax over keys; dropout.  The unfused path (fs2_gemm + fs2_softmax_fwd) writes the (B,H,t,tp) scores (165 MB
// per decoder layer at t ~ 925), reads them back, and writes P and dropout(P).  Here a workgroup owns 64 query rows of
// one (batch, head): it keeps their whole score strip (64 x tp, bf16 -- the precision the unfused path stores) in
// LDS, streams the key rows through a double-buffered 64-key tile, and writes only P and dropout(P).
//
// Layouts: q/k rows of one head are `dk` contiguous bf16 at q + b*batch_stride + i*row_stride + h*head_stride (the
// fused qkv projection output); P/Pd are (B,[layers],H,t,tp) with tp = roundup8(t), pad columns written as 0.
// MFMA: S^T tile = K_tile (16 keys x dk) * Q^T, so that a lane's 4 accumulator values are 4 consecutive keys of one
// query row (one 8-byte LDS store into the strip).  Dropout uses the same Philox counters (element offset in P >> 3)
// as fs2_softmax_fwd, so fs2_softmax_bwd regenerates the masks unchanged.
#include <stdlib.h>
#include "fs2_common.h"

namespace {

constexpr int QB = 64;                 // query rows per workgroup (QBT = 32: half strips, two workgroups per CU)
constexpr int KB = 64;                 // keys per tile
constexpr int LDS_MAX = 160 * 1024;
constexpr int MASK_BYTES = 1024;       // key mask of one batch element (t <= 1024 whenever the strip fits)

template <int DK> struct KTile {
    static constexpr int CPR = DK / 8;             // 16-B chunks per key row
    static constexpr int RPL = 16 / CPR;           // key rows per 256-B LDS line
    static constexpr int BYTES = KB * DK * 2;
    static constexpr int KS = DK / 32;             // MFMA k-steps
    // byte offset of chunk `ch` of key row `row`: 16 rows read at one chunk index land on 16 different 16-B slots
    __device__ static __forceinline__ int off(int row, int ch) {
        const int line = row / RPL;
        return line * 256 + (((row % RPL) * CPR + (ch ^ (line % CPR))) << 4);
    }
};

// strip row length in elements: tp (+8 so that consecutive rows shift by an odd number of 16-B slots)
static inline int strip_ld(int tp) { return ((tp / 8) % 2 == 0) ? tp + 8 : tp; }

template <int N> struct IC { static constexpr int value = N; };

// Branch-free 16-byte global loads: a buffer descriptor over the operand and an out-of-range offset for rows that do
// not exist (the hardware returns zeros).  A load under `if (row < t)` puts control flow between the loads of a
// prefetch ring and hipcc then waits vmcnt(0) at every use -- the ring degenerates to one exposed latency per tile.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ bf16x8 ld16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}

struct AttnArgs {
    const bf16_t* qa;          // query-side rows (MODE 0: Q, MODE 1: dO)
    const bf16_t* kb;          // key-side rows   (MODE 0: K, MODE 1: V)
    int64_t q_row, q_batch, k_row, k_batch;     // element strides of the two operands
    int head_stride;
    const uint8_t* key_mask;   // MODE 0
    bf16_t* P;                 // MODE 0: probabilities out; MODE 1: saved probabilities in
    bf16_t* D;                 // MODE 0: dropout(P) out;    MODE 1: dS out
    int64_t p_batch, d_batch;  // batch strides of P and D
    int H, t, tp, sld;
    float alpha, pdrop;
    const uint64_t* rng;
    uint32_t site;
    // optional second product from the strip (DK == 128): O = strip' * X with X = V (MODE 0: attention output) or
    // K (MODE 1: dQ), strip' = dropout(P) / dS as written to D
    const bf16_t* xb;          // key-side rows of X, strides k_row / k_batch of its own
    int64_t x_row, x_batch;
    bf16_t* O;                 // (B, t, H, dk)-like output rows; nullptr = skip
    int64_t o_row, o_batch;
    float o_alpha;
    unsigned long long* dbg;   // tools/attn_phases.py: 8 shader-clock stamp slots per workgroup (nullptr normally)
};

unsigned long long* g_attn_dbg = nullptr;

// k-major B-operand image of a 64-key x 128-column tile (256-byte rows) and its ds_read_tr16 fragment read, as
// gemm.hip's Tile<bf16,true> / read_frag: fragment of columns n0..n0+15 for k-step ks (32 keys)
__device__ __forceinline__ int km_off(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }
__device__ __forceinline__ bf16x8 km_frag(const unsigned char* lds, int n0, int ks, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
    const int q = i16 >> 2, pp = i16 & 3;
    const int ch = (n0 >> 3) + (pp >> 1);
    const int r0 = ks * 32 + 8 * g + q;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + km_off(r0, ch) + 8 * (pp & 1)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + km_off(r0 + 4, ch) + 8 * (pp & 1)));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

// 512 threads = 8 waves.  Phase 1: wave w multiplies 16 keys of every 128-key super-tile against all 64 query rows;
// the 64 x tp product strip (bf16, what the unfused GEMM would have stored) stays in LDS.
// Phase 2: wave w owns query rows 8w .. 8w+7, two at a time.
//   MODE 0 (forward):  strip = alpha * Q K^T;  P = softmax(mask_keys(strip));  D = dropout(P)
//   MODE 1 (backward): strip = dO V^T = dP;    D = dS = P * (dP' - sum_j dP'_j P_j), dP' = dropout'(dP)
// QBT = query rows per workgroup: 64 (one workgroup per CU, two X-tile buffers) or 32 (strip + one X buffer <= 80 KiB and
// <= 128 VGPRs: two workgroups per CU, so that the VALU-bound phase 2 of one overlaps the memory-bound phases of the other).
template <int DK, int MODE, int QBT>
__global__ __launch_bounds__(512, QBT == 32 ? 2 : 1) void attn_strip_k(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef KTile<DK> KT;
    constexpr int KS = KT::KS;
    constexpr int RT = QBT / 16;          // row tiles of the strip
    constexpr int RPW = QBT / 8;          // strip rows per wave in phase 2
    constexpr int NXB = QBT == 64 ? 2 : 1;   // X / key tile buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int q0 = blockIdx.x * QBT, h = blockIdx.y, b = blockIdx.z;
    const int t = a.t, tp = a.tp, sld = a.sld;
    unsigned char* strip = smem;
    unsigned char* lmask = smem + QBT * sld * 2 + NXB * KT::BYTES;
    const bf16_t* qb = a.qa + (int64_t)b * a.q_batch + (int64_t)h * a.head_stride;
    const bf16_t* kb = a.kb + (int64_t)b * a.k_batch + (int64_t)h * a.head_stride;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(qb), rs_k = make_rsrc(kb);     // per (batch, head): offsets < 2^31

    const int nkt = (t + KB - 1) / KB;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (a.dbg != nullptr && tid == 0) a.dbg[blk * 8 + 0] = __builtin_amdgcn_s_memtime();

    // ---- phase 1: product strip.  Wave w takes keys st*128 + w*16 .. +15 of every 128-key super-tile st and loads
    // its key-side rows straight into MFMA fragments (lane (i16, g): 16 bytes of row i16 at k = 32*ks + 8*g): no LDS
    // staging, no barrier, and a 4-deep ring of super-tiles in flight per wave -- the rows come from L2 with ~1.8 us
    // latency, which a block-wide staged pipeline with one tile of lead exposed on every tile (30k of 84k cycles).
    const int nst = (t + 127) / 128;
    bf16x8 kf[4][KS];
    auto load_frags = [&](auto slot, int st) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        const int row = st * 128 + wave * 16 + i16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            kf[S][ks] = ld16(rs_k, row < t ? (unsigned)(row * (int)a.k_row + ks * 32 + g * 8) * 2u : OOB);
    };
    load_frags(IC<0>{}, 0); load_frags(IC<1>{}, 1); load_frags(IC<2>{}, 2); load_frags(IC<3>{}, 3);
    // query-side rows: ONE coalesced copy of the 64 x DK tile per workgroup into LDS (the key-tile buffers are idle
    // until phase 3), fragments from there.  Eight waves each fetching all 64 rows in fragment shape (16 rows x 64 B
    // per instruction) cost 128 scattered load instructions per workgroup on the vector-memory path.
    bf16x8 qf[RT][KS];                             // fragments of all strip rows (B operand: column = query row)
    {
        unsigned char* qt = smem + QBT * sld * 2;
        constexpr int QCH = QBT * KT::CPR;          // 16-byte chunks of the tile
#pragma unroll
        for (int j = 0; j < (QCH + 511) / 512; ++j) {
            const int c = tid + j * 512;
            if (c < QCH) {
                const int row = c / KT::CPR, ch = c % KT::CPR;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_q, q0 + row < t ? (unsigned)((q0 + row) * (int)a.q_row + ch * 8) * 2u : OOB, 0, 0);
                *reinterpret_cast<u32x4*>(qt + KT::off(row, ch)) = v;
            }
        }
        lds_barrier();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[rt][ks] = *reinterpret_cast<const bf16x8*>(qt + KT::off(rt * 16 + i16, ks * 4 + g));
    }
    if constexpr (MODE == 0)
        for (int j = tid; j < MASK_BYTES; j += 512) lmask[j] = (j < t) ? a.key_mask[(int64_t)b * t + j] : 0;   // 0 beyond t
    if (sld > tp) {            // pad columns [tp, sld): read (times zero-filled X rows) by the second product
        for (int j = tid; j < QBT * (sld - tp); j += 512)
            reinterpret_cast<bf16_t*>(strip)[(j / (sld - tp)) * sld + tp + j % (sld - tp)] = (bf16_t)0.f;
    }
    auto super_tile = [&](auto slot, int st) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        if (st >= nst) return;
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[S][ks], qf[rt][ks], acc[rt], 0, 0, 0);
        load_frags(slot, st + 4);                  // refill the slot (rows >= t load nothing)
        // acc[rt][r] = strip[query rt*16 + i16][key st*128 + wave*16 + g*4 + r]
        const int kcol = st * 128 + wave * 16 + g * 4;
        if (kcol < tp) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[rt][r] * a.alpha);
                *reinterpret_cast<bf16x4*>(strip + ((rt * 16 + i16) * sld + kcol) * 2) = o;
            }
        }
    };
    if (a.dbg != nullptr && tid == 0) a.dbg[blk * 8 + 4] = __builtin_amdgcn_s_memtime();      // prologue issued
    for (int st = 0; st < nst; st += 4) {
        super_tile(IC<0>{}, st);
        if (a.dbg != nullptr && tid == 0 && st == 0) a.dbg[blk * 8 + 5] = __builtin_amdgcn_s_memtime();  // first super-tile done
        super_tile(IC<1>{}, st + 1); super_tile(IC<2>{}, st + 2); super_tile(IC<3>{}, st + 3);
    }
    if (a.dbg != nullptr && tid == 0) a.dbg[blk * 8 + 6] = __builtin_amdgcn_s_memtime();      // wave 0 done with phase 1
    lds_barrier();           // the strip (and the key mask / pad columns) is complete

    if (a.dbg != nullptr && tid == 0) a.dbg[blk * 8 + 1] = __builtin_amdgcn_s_memtime();
    // ---- phase 2: rows straight from the strip; wave w owns query rows 8w .. 8w+7, two at a time
    const DropCtx dc = drop_ctx(a.rng, a.site, a.pdrop);
    const bool second = (DK == 128) && a.O != nullptr;
    const int ng = (tp + 511) / 512;               // 16-byte groups per lane (<= 2)
    constexpr int R = 2;
    // MODE 1: the saved probabilities of this wave's 8 rows come from HBM -- all 16 loads are issued before the first use
    bf16x8 pall[MODE == 1 ? RPW : 1][2];
    const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(a.P + (int64_t)b * a.p_batch + (int64_t)h * t * tp);   // (t, tp) of this head
    if constexpr (MODE == 1) {
#pragma unroll
        for (int r8 = 0; r8 < RPW; ++r8) {
            const int qrow = q0 + wave * RPW + r8;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int col = 8 * (lane + 64 * gi);
                pall[r8][gi] = ld16(rs_p, (qrow < t && col < tp) ? (unsigned)(qrow * tp + col) * 2u : OOB);
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; rr += R) {
        const int ql0 = wave * RPW + rr;
        if (q0 + ql0 >= t) break;                  // wave-uniform
        if constexpr (MODE == 0) {
            float e[R][2][8];
            float mx[R], sum[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ql = ql0 + r;
                mx[r] = -3.0e38f;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int col = 8 * (lane + 64 * gi);
                    bf16x8 raw = {};
                    uint2 mk = make_uint2(0u, 0u);
                    if (gi < ng && col < tp) {
                        raw = *reinterpret_cast<const bf16x8*>(strip + (ql * sld + col) * 2);
                        mk = *reinterpret_cast<const uint2*>(lmask + col);
                    }
                    // branch-free: invalid columns (>= t) become -3e38 (exp -> 0), masked keys -1e4 (masked_fill(mask == 0, -1e4))
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const uint32_t mb = ((c < 4 ? mk.x : mk.y) >> (8 * (c & 3))) & 0xFFu;
                        float x = (float)raw[c];
                        x = mb != 0 ? x : -1e4f;
                        x = (col + c < t) ? x : -3.0e38f;
                        e[r][gi][c] = x;
                        mx[r] = fmaxf(mx[r], x);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) mx[r] = wave_max(mx[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                sum[r] = 0.f;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        e[r][gi][c] = __expf(e[r][gi][c] - mx[r]);       // invalid columns: exp(-3e38 - mx) = 0
                        sum[r] += e[r][gi][c];
                    }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) sum[r] = wave_sum(sum[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int qrow = q0 + ql0 + r;
                if (qrow >= t) continue;               // wave-uniform
                const float inv = 1.f / sum[r];
                const int64_t off = (int64_t)b * a.p_batch + ((int64_t)h * t + qrow) * tp;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int col = 8 * (lane + 64 * gi);
                    if (!(gi < ng && col < tp)) continue;
                    bf16x8 o;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { e[r][gi][c] *= inv; o[c] = (bf16_t)e[r][gi][c]; }
                    *reinterpret_cast<bf16x8*>(a.P + off + col) = o;
                    if (a.D != a.P) {
                        if (dc.on) {
                            float ds[8];
                            drop_scale8(dc, (uint64_t)(off + col) >> 3, ds);
#pragma unroll
                            for (int c = 0; c < 8; ++c) e[r][gi][c] *= ds[c];
                        }
#pragma unroll
                        for (int c = 0; c < 8; ++c) o[c] = (bf16_t)e[r][gi][c];
                        *reinterpret_cast<bf16x8*>(a.D + off + col) = o;
                    }
                    if (second) *reinterpret_cast<bf16x8*>(strip + ((ql0 + r) * sld + col) * 2) = o;
                }
            }
        } else {
            float ge[R][2][8], pe[R][2][8];
            float dot[R];
            int64_t poff[R], doff[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int qrow = (q0 + ql0 + r < t) ? q0 + ql0 + r : t - 1;       // clamped duplicate, not stored
                poff[r] = (int64_t)b * a.p_batch + ((int64_t)h * t + qrow) * tp;
                doff[r] = (int64_t)b * a.d_batch + ((int64_t)h * t + qrow) * tp;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ql = (q0 + ql0 + r < t) ? ql0 + r : ql0;
                dot[r] = 0.f;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int col = 8 * (lane + 64 * gi);
                    bf16x8 raw = {};
                    if (gi < ng && col < tp) raw = *reinterpret_cast<const bf16x8*>(strip + (ql * sld + col) * 2);
                    float ds[8];
                    drop_scale8(dc, (uint64_t)(poff[r] + col) >> 3, ds);      // the forward's offsets
#pragma unroll
                    for (int c = 0; c < 8; ++c) {                              // pad columns [t,tp): dP' = P = 0
                        ge[r][gi][c] = (col + c < t) ? (float)raw[c] * ds[c] : 0.f;
                        pe[r][gi][c] = (col + c < t) ? (float)pall[rr + r][gi][c] : 0.f;
                        dot[r] += ge[r][gi][c] * pe[r][gi][c];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) dot[r] = wave_sum(dot[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (q0 + ql0 + r >= t) continue;       // wave-uniform
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int col = 8 * (lane + 64 * gi);
                    if (!(gi < ng && col < tp)) continue;
                    bf16x8 o;
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] = (bf16_t)(pe[r][gi][c] * (ge[r][gi][c] - dot[r]));
                    *reinterpret_cast<bf16x8*>(a.D + doff[r] + col) = o;
                    if (second) *reinterpret_cast<bf16x8*>(strip + ((ql0 + r) * sld + col) * 2) = o;
                }
            }
        }
    }

    // ---- phase 3 (DK == 128): O[64 x 128] = strip' (64 x tp, now dropout(P) / dS in bf16) * X (tp x 128).
    // Wave w: rows 16*(w&3).., columns 64*(w>>2)..; X tiles of 64 keys stream through the two key-tile buffers in
    // the k-major image (ds_read_tr16 fragments), one barrier per tile.
    if constexpr (DK == 128) {
        if (second) {
            lds_barrier();                       // strip rewritten by every wave; key-tile buffers free
            if (a.dbg != nullptr && tid == 0) a.dbg[blk * 8 + 2] = __builtin_amdgcn_s_memtime();
            unsigned char* xt = smem + QBT * sld * 2;
            const bf16_t* xb = a.xb + (int64_t)b * a.x_batch + (int64_t)h * a.head_stride;
            // wave tile: 16 rows x CW columns (QBT = 64: 4 row tiles x 2 column groups of 64; QBT = 32: 2 x 4 groups of 32)
            constexpr int NCW = 8 / RT, CW = 128 / NCW, NT = CW / 16;
            const int rt = wave % RT, cg = wave / RT;
            f32x4 oacc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // X tiles come from L2 (~1 us): an 8-deep register ring keeps eight tile loads in flight per thread; with two
            // LDS buffers (QBT = 64) they alternate and a tile costs one barrier, with one buffer (QBT = 32) two
            u32x4 xr[8][2];
            const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xb);
            auto xload = [&](auto slot, int kt) __attribute__((always_inline)) {
                constexpr int S = decltype(slot)::value;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = tid + j * 512;
                    const int key = kt * KB + (c >> 4);
                    xr[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, key < t ? (unsigned)(key * (int)a.x_row + (c & 15) * 8) * 2u : OOB, 0, 0);
                }
            };
            auto xstore = [&](auto slot, int buf) __attribute__((always_inline)) {
                constexpr int S = decltype(slot)::value;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = tid + j * 512;
                    *reinterpret_cast<u32x4*>(xt + buf * 16384 + km_off(c >> 4, c & 15)) = xr[S][j];
                }
            };
            // one tile: refill the ring slot this tile came from, multiply from LDS, move the next tile into LDS
            auto xstep = [&](auto slot, auto next, int kt) __attribute__((always_inline)) {
                if (kt >= nkt) return;
                xload(slot, kt + 8);               // keys >= t load nothing (zeros)
                const unsigned char* xl = xt + (NXB == 2 ? (kt & 1) : 0) * 16384;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int kk = kt * KB + ks * 32;
                    if (kk < tp) {                 // uniform
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(strip + ((rt * 16 + i16) * sld + kk + g * 8) * 2);
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, km_frag(xl, cg * CW + j * 16, ks, lane), oacc[j], 0, 0, 0);
                    }
                }
                if constexpr (NXB == 2) {
                    if (kt + 1 < nkt) xstore(next, (kt + 1) & 1);
                } else {
                    lds_barrier();                 // every wave is done with the only buffer
                    if (kt + 1 < nkt) xstore(next, 0);
                }
                lds_barrier();
            };
            xload(IC<0>{}, 0); xload(IC<1>{}, 1); xload(IC<2>{}, 2); xload(IC<3>{}, 3);
            xload(IC<4>{}, 4); xload(IC<5>{}, 5); xload(IC<6>{}, 6); xload(IC<7>{}, 7);
            xstore(IC<0>{}, 0);
            lds_barrier();
            for (int kt = 0; kt < nkt; kt += 8) {
                xstep(IC<0>{}, IC<1>{}, kt);     xstep(IC<1>{}, IC<2>{}, kt + 1);
                xstep(IC<2>{}, IC<3>{}, kt + 2); xstep(IC<3>{}, IC<4>{}, kt + 3);
                xstep(IC<4>{}, IC<5>{}, kt + 4); xstep(IC<5>{}, IC<6>{}, kt + 5);
                xstep(IC<6>{}, IC<7>{}, kt + 6); xstep(IC<7>{}, IC<0>{}, kt + 7);
            }
            // stage the QBT x 128 tile (bf16) in X buffer 0, then 16-byte row-contiguous stores
            bf16_t* ot = reinterpret_cast<bf16_t*>(xt);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ot[(rt * 16 + g * 4 + r) * 128 + cg * CW + j * 16 + i16] = (bf16_t)(oacc[j][r] * a.o_alpha);
            lds_barrier();
#pragma unroll
            for (int j = 0; j < (QBT * 16 + 511) / 512; ++j) {
                const int c = tid + j * 512;
                const int row = c >> 4;
                if (row < QBT && q0 + row < t)
                    *reinterpret_cast<uint4*>(a.O + (int64_t)b * a.o_batch + (int64_t)(q0 + row) * a.o_row + (int64_t)h * a.head_stride + (c & 15) * 8) =
                        *reinterpret_cast<const uint4*>(ot + row * 128 + (c & 15) * 8);
            }
            if (a.dbg != nullptr && tid == 0) a.dbg[blk * 8 + 3] = __builtin_amdgcn_s_memtime();
        }
    }
}

template <int DK, int MODE, int QBT>
int launch_strip(const AttnArgs& a, int B, hipStream_t st, const char* name) {
    const int lds = QBT * a.sld * 2 + (QBT == 64 ? 2 : 1) * KTile<DK>::BYTES + MASK_BYTES;
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_strip_k<DK, MODE, QBT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    }
    dim3 grid((a.t + QBT - 1) / QBT, a.H, B);
    hipLaunchKernelGGL((attn_strip_k<DK, MODE, QBT>), grid, dim3(512), lds, st, a);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) {
        fs2_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));
        return FS2_ELAUNCH;
    }
    return FS2_OK;
}

// rows per workgroup: 64.  The 32-row variant (two workgroups per CU, so that the VALU-bound phase 2 of one could overlap
// the memory-bound phases of the other) is kept behind FS2_ATTN_QB=32 for measurements: at t ~ 925 it gives the same
// throughput (254 / 214 us vs 243 / 219 us per layer) -- per-CU vector-memory and VALU throughput, not latency, bound
// the kernel: each half-size workgroup takes as long as a full-size one alone.
static int pick_qb(const AttnArgs& a, int dk) {
    static const int forced = [] { const char* e = getenv("FS2_ATTN_QB"); return e ? atoi(e) : 0; }();
    const bool fits32 = 32 * a.sld * 2 + KB * dk * 2 + MASK_BYTES <= LDS_MAX / 2;
    if (forced == 64 || !fits32) return 64;
    return forced == 32 ? 32 : 64;
}

template <int MODE>
int dispatch_strip(const AttnArgs& a, int dk, int B, hipStream_t st, const char* name) {
    const bool half = pick_qb(a, dk) == 32;
    if (dk == 128) return half ? launch_strip<128, MODE, 32>(a, B, st, name) : launch_strip<128, MODE, 64>(a, B, st, name);
    if (dk == 64) return half ? launch_strip<64, MODE, 32>(a, B, st, name) : launch_strip<64, MODE, 64>(a, B, st, name);
    return half ? launch_strip<32, MODE, 32>(a, B, st, name) : launch_strip<32, MODE, 64>(a, B, st, name);
}

}  // namespace

// measurement hook (not part of the ABI header): device buffer of 4 stamps per workgroup, or nullptr to switch off
extern "C" void fs2_debug_attn_timer(unsigned long long* buf) { g_attn_dbg = buf; }

extern "C" int fs2_attn_probs_lds_bytes(int t, int dk) {
    if (t <= 0 || t > MASK_BYTES || (dk != 32 && dk != 64 && dk != 128)) return -1;
    const int tp = (t + 7) / 8 * 8;
    const int lds = QB * strip_ld(tp) * 2 + 2 * KB * dk * 2 + MASK_BYTES;
    return lds <= LDS_MAX ? lds : -1;
}

extern "C" int fs2_attn_probs_fwd(const void* q, const void* k, int64_t row_stride, int64_t batch_stride, int head_stride,
                                  int dk, const uint8_t* key_mask, void* p_out, void* pd_out, int64_t p_batch_stride,
                                  int B, int H, int t, int tp, float alpha, float p, const uint64_t* rng, uint32_t site,
                                  const void* v, void* o_out, int64_t o_row_stride, int64_t o_batch_stride, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FS2_REQUIRE(o_out == nullptr || (dk == 128 && v != nullptr && fs2_aligned16(v) && fs2_aligned16(o_out) && o_row_stride % 8 == 0 && o_batch_stride % 8 == 0),
                "fs2_attn_probs_fwd: the fused P V product needs dk == 128, v, and 16-byte aligned o_out rows");
    FS2_REQUIRE(fs2_attn_probs_lds_bytes(t, dk) > 0, "fs2_attn_probs_fwd: t=%d dk=%d does not fit the LDS strip (use fs2_gemm + fs2_softmax_fwd)", t, dk);
    FS2_REQUIRE(tp == (t + 7) / 8 * 8 && tp <= 1024, "fs2_attn_probs_fwd: tp must be roundup8(t) <= 1024 (t=%d tp=%d)", t, tp);
    FS2_REQUIRE(B > 0 && H > 0 && B <= 65535 && H <= 65535, "fs2_attn_probs_fwd: bad B/H");
    FS2_REQUIRE(row_stride % 8 == 0 && batch_stride % 8 == 0 && head_stride % 8 == 0 && p_batch_stride % 8 == 0,
                "fs2_attn_probs_fwd: strides must be multiples of 8 elements (16-byte accesses)");
    FS2_REQUIRE(fs2_aligned16(q) && fs2_aligned16(k) && fs2_aligned16(p_out) && fs2_aligned16(pd_out), "fs2_attn_probs_fwd: pointers must be 16-byte aligned");
    FS2_REQUIRE(p == 0.f || (rng != nullptr && pd_out != p_out), "fs2_attn_probs_fwd: dropout needs rng and a separate p_drop buffer");
    FS2_REQUIRE(p >= 0.f && p < 1.f, "fs2_attn_probs_fwd: p out of range");
    AttnArgs a;
    a.qa = (const bf16_t*)q; a.kb = (const bf16_t*)k;
    a.q_row = a.k_row = row_stride; a.q_batch = a.k_batch = batch_stride; a.head_stride = head_stride;
    a.key_mask = key_mask; a.P = (bf16_t*)p_out; a.D = (bf16_t*)pd_out; a.p_batch = a.d_batch = p_batch_stride;
    a.H = H; a.t = t; a.tp = tp; a.sld = strip_ld(tp); a.alpha = alpha; a.pdrop = p; a.rng = rng; a.site = site;
    a.xb = (const bf16_t*)v; a.x_row = row_stride; a.x_batch = batch_stride;       // v lives in the same fused tensor as q, k
    a.O = (bf16_t*)o_out; a.o_row = o_row_stride; a.o_batch = o_batch_stride; a.o_alpha = 1.f; a.dbg = g_attn_dbg;
    return dispatch_strip<0>(a, dk, B, st, "fs2_attn_probs_fwd");
}

extern "C" int fs2_attn_ds_bwd(const void* d_out, int64_t do_row_stride, int64_t do_batch_stride, const void* v,
                               int64_t v_row_stride, int64_t v_batch_stride, int head_stride, int dk, const void* p_saved,
                               int64_t p_batch_stride, void* ds_out, int64_t ds_batch_stride, int B, int H, int t, int tp,
                               float p, const uint64_t* rng, uint32_t site, const void* k, void* dq_out,
                               int64_t dq_row_stride, int64_t dq_batch_stride, float dq_alpha, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FS2_REQUIRE(dq_out == nullptr || (dk == 128 && k != nullptr && fs2_aligned16(k) && fs2_aligned16(dq_out) && dq_row_stride % 8 == 0 && dq_batch_stride % 8 == 0),
                "fs2_attn_ds_bwd: the fused dS K product needs dk == 128, k, and 16-byte aligned dq_out rows");
    FS2_REQUIRE(fs2_attn_probs_lds_bytes(t, dk) > 0, "fs2_attn_ds_bwd: t=%d dk=%d does not fit the LDS strip (use fs2_gemm + fs2_softmax_bwd)", t, dk);
    FS2_REQUIRE(tp == (t + 7) / 8 * 8 && tp <= 1024, "fs2_attn_ds_bwd: tp must be roundup8(t) <= 1024 (t=%d tp=%d)", t, tp);
    FS2_REQUIRE(B > 0 && H > 0 && B <= 65535 && H <= 65535, "fs2_attn_ds_bwd: bad B/H");
    FS2_REQUIRE(do_row_stride % 8 == 0 && do_batch_stride % 8 == 0 && v_row_stride % 8 == 0 && v_batch_stride % 8 == 0 &&
                head_stride % 8 == 0 && p_batch_stride % 8 == 0 && ds_batch_stride % 8 == 0,
                "fs2_attn_ds_bwd: strides must be multiples of 8 elements (16-byte accesses)");
    FS2_REQUIRE(fs2_aligned16(d_out) && fs2_aligned16(v) && fs2_aligned16(p_saved) && fs2_aligned16(ds_out), "fs2_attn_ds_bwd: pointers must be 16-byte aligned");
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_attn_ds_bwd: dropout needs rng");
    FS2_REQUIRE(p >= 0.f && p < 1.f, "fs2_attn_ds_bwd: p out of range");
    AttnArgs a;
    a.qa = (const bf16_t*)d_out; a.kb = (const bf16_t*)v;
    a.q_row = do_row_stride; a.q_batch = do_batch_stride; a.k_row = v_row_stride; a.k_batch = v_batch_stride;
    a.head_stride = head_stride; a.key_mask = nullptr;
    a.P = (bf16_t*)const_cast<void*>(p_saved); a.D = (bf16_t*)ds_out; a.p_batch = p_batch_stride; a.d_batch = ds_batch_stride;
    a.H = H; a.t = t; a.tp = tp; a.sld = strip_ld(tp); a.alpha = 1.f; a.pdrop = p; a.rng = rng; a.site = site;
    a.xb = (const bf16_t*)k; a.x_row = v_row_stride; a.x_batch = v_batch_stride;   // k lives in the same fused tensor as v
    a.O = (bf16_t*)dq_out; a.o_row = dq_row_stride; a.o_batch = dq_batch_stride; a.o_alpha = dq_alpha; a.dbg = g_attn_dbg;
    return dispatch_strip<1>(a, dk, B, st, "fs2_attn_ds_bwd");
}

// Flash-style attention for gfx950 (bf16, d_k = 128): attention() of the reference (Models/modules.py:7-21) and its backward
// WITHOUT the (t x t) probability tensors in HBM -- the mode the trainer runs when the attention maps are not requested
// (hp.return_attn = False).  Forward keeps per query row the running maximum and the sum of exponentials; backward recomputes
// the probabilities from Q, K and those two numbers and regenerates the dropout mask from the same Philox counters as every
// other kernel of the library (element offset of P[b, h, q, key] in the (B, [layers], H, t, tp) layout, >> 3), so this path and
// the LDS-strip path (attention.hip) draw IDENTICAL masks and differ only by rounding.
//
// Common structure (all three kernels): a workgroup = 8 waves owns 128 rows of one (batch, head) -- queries in the forward and
// dQ kernels, keys in the dK/dV kernel --, 16 rows per wave, held as MFMA fragments in registers; the other side streams through
// LDS in tiles of 64 rows x 128 columns by LDS-DMA (buffer_load_dwordx4 ... lds), double buffered, in the dual-use image of the
// CDNA4 guide (T10, image (b): 256-byte rows, 16-byte chunk c of row r at c ^ (((r&3)<<2) | ((r>>2)&3))), which serves both the
// row reads (ds_read_b128: operand rows) and the transposed reads (ds_read_b64_tr_b16: the same tile as the k-major operand of
// the second product).  Score tiles are computed TRANSPOSED relative to the product that consumes them, so that an accumulator
// tile is already the next MFMA's operand: with D = A B, lane (i16, g) holds D[4g + r][i16]; two 16-row tiles T, T+1 give the
// 8 k-values {16T + 4g + r} u {16(T+1) + 4g + r} of column i16 -- the B-operand fragment of a 32-deep k-step whose k order is
// that permutation; the A operand of the same k-step is gathered in the same order by two transposed reads of rows 16T + 4g + q
// and 16(T+1) + 4g + q.  No LDS round trip for the probabilities.
//
// Dropout: one Philox call covers 8 consecutive keys of one query.  A wave's 16 x 64 tile needs 128 calls = 2 per lane; each
// lane turns its two calls into 16 keep-bits and the tile's owners fetch them with one cross-lane move per 4 (forward, dQ:
// ds_bpermute) or 1 (dK/dV: DPP row_share) elements.
#include <stdlib.h>
#include "common.cuh"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr unsigned OOB = 0x80000000u;
constexpr int TILE = 64 * 256;          // bytes of one 64-row x 128-column bf16 tile
constexpr int MASK_BYTES = 1024;
constexpr int AUX_BYTES = 64 * 16;      // dK/dV kernel: {m, 1/l, delta, -} of the 64 queries of a tile

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ bf16x8 ld16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
__device__ __forceinline__ int img_f(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int img_off(int row, int ch) { return row * 256 + ((ch ^ img_f(row)) << 4); }

// operand rows r0 + i16, k-step ks (32 columns): the 16 bytes of chunk 4ks + g
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* img, int r0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + img_off(r0 + (lane & 15), 4 * ks + (lane >> 4)));
}
// transposed operand: column 16ct + i16 of the rows {rA + 4g + q} u {rB + 4g + q}, q = 0..3
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* img, int rA, int rB, int ct, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int ch = 2 * ct + (pp >> 1), sub = 8 * (pp & 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(rA + 4 * g + q, ch) + sub));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(rB + 4 * g + q, ch) + sub));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}
// 16 x 16 product tile over the 128 columns: rows r0.. of the LDS image against the register fragments bf (B operand)
__device__ __forceinline__ f32x4 tile128(const unsigned char* img, int r0, const bf16x8 (&bf)[4], int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(img, r0, ks, lane), bf[ks], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ bf16x8 pack8(const float (&a)[4], const float (&b)[4]) {
    bf16x8 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) { o[c] = (bf16_t)a[c]; o[4 + c] = (bf16_t)b[c]; }
    return o;
}
__device__ __forceinline__ float xor16_32_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor16_32_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
// keep-bits of the 16 consecutive elements e0 .. e0+15 (e0 % 8 == 0): bit j = element e0 + j survives (drop_scale8's rule)
__device__ __forceinline__ unsigned drop_bits16(const DropCtx& c, uint64_t e0) {
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint64_t q8 = (e0 >> 3) + j;
        const Philox4 r = philox4x32((uint32_t)q8, (uint32_t)(q8 >> 32), c.site, c.off, c.k0, c.k1);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bits |= ((w[i] & 0xFFFFu) >= c.thr16 ? 1u : 0u) << (8 * j + 2 * i);
            bits |= ((w[i] >> 16) >= c.thr16 ? 1u : 0u) << (8 * j + 2 * i + 1);
        }
    }
    return bits;
}
template <int N> __device__ __forceinline__ unsigned row_share(unsigned v) {       // lane N of the caller's row of 16 lanes
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xF, 0xF, false);
}

struct FlashArgs {
    const bf16_t *q, *k, *v;            // rows of one head: 128 contiguous bf16 at base + b*batch + i*row + h*head
    int64_t row, batch;                 // element strides of q / k / v (one fused projection tensor)
    int head;
    const uint8_t* key_mask;            // (B, t)
    bf16_t* O;                          // attention output (written by forward, read by backward), rows at O + b*o_batch + i*o_row + h*head
    int64_t o_row, o_batch;
    float* stats;                       // (B, H, t, 2): {row maximum of the masked scaled scores, sum of exponentials}
    int64_t p_batch;                    // batch stride of the virtual P tensor (dropout counters)
    int H, t, tp;
    float alpha, pdrop;
    const uint64_t* rng;
    uint32_t site;
    // backward
    const bf16_t* dO;                   // rows at dO + b*do_batch + i*do_row + h*head
    int64_t do_row, do_batch;
    float* aux;                         // (B, H, t, 4) workspace: {m, 1/l, delta = rowsum(dO * O), 0}: dQ kernel -> dK/dV kernel
    bf16_t *dq, *dk, *dv;               // rows at d? + b*g_batch + i*g_row + h*head
    int64_t g_row, g_batch;
};

// stage one 64-row tile: instruction i (0,1) of wave w covers tile rows 4*(8i + w) .. +3; lane -> row lane>>4, logical chunk
// (lane&15) ^ f(row); rows >= t use an out-of-range offset (zeros)
__device__ __forceinline__ void stage_tile(const __amdgpu_buffer_rsrc_t rs, unsigned char* dst, int row0, int t, int row_stride,
                                           int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 4 * (8 * i + wave) + (lane >> 4);
        const int ch = (lane & 15) ^ img_f(r);
        const int row = row0 + r;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + 1024 * (8 * i + wave)), 16,
                                                 (int)(row < t ? (unsigned)((row * row_stride + ch * 8) * 2) : OOB), 0, 0, 0);
    }
}

// masked, scaled score of a key: masked_fill(mask == 0, -1e4) (modules.py:12-14); keys that do not exist -> exp() = 0
__device__ __forceinline__ float mask_score(float s, float alpha, unsigned mk_byte, bool exists) {
    float v = s * alpha;
    v = mk_byte != 0 ? v : -1e4f;
    return exists ? v : -3.0e38f;
}

// ------------------------------------------------------------------------------------------------ forward
// O = dropout(softmax(mask(alpha Q K^T))) V, stats = {m, l}.  Wave: 16 queries (columns of the transposed score tiles).
__global__ __launch_bounds__(512, 4) void flash_fwd_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z, t = a.t;
    const int qrow = blockIdx.x * 128 + wave * 16 + i16;
    unsigned char* kimg = smem;                       // [2][TILE]
    unsigned char* vimg = smem + 2 * TILE;            // [2][TILE]
    unsigned char* lmask = smem + 4 * TILE;
    const int64_t hb = (int64_t)b * a.batch + (int64_t)h * a.head;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hb), rs_k = make_rsrc(a.k + hb), rs_v = make_rsrc(a.v + hb);
    const int rowst = (int)a.row;
    for (int j = tid; j < MASK_BYTES; j += 512) lmask[j] = (j < t) ? a.key_mask[(int64_t)b * t + j] : 0;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld16(rs_q, qrow < t ? (unsigned)((qrow * rowst + 32 * ks + 8 * g) * 2) : OOB);

    const int nkt = (t + 63) >> 6;
    stage_tile(rs_k, kimg, 0, t, rowst, wave, lane);
    stage_tile(rs_v, vimg, 0, t, rowst, wave, lane);
    const DropCtx dc = drop_ctx(a.rng, a.site, a.pdrop);
    const int64_t prow = (int64_t)b * a.p_batch + ((int64_t)h * t + (qrow < t ? qrow : 0)) * a.tp;
    float m = -3.0e38f, l = 0.f;
    f32x4 oacc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) {
            stage_tile(rs_k, kimg + (buf ^ 1) * TILE, 64 * (kt + 1), t, rowst, wave, lane);
            stage_tile(rs_v, vimg + (buf ^ 1) * TILE, 64 * (kt + 1), t, rowst, wave, lane);
        }
        const unsigned char* ki = kimg + buf * TILE;
        const unsigned char* vi = vimg + buf * TILE;
        // keep-bits of query i16, keys 64kt + 16g .. +15
        const unsigned mybits = dc.on ? drop_bits16(dc, (uint64_t)(prow + 64 * kt + 16 * g)) : 0xFFFFu;
        // ---- S^T tiles: x[T][r] = score of key 64kt + 16T + 4g + r against query i16
        float x[4][4];
        float tmax = -3.0e38f;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const f32x4 s = tile128(ki, 16 * T, qf, lane);
            const int key0 = 64 * kt + 16 * T + 4 * g;
            const unsigned mk = *reinterpret_cast<const unsigned*>(lmask + key0);        // key0 % 4 == 0; bytes >= t are 0
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[T][r] = mask_score(s[r], a.alpha, (mk >> (8 * r)) & 0xFFu, key0 + r < t);
                tmax = fmaxf(tmax, x[T][r]);
            }
        }
        tmax = xor16_32_max(tmax);
        const float m_new = fmaxf(m, tmax);
        const float corr = __expf(m - m_new);
        l *= corr;
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[d][r] *= corr;
        m = m_new;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const unsigned bT = (unsigned)__shfl((int)mybits, i16 + 16 * T, 64) >> (4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(x[T][r] - m);
                l += pv;
                x[T][r] = ((bT >> r) & 1u) ? pv * dc.scale : 0.f;
            }
        }
        // ---- O^T += V^T P^T: k-step kp covers the keys of score tiles 2kp, 2kp+1 (in the accumulators' own order)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const bf16x8 pb = pack8(x[2 * kp], x[2 * kp + 1]);
#pragma unroll
            for (int d = 0; d < 8; ++d)
                oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(vi, 32 * kp, 32 * kp + 16, d, lane), pb, oacc[d], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    l = xor16_32_sum(l);
    if (qrow < t) {
        const float inv = 1.f / l;
        bf16_t* orow = a.O + (int64_t)b * a.o_batch + (int64_t)qrow * a.o_row + (int64_t)h * a.head;
#pragma unroll
        for (int d = 0; d < 8; ++d) {           // oacc[d][r] = O[qrow][16d + 4g + r]
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(oacc[d][r] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * d + 4 * g) = o;
        }
        if (g == 0) *reinterpret_cast<float2*>(a.stats + (((int64_t)b * a.H + h) * t + qrow) * 2) = make_float2(m, l);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ aux)
// Per wave 16 queries; recomputes S^T and dPd^T = V dO^T per 64-key tile, dS^T = P (dPd keep c - delta) (0 at masked keys:
// masked_fill's backward), dQ^T += K^T dS^T.  Also writes aux = {m, 1/l, delta, 0} per query for the dK/dV kernel.
__global__ __launch_bounds__(512, 4) void flash_bwd_dq_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z, t = a.t;
    const int qrow = blockIdx.x * 128 + wave * 16 + i16;
    unsigned char* kimg = smem;
    unsigned char* vimg = smem + 2 * TILE;
    unsigned char* lmask = smem + 4 * TILE;
    const int64_t hb = (int64_t)b * a.batch + (int64_t)h * a.head;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hb), rs_k = make_rsrc(a.k + hb), rs_v = make_rsrc(a.v + hb);
    const __amdgpu_buffer_rsrc_t rs_do = make_rsrc(a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.head);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.O + (int64_t)b * a.o_batch + (int64_t)h * a.head);
    const int rowst = (int)a.row;
    for (int j = tid; j < MASK_BYTES; j += 512) lmask[j] = (j < t) ? a.key_mask[(int64_t)b * t + j] : 0;

    bf16x8 qf[4], dof[4];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = ld16(rs_q, qrow < t ? (unsigned)((qrow * rowst + 32 * ks + 8 * g) * 2) : OOB);
        dof[ks] = ld16(rs_do, qrow < t ? (unsigned)((qrow * (int)a.do_row + 32 * ks + 8 * g) * 2) : OOB);
        const bf16x8 of = ld16(rs_o, qrow < t ? (unsigned)((qrow * (int)a.o_row + 32 * ks + 8 * g) * 2) : OOB);
#pragma unroll
        for (int c = 0; c < 8; ++c) delta += (float)dof[ks][c] * (float)of[c];
    }
    delta = xor16_32_sum(delta);
    float m = 0.f, linv = 0.f;
    if (qrow < t) {
        const float2 st = *reinterpret_cast<const float2*>(a.stats + (((int64_t)b * a.H + h) * t + qrow) * 2);
        m = st.x;
        linv = 1.f / st.y;
        if (g == 0) *reinterpret_cast<float4*>(a.aux + (((int64_t)b * a.H + h) * t + qrow) * 4) = make_float4(m, linv, delta, 0.f);
    }

    const int nkt = (t + 63) >> 6;
    stage_tile(rs_k, kimg, 0, t, rowst, wave, lane);
    stage_tile(rs_v, vimg, 0, t, rowst, wave, lane);
    const DropCtx dc = drop_ctx(a.rng, a.site, a.pdrop);
    const int64_t prow = (int64_t)b * a.p_batch + ((int64_t)h * t + (qrow < t ? qrow : 0)) * a.tp;
    f32x4 dqacc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) dqacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) {
            stage_tile(rs_k, kimg + (buf ^ 1) * TILE, 64 * (kt + 1), t, rowst, wave, lane);
            stage_tile(rs_v, vimg + (buf ^ 1) * TILE, 64 * (kt + 1), t, rowst, wave, lane);
        }
        const unsigned char* ki = kimg + buf * TILE;
        const unsigned char* vi = vimg + buf * TILE;
        const unsigned mybits = dc.on ? drop_bits16(dc, (uint64_t)(prow + 64 * kt + 16 * g)) : 0xFFFFu;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            float ds[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int T = 2 * kp + u;
                const f32x4 s = tile128(ki, 16 * T, qf, lane);
                const f32x4 dp = tile128(vi, 16 * T, dof, lane);
                const int key0 = 64 * kt + 16 * T + 4 * g;
                const unsigned mk = *reinterpret_cast<const unsigned*>(lmask + key0);
                const unsigned bT = (unsigned)__shfl((int)mybits, i16 + 16 * T, 64) >> (4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned mb = (mk >> (8 * r)) & 0xFFu;
                    const float pn = __expf(mask_score(s[r], a.alpha, mb, key0 + r < t) - m) * linv;
                    const float kc = ((bT >> r) & 1u) ? dc.scale : 0.f;
                    ds[u][r] = mb != 0 ? pn * (dp[r] * kc - delta) : 0.f;
                }
            }
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int d = 0; d < 8; ++d)
                dqacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(ki, 32 * kp, 32 * kp + 16, d, lane), dsb, dqacc[d], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (qrow < t) {
        bf16_t* drow = a.dq + (int64_t)b * a.g_batch + (int64_t)qrow * a.g_row + (int64_t)h * a.head;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(dqacc[d][r] * a.alpha);
            *reinterpret_cast<bf16x4*>(drow + 16 * d + 4 * g) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Per wave 16 keys (K and V fragments in registers); streams Q, dO and aux tiles of 64 queries.  S = Q K^T and dPd = dO V^T in
// the layout D[query 4g + r][key i16]; dV^T += dO^T Pd, dK^T += Q^T dS (contraction over the queries, operands as above).
__global__ __launch_bounds__(512, 2) void flash_bwd_dkv_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z, t = a.t;
    const int kb0 = blockIdx.x * 128 + wave * 16;
    const int key = kb0 + i16;
    unsigned char* qimg = smem;                         // [2][TILE]
    unsigned char* doimg = smem + 2 * TILE;             // [2][TILE]
    unsigned char* auximg = smem + 4 * TILE;            // [2][AUX_BYTES]
    const int64_t hb = (int64_t)b * a.batch + (int64_t)h * a.head;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hb), rs_k = make_rsrc(a.k + hb), rs_v = make_rsrc(a.v + hb);
    const __amdgpu_buffer_rsrc_t rs_do = make_rsrc(a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.head);
    const __amdgpu_buffer_rsrc_t rs_aux = make_rsrc(a.aux + (((int64_t)b * a.H + h) * t) * 4);
    const int rowst = (int)a.row, dorow = (int)a.do_row;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = ld16(rs_k, key < t ? (unsigned)((key * rowst + 32 * ks + 8 * g) * 2) : OOB);
        vf[ks] = ld16(rs_v, key < t ? (unsigned)((key * rowst + 32 * ks + 8 * g) * 2) : OOB);
    }
    const bool kvalid = key < t;
    const unsigned mkb = kvalid ? a.key_mask[(int64_t)b * t + key] : 0;

    auto stage = [&](int qt, int buf) {
        stage_tile(rs_q, qimg + buf * TILE, 64 * qt, t, rowst, wave, lane);
        stage_tile(rs_do, doimg + buf * TILE, 64 * qt, t, dorow, wave, lane);
        if (wave == 0) {
            const int q = 64 * qt + lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_aux, (lds_void_t*)(auximg + buf * AUX_BYTES), 16,
                                                     (int)(q < t ? (unsigned)(q * 16) : OOB), 0, 0, 0);
        }
    };
    const int nqt = (t + 63) >> 6;
    stage(0, 0);
    const DropCtx dc = drop_ctx(a.rng, a.site, a.pdrop);
    const int64_t pbase = (int64_t)b * a.p_batch + (int64_t)h * t * a.tp + kb0;
    f32x4 dvacc[8], dkacc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) { dvacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dkacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int qt = 0; qt < nqt; ++qt) {
        const int buf = qt & 1;
        if (qt + 1 < nqt) stage(qt + 1, buf ^ 1);
        const unsigned char* qi = qimg + buf * TILE;
        const unsigned char* di = doimg + buf * TILE;
        const float4* aux = reinterpret_cast<const float4*>(auximg + buf * AUX_BYTES);
        // keep-bits of the wave's 16 keys for query 64qt + 16(i16>>2) + 4g + (i16&3): the element (T, r) = (i16>>2, i16&3) of
        // this lane's row of 16 lanes
        unsigned mybits = 0xFFFFu;
        if (dc.on) {
            const int qq = 64 * qt + 16 * (i16 >> 2) + 4 * g + (i16 & 3);
            mybits = drop_bits16(dc, (uint64_t)(pbase + (int64_t)(qq < t ? qq : 0) * a.tp));
        }
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            float pd[2][4], ds[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int T = 2 * kp + u;
                const f32x4 s = tile128(qi, 16 * T, kf, lane);
                const f32x4 dp = tile128(di, 16 * T, vf, lane);
                unsigned kb[4];
                if (T == 0) { kb[0] = row_share<0>(mybits); kb[1] = row_share<1>(mybits); kb[2] = row_share<2>(mybits); kb[3] = row_share<3>(mybits); }
                else if (T == 1) { kb[0] = row_share<4>(mybits); kb[1] = row_share<5>(mybits); kb[2] = row_share<6>(mybits); kb[3] = row_share<7>(mybits); }
                else if (T == 2) { kb[0] = row_share<8>(mybits); kb[1] = row_share<9>(mybits); kb[2] = row_share<10>(mybits); kb[3] = row_share<11>(mybits); }
                else { kb[0] = row_share<12>(mybits); kb[1] = row_share<13>(mybits); kb[2] = row_share<14>(mybits); kb[3] = row_share<15>(mybits); }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 ax = aux[16 * T + 4 * g + r];                         // {m, 1/l, delta, -}; zeros for q >= t
                    const float pn = kvalid ? __expf(mask_score(s[r], a.alpha, mkb, true) - ax.x) * ax.y : 0.f;
                    const float kc = ((kb[r] >> i16) & 1u) ? dc.scale : 0.f;
                    pd[u][r] = pn * kc;
                    ds[u][r] = mkb != 0 ? pn * (dp[r] * kc - ax.z) : 0.f;
                }
            }
            const bf16x8 pdb = pack8(pd[0], pd[1]);
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                dvacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(di, 32 * kp, 32 * kp + 16, d, lane), pdb, dvacc[d], 0, 0, 0);
                dkacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(qi, 32 * kp, 32 * kp + 16, d, lane), dsb, dkacc[d], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (kvalid) {
        const int64_t off = (int64_t)b * a.g_batch + (int64_t)key * a.g_row + (int64_t)h * a.head;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            bf16x4 ov, ok;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ov[r] = (bf16_t)dvacc[d][r]; ok[r] = (bf16_t)(dkacc[d][r] * a.alpha); }
            *reinterpret_cast<bf16x4*>(a.dv + off + 16 * d + 4 * g) = ov;
            *reinterpret_cast<bf16x4*>(a.dk + off + 16 * d + 4 * g) = ok;
        }
    }
}

int check_common(const char* who, const void* q, const void* k, const void* v, int64_t row, int64_t batch, int head, int B, int H, int t,
                 int tp, int64_t p_batch, float p, const uint64_t* rng) {
    FS2_REQUIRE(q && k && v, "%s: null argument", who);
    FS2_REQUIRE(t > 0 && t <= MASK_BYTES && tp == (t + 7) / 8 * 8, "%s: need 0 < t <= 1024 and tp = roundup8(t) (t=%d tp=%d)", who, t, tp);
    FS2_REQUIRE(B > 0 && H > 0 && B <= 65535 && H <= 65535, "%s: bad B/H", who);
    FS2_REQUIRE(row % 8 == 0 && batch % 8 == 0 && head % 8 == 0 && p_batch % 8 == 0, "%s: strides must be multiples of 8 elements", who);
    FS2_REQUIRE(fs2_aligned16(q) && fs2_aligned16(k) && fs2_aligned16(v), "%s: pointers must be 16-byte aligned", who);
    FS2_REQUIRE((int64_t)(t + 64) * row * 2 < 0x7FFFFFF0LL, "%s: one (batch, head) slice exceeds 2 GiB", who);
    FS2_REQUIRE(p >= 0.f && p < 1.f && (p == 0.f || rng != nullptr), "%s: bad dropout arguments", who);
    return FS2_OK;
}

}  // namespace

extern "C" int fs2_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, int64_t batch_stride, int head_stride,
                                  const uint8_t* key_mask, void* o_out, int64_t o_row_stride, int64_t o_batch_stride, float* stats,
                                  int64_t p_batch_stride, int B, int H, int t, int tp, float alpha, float p, const uint64_t* rng,
                                  uint32_t site, void* stream) {
    const int rc = check_common("fs2_flash_attn_fwd", q, k, v, row_stride, batch_stride, head_stride, B, H, t, tp, p_batch_stride, p, rng);
    if (rc != FS2_OK) return rc;
    FS2_REQUIRE(key_mask && o_out && stats, "fs2_flash_attn_fwd: null argument");
    FS2_REQUIRE(o_row_stride % 4 == 0 && o_batch_stride % 4 == 0 && fs2_aligned16(o_out), "fs2_flash_attn_fwd: output rows must be 8-byte aligned");
    FlashArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.row = row_stride; a.batch = batch_stride; a.head = head_stride;
    a.key_mask = key_mask; a.O = (bf16_t*)o_out; a.o_row = o_row_stride; a.o_batch = o_batch_stride; a.stats = stats; a.p_batch = p_batch_stride;
    a.H = H; a.t = t; a.tp = tp; a.alpha = alpha; a.pdrop = p; a.rng = rng; a.site = site;
    const int lds = 4 * TILE + MASK_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(flash_fwd_k, dim3((t + 127) / 128, H, B), dim3(512), lds, (hipStream_t)stream, a);
    FS2_CHECK_LAUNCH("fs2_flash_attn_fwd");
    return FS2_OK;
}

extern "C" int fs2_flash_attn_bwd(const void* q, const void* k, const void* v, int64_t row_stride, int64_t batch_stride, int head_stride,
                                  const uint8_t* key_mask, const void* o_saved, int64_t o_row_stride, int64_t o_batch_stride,
                                  const void* d_out, int64_t do_row_stride, int64_t do_batch_stride, const float* stats, float* aux,
                                  void* dq, void* dk, void* dv, int64_t g_row_stride, int64_t g_batch_stride, int64_t p_batch_stride,
                                  int B, int H, int t, int tp, float alpha, float p, const uint64_t* rng, uint32_t site, void* stream) {
    const int rc = check_common("fs2_flash_attn_bwd", q, k, v, row_stride, batch_stride, head_stride, B, H, t, tp, p_batch_stride, p, rng);
    if (rc != FS2_OK) return rc;
    FS2_REQUIRE(key_mask && o_saved && d_out && stats && aux && dq && dk && dv, "fs2_flash_attn_bwd: null argument");
    FS2_REQUIRE(o_row_stride % 8 == 0 && o_batch_stride % 8 == 0 && do_row_stride % 8 == 0 && do_batch_stride % 8 == 0 && g_row_stride % 4 == 0 &&
                    g_batch_stride % 4 == 0, "fs2_flash_attn_bwd: strides must be multiples of 8 elements (gradients: 4)");
    FS2_REQUIRE(fs2_aligned16(o_saved) && fs2_aligned16(d_out) && fs2_aligned16(aux) && fs2_aligned16(dq) && fs2_aligned16(dk) && fs2_aligned16(dv),
                "fs2_flash_attn_bwd: pointers must be 16-byte aligned");
    FS2_REQUIRE((int64_t)(t + 64) * do_row_stride * 2 < 0x7FFFFFF0LL && (int64_t)(t + 64) * o_row_stride * 2 < 0x7FFFFFF0LL,
                "fs2_flash_attn_bwd: one (batch, head) slice exceeds 2 GiB");
    FlashArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.row = row_stride; a.batch = batch_stride; a.head = head_stride;
    a.key_mask = key_mask; a.O = (bf16_t*)const_cast<void*>(o_saved); a.o_row = o_row_stride; a.o_batch = o_batch_stride;
    a.stats = const_cast<float*>(stats); a.p_batch = p_batch_stride;
    a.H = H; a.t = t; a.tp = tp; a.alpha = alpha; a.pdrop = p; a.rng = rng; a.site = site;
    a.dO = (const bf16_t*)d_out; a.do_row = do_row_stride; a.do_batch = do_batch_stride; a.aux = aux;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.g_row = g_row_stride; a.g_batch = g_batch_stride;
    const int lds_q = 4 * TILE + MASK_BYTES, lds_kv = 4 * TILE + 2 * AUX_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dq_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        attr_set = true;
    }
    const dim3 grid((t + 127) / 128, H, B);
    hipLaunchKernelGGL(flash_bwd_dq_k, grid, dim3(512), lds_q, (hipStream_t)stream, a);
    hipLaunchKernelGGL(flash_bwd_dkv_k, grid, dim3(512), lds_kv, (hipStream_t)stream, a);
    FS2_CHECK_LAUNCH("fs2_flash_attn_bwd");
    return FS2_OK;
}

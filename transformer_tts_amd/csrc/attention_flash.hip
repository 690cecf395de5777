// Flash-style attention for gfx950 (bf16, d_k = 128): attention() of the reference (Models/modules.py:7-21) and its backward
// WITHOUT the (t x t) probability tensors in HBM -- the mode the trainer runs when the attention maps are not requested
// (hp.return_attn = False).  Forward keeps per query row the running maximum and the sum of exponentials; backward recomputes
// the probabilities from Q, K and those two numbers and regenerates the dropout mask from the same Philox counters as every
// other kernel of the library (element offset of P[b, h, q, key] in the (B, [layers], H, t, tp) layout, >> 3), so this path and
// the LDS-strip path (attention.hip) draw IDENTICAL masks and differ only by rounding.
//
// Common structure (all three kernels): a workgroup owns 128 rows of one (batch, head) -- queries in the forward and dQ kernels,
// keys in the dK/dV kernel -- as 8 waves x 16 rows (forward, dK/dV) or 4 waves x 2 x 16 rows (dQ), held as MFMA fragments in registers; the other side streams through
// LDS in tiles of 64 rows x 128 columns by LDS-DMA (buffer_load_dwordx4 ... lds), double buffered, in the dual-use image of the
// CDNA4 guide (T10: 256-byte rows, 16-byte chunk c of row r at c ^ f(r); f below), which serves both the
// row reads (ds_read_b128: operand rows) and the transposed reads (ds_read_b64_tr_b16: the same tile as the k-major operand of
// the second product).  Score tiles are computed TRANSPOSED relative to the product that consumes them, so that an accumulator
// tile is already the next MFMA's operand: with D = A B, lane (i16, g) holds D[4g + r][i16]; two 16-row tiles T, T+1 give the
// 8 k-values {16T + 4g + r} u {16(T+1) + 4g + r} of column i16 -- the B-operand fragment of a 32-deep k-step whose k order is
// that permutation; the A operand of the same k-step is gathered in the same order by two transposed reads of rows 16T + 4g + q
// and 16(T+1) + 4g + q.  No LDS round trip for the probabilities.
//
// Dropout: one Philox call covers 8 consecutive keys of one query.  A wave's 16 x 64 tile needs 128 calls = 2 per lane; each
// lane turns its two calls into 16 keep-bits and the tile's owners fetch them with one cross-lane move per 4 (forward, dQ:
// ds_bpermute) or 1 (dK/dV: DPP row_share) elements.  The forward kernel is the only one that runs Philox (the integer
// multiplies are quarter rate and the loop is VALU-bound): it stashes the keep-bits, ONE BIT per probability (16-bit words
// [b][h][key tile][query][16-key group]), and the two backward kernels read them back.  The factor 1/(1-p) is applied to the
// accumulators at the end, not per element.
//
// Masks: the workgroup finds kfull = number of leading unmasked keys and kmax = last unmasked key + 1 of its batch row.  Tiles
// below kfull run without any mask arithmetic; tiles at or beyond kmax are skipped: every key in them is masked, its
// probability is exp(-1e4 - m) = 0 in fp32 as soon as one unmasked key exists (kmax > 0; otherwise nothing is skipped and the
// uniform distribution of the reference comes out), so O, dQ receive nothing from them and their dK, dV rows are zero.
// Softmax in the exp2 domain: p = exp2(s * (alpha log2 e) - m * (alpha log2 e)) is one fma + v_exp_f32.
#include <stdlib.h>
#include "fs2_common.h"

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
// rows of 128 bf16 at a stride of row_stride elements: a 16-byte read that starts past row rows-1 returns zeros
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_rows(const void* p, int rows, int row_stride) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (rows - 1) * row_stride * 2 + 256, 0x00020000);
}
__device__ __forceinline__ bf16x8 ld16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
// chunk swizzle of the 64-row x 256-byte images: chunk c of row r sits at c ^ ((r & 7) << 1).  With 64 banks of 4 bytes a row is one
// bank period, so a read is conflict-free when the lanes of a service group touch distinct 16-byte chunks: the transposed reads
// (groups of 32 lanes = 8 consecutive rows x one aligned 32-byte pair) get pair index ct ^ (r & 7), all different; the row
// reads (groups of 16 lanes = 16 rows, half of them on chunk c and half on c ^ 1) get (c ^ 2(r & 7)) and its odd neighbour for
// the row 8 further on.  (The T10 image (b) swizzle ((r&3)<<2 | (r>>2)&3) is 2-way on both access patterns of these kernels:
// 40 % of the LDS cycles of the dK/dV kernel were conflicts.)
__device__ __forceinline__ int img_f(int r) { return (r & 7) << 1; }
__device__ __forceinline__ int img_off(int row, int ch) { return row * 256 + ((ch ^ img_f(row)) << 4); }

// LDS byte addresses of this lane's fragment reads inside an image at offset 0; everything else (which image, which buffer,
// which 16-row tile) is a compile-time constant that lands in the instruction's offset field.  img_off(r0 + x, ch) =
// 256 r0 + img_off(x, ch) for r0 % 16 == 0 because the swizzle only looks at the row's low 4 bits.
struct FragAddr {
    unsigned row[4];        // operand rows i16 (+ r0), k-step ks: chunk 4ks + g
    unsigned tr[8];         // transposed operand, column tile ct: row 4g + q (+ rA / rB), chunk 2ct + (pp >> 1), half pp & 1
};
__device__ __forceinline__ FragAddr frag_addr(int lane, unsigned lds0) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    FragAddr fa;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa.row[ks] = lds0 + img_off(i16, 4 * ks + g);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) fa.tr[ct] = lds0 + img_off(4 * g + q, 2 * ct + (pp >> 1)) + 8 * (pp & 1);
    return fa;
}
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
template <int N> struct IC { static constexpr int value = N; };

// operand rows r0 + i16 of the image at byte offset OFF, k-step ks (32 columns)
template <int OFF> __device__ __forceinline__ bf16x8 row_frag(const FragAddr& fa, int ks) {
    return *reinterpret_cast<lds_bf16x8*>(fa.row[ks] + OFF);
}
// transposed operand: column 16ct + i16 of the rows {rA + 4g + q} u {rA + 16 + 4g + q}, q = 0..3; OFF = image offset + 256 rA
template <int OFF> __device__ __forceinline__ bf16x8 tr_frag(const FragAddr& fa, int ct) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(fa.tr[ct] + OFF));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(fa.tr[ct] + OFF + 16 * 256));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}
// The same through inline assembly, for the phases that run while the next tile's LDS-DMA is in flight: the builtin carries no
// address information, so hipcc puts an `s_waitcnt vmcnt(0)` in front of the first transposed read that follows a DMA issue (it must
// assume the DMA writes what the read reads) -- the prefetch then has to land in the middle of the tile it was meant to hide behind.
// Issue and wait are separate statements so that the reads of fragment d+1 can be in flight while fragment d feeds its MFMA: the
// wait takes the fragment as an in/out operand, which is what orders the consumer after it.  (LDS operations return in order, so
// lgkmcnt(2) = "everything but the two reads issued last".)
struct TrFrag { s16x4 lo, hi; };
template <int OFF> __device__ __forceinline__ void tr_issue(const FragAddr& fa, int ct, TrFrag& f) {
    if constexpr (OFF + 4096 < 65536) {
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                     : "=&v"(f.lo), "=&v"(f.hi) : "v"(fa.tr[ct]), "n"(OFF), "n"(OFF + 4096));
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096"
                     : "=&v"(f.lo), "=&v"(f.hi) : "v"(fa.tr[ct] + OFF));
    }
}
template <int PENDING> __device__ __forceinline__ bf16x8 tr_wait(TrFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(PENDING));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = f.lo; u.s.hi = f.hi;
    return u.v;
}
// 16 x 16 product tile over the 128 columns: rows r0.. of the LDS image (OFF = image offset + 256 r0) against the register
// fragments bf (B operand)
template <int OFF> __device__ __forceinline__ f32x4 tile128(const FragAddr& fa, const bf16x8 (&bf)[4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<OFF>(fa, ks), bf[ks], acc, 0, 0, 0);
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

constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED_NAT = -1e4f;     // masked_fill value of the reference (Models/modules.py:12-14), scaled-score domain
constexpr float NOKEY = -3.0e38f;       // keys that do not exist: exp2() = 0

// total of each 16-lane row in all of its lanes (quad xor 1, quad xor 2, row rotate 4 and 8)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x128>(v);
    return v;
}
// bias gradient of a projection = column sums of its gradient rows: acc[d][r] is this lane's value for column 16d + 4g + r of one
// row (the lane's i16 selects the row); rows of the workgroup are summed through DPP, then LDS, then one global atomic per
// column per workgroup.  lds: 128 floats nobody else uses any more; NT threads; ends with a barrier-free tail.
template <int NT>
__device__ __forceinline__ void block_colsum(const f32x4 (&acc)[8], float factor, float* lds, float* out, int tid, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
    if (tid < 128) lds[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float tot = row16_sum(acc[d][r]);
            if (i16 == 0) atomicAdd(&lds[16 * d + 4 * g + r], tot * factor);
        }
    __syncthreads();
    if (tid < 128) atomicAdd(out + tid, lds[tid]);
}

struct FlashArgs {
    const bf16_t *q, *k, *v;            // rows of one head: 128 contiguous bf16 at base + b*batch + i*row + h*head
    int64_t row, batch;                 // element strides of q / k / v (one fused projection tensor)
    int head;
    const uint8_t* key_mask;            // (B, t)
    const int32_t* kinfo;               // optional (B, 3): {kfull, kmax, the batch row of rank b by kmax} (fs2_flash_attn_mask_info), or nullptr
    bf16_t* O;                          // attention output (written by forward, read by backward), rows at O + b*o_batch + i*o_row + h*head
    int64_t o_row, o_batch;
    float* stats;                       // (B, H, t, 2): {row maximum of the masked scaled scores, sum of exponentials}
    uint16_t* keep;                     // (B, H, nkt, t, 4) keep-bits of the dropout: bit j of word [b][h][kt][q][g] = key 64kt + 16g + j
    int64_t p_batch;                    // batch stride of the virtual P tensor (dropout counters)
    int B, H, t, tp, nkt;
    float alpha, pdrop;
    const uint64_t* rng;
    uint32_t site;
    // backward
    const bf16_t* dO;                   // rows at dO + b*do_batch + i*do_row + h*head
    int64_t do_row, do_batch;
    float* aux;                         // (B, H, t, 4) workspace: {-m log2 e, 1/l, delta = rowsum(dO * O), kmax bits}: dQ kernel -> dK/dV kernel
    bf16_t *dq, *dk, *dv;               // rows at d? + b*g_batch + i*g_row + h*head
    int64_t g_row, g_batch;
    float *dbq, *dbk, *dbv;             // optional bias gradients of the three projections (H*128 floats each): += column sums of dq / dk / dv
};

// Work item of this workgroup.  Workgroups are dealt round-robin to the 8 XCDs; the row blocks of one (batch, head) pair -- which
// stream the same K/V (or Q/dO) tiles -- are given to ONE XCD (one L2), and the pairs go round-robin over the XCDs so that a
// batch sorted by length does not leave one XCD with all the long sequences.  Grid = 8 * ceil(B H / 8) * nblk.
template <int BLK = 128>
__device__ __forceinline__ bool flash_item(const FlashArgs& a, int& blk, int& h, int& b) {
    const int nblk = (a.t + BLK - 1) / BLK;
    const int slot = (int)(blockIdx.x >> 3);
    const int pair = (int)(blockIdx.x & 7) + 8 * (slot / nblk);
    if (pair >= a.B * a.H) return false;
    blk = slot % nblk;
    h = pair % a.H;
    b = pair / a.H;
    // Longest sequences first (fs2_flash_attn_mask_info ranks the batch rows by their last unmasked key): a forward / dQ workgroup
    // costs as many key tiles as its row has, there are 1.5 workgroups per slot at config 2, and in batch order a long row could
    // start in the second half-round.  The rank list is dealt round-robin to the XCDs like the pairs themselves.
    if (a.kinfo != nullptr) b = a.kinfo[3 * b + 2];
    return true;
}

// kfull = number of leading unmasked keys, kmax = last unmasked key + 1 of one batch row (optionally copies the row to LDS,
// zero padded to MASK_BYTES).  red: two LDS words.  Ends with a barrier.
template <int NT = 512>
__device__ __forceinline__ void scan_mask(const uint8_t* km_row, int t, unsigned char* lmask, int* red, int tid) {
    if (tid == 0) { red[0] = t; red[1] = 0; }
    __syncthreads();
    for (int j = tid; j < MASK_BYTES; j += NT) {
        const unsigned char mk = j < t ? km_row[j] : 0;
        if (lmask) lmask[j] = mk;
        if (j < t) {
            if (mk) atomicMax(&red[1], j + 1);
            else atomicMin(&red[0], j);
        }
    }
    __syncthreads();
}

// the same with kfull / kmax already known (fs2_flash_attn_mask_info ran once for the whole stack): only the LDS copy of the mask
// row, no atomics and no barrier of its own (the prologue's closing barrier orders the copy before its first reader)
template <int NT = 512>
__device__ __forceinline__ void mask_setup(const FlashArgs& a, int b, int t, unsigned char* lmask, int* red, int tid, int& kfull, int& kmax) {
    if (a.kinfo != nullptr) {
        const uint8_t* km_row = a.key_mask + (int64_t)b * t;
        for (int j = tid; j < MASK_BYTES; j += NT) lmask[j] = j < t ? km_row[j] : 0;
        kfull = a.kinfo[3 * b];
        kmax = a.kinfo[3 * b + 1];
    } else {
        scan_mask<NT>(a.key_mask + (int64_t)b * t, t, lmask, red, tid);
        kfull = red[0];
        kmax = red[1];
    }
}
__global__ __launch_bounds__(512) void flash_mask_info_k(const uint8_t* __restrict__ key_mask, int t, int32_t* __restrict__ info) {
    __shared__ int red[2];
    scan_mask<512>(key_mask + (int64_t)blockIdx.x * t, t, nullptr, red, threadIdx.x);
    if (threadIdx.x < 2) info[3 * blockIdx.x + threadIdx.x] = red[threadIdx.x];
}
// info[3 r + 2] = the batch row with the r-th largest kmax (ties: lower index first); one thread per row, B comparisons each
__global__ __launch_bounds__(256) void flash_order_k(int B, int32_t* __restrict__ info) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        const int ki = info[3 * i + 1];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int kj = info[3 * j + 1];
            rank += (kj > ki || (kj == ki && j < i)) ? 1 : 0;
        }
        info[3 * rank + 2] = i;
    }
}

// stage one 64-row tile with NW waves: instruction i of wave w covers tile rows 4*(NW i + w) .. +3; lane -> row lane>>4, logical
// chunk (lane&15) ^ f(row); rows >= t use an out-of-range offset (zeros)
template <int NW = 8>
__device__ __forceinline__ void stage_tile(const __amdgpu_buffer_rsrc_t rs, unsigned char* dst, int row0, int row_stride, int wave,
                                           unsigned voff0) {
    __builtin_assume(wave >= 0 && wave < NW);      // the writes stay inside this one image (alias analysis of the LDS reads that follow)
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + 1024 * (NW * i + wave)), 16,
                                                 (int)(voff0 + (unsigned)((row0 + 4 * NW * i) * row_stride * 2)), 0, 0, 0);
}
// the lane's share of that: row 4 wave + lane/16 of the first instruction, chunk (lane&15) ^ f(row) -- the later instructions are
// 4 NW rows further on (same low row bits, same chunk), so ONE per-lane offset serves a whole kernel: the scalar part carries the
// tile and the instruction.  Rows >= t are cut off by the descriptor (make_rsrc_rows): no per-lane compare, nothing to spill.
__device__ __forceinline__ unsigned stage_voff(int row_stride, int wave, int lane) {
    const int r = 4 * wave + (lane >> 4);
    return (unsigned)((r * row_stride + (((lane & 15) ^ img_f(r)) << 3)) * 2);
}

__device__ __forceinline__ float and_mask(float v, int msk) { return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & msk); }
__device__ __forceinline__ int keep_mask(unsigned bits, unsigned pos) { return __builtin_amdgcn_sbfe((int)bits, pos, 1u); }   // bit -> 0 / ~0

// dQ kernel shape: workgroup = 4 waves x 32 queries (two 16-query sub-tiles per wave: every K / V fragment read from LDS feeds two
// MFMAs; two independent workgroups per CU).  The forward kernel measured faster as 8 waves x 16 queries (4 waves per SIMD:
// 55.7 vs 60.6 us per launch averaged over the model's eight attention layers).
constexpr int FQ_WAVES = 4, FQ_THREADS = 256;

// masked (workgroup-uniform): the tile holds masked or non-existent keys -- keys with mask 0 get the raw value whose scaled
// score is -1e4 (masked_fill), keys >= t get NOKEY.  The MFMA work is common to both cases; only these fix-ups sit under a branch.
__device__ __forceinline__ void mask_fix(float (&v)[4], unsigned mk, int key0, int t, float masked_raw) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = ((mk >> (8 * r)) & 0xFFu) != 0 ? v[r] : masked_raw;
        v[r] = (key0 + r < t) ? v[r] : NOKEY;
    }
}

// raw scores S^T of one 64-key tile (image at byte offset KOFF) against the wave's 16 queries: x[T][r] = key 64kt + 16T + 4g + r,
// query i16.  masked (workgroup-uniform): the tile holds masked or non-existent keys -- keys with mask 0 get the raw value whose
// scaled score is -1e4 (masked_fill), keys >= t get NOKEY.  The MFMA work is common to both cases, only the element-wise fix-ups
// sit under the branch (two instantiated copies of a whole step cost ~100 spilled registers in the dQ kernel).
template <int KOFF>
__device__ __forceinline__ void score_tiles(const bool masked, const FragAddr& fa, const bf16x8 (&qf)[4], const unsigned char* lmask, int kt,
                                            int t, float masked_raw, int lane, float (&x)[4][4], float& tmax) {
    const int g = lane >> 4;
    auto one = [&](auto TC) {
        constexpr int T = decltype(TC)::value;
        const f32x4 s = tile128<KOFF + 4096 * T>(fa, qf);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[T][r] = s[r];
        if (masked) {
            const int key0 = 64 * kt + 16 * T + 4 * g;
            const unsigned mk = *reinterpret_cast<const unsigned*>(lmask + key0);        // key0 % 4 == 0; bytes >= t are 0
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[T][r] = ((mk >> (8 * r)) & 0xFFu) != 0 ? x[T][r] : masked_raw;
                x[T][r] = (key0 + r < t) ? x[T][r] : NOKEY;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, x[T][r]);
    };
    one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{});
}

// ------------------------------------------------------------------------------------------------ forward
// O = dropout(softmax(mask(alpha Q K^T))) V, stats = {m, l}.  Wave: 16 queries (columns of the transposed score tiles).
// LDS: K images [2][TILE] at 0, V images [2][TILE] at 2 TILE, key mask, two reduction words.
// DROP: 0 no dropout; 1 draw the keep-bits (Philox) and stash them; 2 read bits that fs2_flash_attn_keep_bits wrote beforehand
template <int DROP>
__global__ __launch_bounds__(512, 4) void flash_fwd_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int blk, h, b;
    if (!flash_item(a, blk, h, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = a.t;
    const int qrow = blk * 128 + wave * 16 + i16;
    unsigned char* kimg = smem;                       // [2][TILE]
    unsigned char* vimg = smem + 2 * TILE;            // [2][TILE]
    unsigned char* lmask = smem + 4 * TILE;
    int* red = reinterpret_cast<int*>(smem + 4 * TILE + MASK_BYTES);
    const FragAddr fa = frag_addr(lane, (unsigned)(uintptr_t)(lds_void_t*)smem);
    const int64_t hb = (int64_t)b * a.batch + (int64_t)h * a.head;
    const int rowst = (int)a.row;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hb), rs_k = make_rsrc_rows(a.k + hb, t, rowst), rs_v = make_rsrc_rows(a.v + hb, t, rowst);
    const unsigned voff0 = stage_voff(rowst, wave, lane);
    stage_tile(rs_k, kimg, 0, rowst, wave, voff0);
    stage_tile(rs_v, vimg, 0, rowst, wave, voff0);
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld16(rs_q, qrow < t ? (unsigned)((qrow * rowst + 32 * ks + 8 * g) * 2) : OOB);
    int kfull, kmax;
    mask_setup<512>(a, b, t, lmask, red, tid, kfull, kmax);
    const int nkt = kmax > 0 ? (kmax + 63) >> 6 : (t + 63) >> 6;

    const DropCtx dc = drop_ctx(a.rng, a.site, a.pdrop);
    const int64_t prow = (int64_t)b * a.p_batch + ((int64_t)h * t + (qrow < t ? qrow : 0)) * a.tp;
    uint16_t* keep = a.keep + ((((int64_t)b * a.H + h) * a.nkt) * t + (qrow < t ? qrow : 0)) * 4 + g;
    const float c2 = a.alpha * LOG2E, masked_raw = MASKED_NAT / a.alpha;
    float m = NOKEY, l = 0.f;            // m: running maximum of the RAW scores
    f32x4 oacc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned bits_next = DROP == 2 ? keep[0] : 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto tile = [&](const int kt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr int KOFF = BUF * TILE, VOFF = 2 * TILE + BUF * TILE;
        unsigned mybits = bits_next;
        if (kt + 1 < nkt) {
            stage_tile(rs_k, kimg + (BUF ^ 1) * TILE, 64 * (kt + 1), rowst, wave, voff0);
            stage_tile(rs_v, vimg + (BUF ^ 1) * TILE, 64 * (kt + 1), rowst, wave, voff0);
            if (DROP == 2) bits_next = keep[(int64_t)(kt + 1) * t * 4];
        }
        if (DROP == 1) {                  // keep-bits of query i16, keys 64kt + 16g .. +15; stashed for the backward kernels
            mybits = drop_bits16(dc, (uint64_t)(prow + 64 * kt + 16 * g));
            if (qrow < t) keep[(int64_t)kt * t * 4] = (uint16_t)mybits;
        }
        float x[4][4];
        float tmax = NOKEY;
        score_tiles<KOFF>(64 * (kt + 1) > kfull, fa, qf, lmask, kt, t, masked_raw, lane, x, tmax);
        tmax = xor16_32_max(tmax);
        if (__any(tmax > m)) {           // a new row maximum somewhere in the wave: rescale
            const float m_new = fmaxf(m, tmax);
            const float corr = __builtin_amdgcn_exp2f((m - m_new) * c2);
            l *= corr;
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[d][r] *= corr;
            m = m_new;
        }
        const float nm = -m * c2;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            unsigned bT = 0;
            if (DROP != 0) bT = (unsigned)__shfl((int)mybits, i16 + 16 * T, 64) >> (4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(x[T][r], c2, nm));
                l += pv;
                x[T][r] = DROP != 0 ? and_mask(pv, keep_mask(bT, r)) : pv;
            }
        }
        // ---- O^T += V^T P^T: k-step kp covers the keys of score tiles 2kp, 2kp+1 (in the accumulators' own order)
        const bf16x8 pb0 = pack8(x[0], x[1]), pb1 = pack8(x[2], x[3]);
        // sixteen V fragments, read one ahead: the reads of fragment n+1 are in flight while fragment n feeds its MFMA
        TrFrag vf[2];
        tr_issue<VOFF>(fa, 0, vf[0]);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (n + 1 < 8) tr_issue<VOFF>(fa, n + 1, vf[(n + 1) & 1]);
            else if (n + 1 < 16) tr_issue<VOFF + 8192>(fa, n + 1 - 8, vf[(n + 1) & 1]);
            const bf16x8 v8 = n + 1 < 16 ? tr_wait<2>(vf[n & 1]) : tr_wait<0>(vf[n & 1]);
            oacc[n & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v8, n < 8 ? pb0 : pb1, oacc[n & 7], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, IC<0>{});
        if (kt + 1 < nkt) tile(kt + 1, IC<1>{});
    }
    l = xor16_32_sum(l);
    if (qrow < t) {
        const float inv = dc.scale / l;             // 1/(1-p) of the kept probabilities, applied once
        bf16_t* orow = a.O + (int64_t)b * a.o_batch + (int64_t)qrow * a.o_row + (int64_t)h * a.head;
#pragma unroll
        for (int d = 0; d < 8; ++d) {           // oacc[d][r] = O[qrow][16d + 4g + r]
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(oacc[d][r] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * d + 4 * g) = o;
        }
        if (g == 0) *reinterpret_cast<float2*>(a.stats + (((int64_t)b * a.H + h) * t + qrow) * 2) = make_float2(m * a.alpha, l);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ aux)
// Same shape as the forward (4 waves x 32 queries); recomputes S^T and dPd^T = V dO^T per 64-key tile, dS^T = P (dPd keep / (1-p) -
// delta) (0 at masked keys: masked_fill's backward), dQ^T += K^T dS^T.  Also writes aux = {-m log2 e, 1/l, delta, 0} per query for
// the dK/dV kernel.
template <bool DROP>
__global__ __launch_bounds__(FQ_THREADS, 2) void flash_bwd_dq_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int blk, h, b;
    if (!flash_item(a, blk, h, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = a.t;
    const int q0 = blk * 128 + wave * 32 + i16;
    unsigned char* kimg = smem;
    unsigned char* vimg = smem + 2 * TILE;
    unsigned char* lmask = smem + 4 * TILE;
    int* red = reinterpret_cast<int*>(smem + 4 * TILE + MASK_BYTES);
    const FragAddr fa = frag_addr(lane, (unsigned)(uintptr_t)(lds_void_t*)smem);
    const int64_t hb = (int64_t)b * a.batch + (int64_t)h * a.head;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hb);
    const __amdgpu_buffer_rsrc_t rs_do = make_rsrc(a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.head);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.O + (int64_t)b * a.o_batch + (int64_t)h * a.head);
    const int rowst = (int)a.row;
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc_rows(a.k + hb, t, rowst), rs_v = make_rsrc_rows(a.v + hb, t, rowst);
    const unsigned voff0 = stage_voff(rowst, wave, lane);
    stage_tile<FQ_WAVES>(rs_k, kimg, 0, rowst, wave, voff0);
    stage_tile<FQ_WAVES>(rs_v, vimg, 0, rowst, wave, voff0);

    bf16x8 qf[2][4], dof[2][4];
    float delta[2], nm2[2] = {0.f, 0.f}, linv[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = q0 + 16 * s;
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[s][ks] = ld16(rs_q, qrow < t ? (unsigned)((qrow * rowst + 32 * ks + 8 * g) * 2) : OOB);
            dof[s][ks] = ld16(rs_do, qrow < t ? (unsigned)((qrow * (int)a.do_row + 32 * ks + 8 * g) * 2) : OOB);
            const bf16x8 of = ld16(rs_o, qrow < t ? (unsigned)((qrow * (int)a.o_row + 32 * ks + 8 * g) * 2) : OOB);
#pragma unroll
            for (int c = 0; c < 8; ++c) dl += (float)dof[s][ks][c] * (float)of[c];
        }
        delta[s] = xor16_32_sum(dl);
        if (qrow < t) {
            const float2 st = *reinterpret_cast<const float2*>(a.stats + (((int64_t)b * a.H + h) * t + qrow) * 2);
            nm2[s] = -st.x * LOG2E;
            linv[s] = 1.f / st.y;
        }
    }
    int kfull, kmax;
    mask_setup<FQ_THREADS>(a, b, t, lmask, red, tid, kfull, kmax);
    const int nkt = kmax > 0 ? (kmax + 63) >> 6 : (t + 63) >> 6;
    // per query for the dK/dV kernel: {-m log2 e, 1/l, delta, kmax of this batch row (as bits: saves that kernel the mask scan)}
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = q0 + 16 * s;
        if (qrow < t && g == 0)
            *reinterpret_cast<float4*>(a.aux + (((int64_t)b * a.H + h) * t + qrow) * 4) = make_float4(nm2[s], linv[s], delta[s], __int_as_float(kmax));
    }
    const float scale = DROP ? 65536.f / (65536.f - (float)(uint32_t)(a.pdrop * 65536.f + 0.5f)) : 1.f;
    const int qc0 = q0 < t ? q0 : 0, qc1 = q0 + 16 < t ? q0 + 16 : 0;
    const uint16_t* keep = a.keep + ((((int64_t)b * a.H + h) * a.nkt) * t) * 4 + g;
    const float c2 = a.alpha * LOG2E, masked_raw = MASKED_NAT / a.alpha;
    f32x4 dqacc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int d = 0; d < 8; ++d) dqacc[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned bits_next[2] = {0, 0};
    if (DROP) { bits_next[0] = keep[(int64_t)qc0 * 4]; bits_next[1] = keep[(int64_t)qc1 * 4]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto tile = [&](const int kt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr int KOFF = BUF * TILE, VOFF = 2 * TILE + BUF * TILE;
        const unsigned mybits[2] = {bits_next[0], bits_next[1]};
        if (kt + 1 < nkt) {
            stage_tile<FQ_WAVES>(rs_k, kimg + (BUF ^ 1) * TILE, 64 * (kt + 1), rowst, wave, voff0);
            stage_tile<FQ_WAVES>(rs_v, vimg + (BUF ^ 1) * TILE, 64 * (kt + 1), rowst, wave, voff0);
            if (DROP) {
                bits_next[0] = keep[((int64_t)(kt + 1) * t + qc0) * 4];
                bits_next[1] = keep[((int64_t)(kt + 1) * t + qc1) * 4];
            }
        }
        const bool masked = 64 * (kt + 1) > kfull;
        auto pair = [&](auto KPC) {
            constexpr int kp = decltype(KPC)::value;
            float ds[2][2][4];                               // [sub-tile][u][r]
            auto one = [&](auto UC) {
                constexpr int u = decltype(UC)::value, T = 2 * kp + u;
                __builtin_amdgcn_sched_barrier(0);          // keep the operand reads of later tiles from being hoisted (register pressure)
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa, pa = sa, pbb = sa;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 kfrag = row_frag<KOFF + 4096 * T>(fa, ks);
                    sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[0][ks], sa, 0, 0, 0);
                    sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[1][ks], sb, 0, 0, 0);
                    const bf16x8 vfrag = row_frag<VOFF + 4096 * T>(fa, ks);
                    pa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, dof[0][ks], pa, 0, 0, 0);
                    pbb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, dof[1][ks], pbb, 0, 0, 0);
                }
                const int key0 = 64 * kt + 16 * T + 4 * g;
                float v[2][4] = {{sa[0], sa[1], sa[2], sa[3]}, {sb[0], sb[1], sb[2], sb[3]}};
                const float dp[2][4] = {{pa[0], pa[1], pa[2], pa[3]}, {pbb[0], pbb[1], pbb[2], pbb[3]}};
                unsigned mk = 0x01010101u;
                if (masked) {
                    mk = *reinterpret_cast<const unsigned*>(lmask + key0);
                    mask_fix(v[0], mk, key0, t, masked_raw);
                    mask_fix(v[1], mk, key0, t, masked_raw);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    unsigned bT = 0;
                    if (DROP) bT = (unsigned)__shfl((int)mybits[s], i16 + 16 * T, 64) >> (4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pn = __builtin_amdgcn_exp2f(__builtin_fmaf(v[s][r], c2, nm2[s])) * linv[s];
                        const float t1 = DROP ? and_mask(dp[s][r], keep_mask(bT, r)) : dp[s][r];
                        ds[s][u][r] = pn * __builtin_fmaf(t1, scale, -delta[s]);
                    }
                }
                if (masked) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[s][u][r] = ((mk >> (8 * r)) & 0xFFu) != 0 ? ds[s][u][r] : 0.f;      // masked_fill's backward
                }
            };
            one(IC<0>{}); one(IC<1>{});
            const bf16x8 dsa = pack8(ds[0][0], ds[0][1]), dsb = pack8(ds[1][0], ds[1][1]);
            TrFrag kf_[2];                  // K fragments read one ahead of the MFMA pair they feed
            tr_issue<KOFF + 8192 * kp>(fa, 0, kf_[0]);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                if (d + 1 < 8) tr_issue<KOFF + 8192 * kp>(fa, d + 1, kf_[(d + 1) & 1]);
                const bf16x8 kfrag = d + 1 < 8 ? tr_wait<2>(kf_[d & 1]) : tr_wait<0>(kf_[d & 1]);
                dqacc[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, dsa, dqacc[0][d], 0, 0, 0);
                dqacc[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, dsb, dqacc[1][d], 0, 0, 0);
            }
        };
        pair(IC<0>{}); pair(IC<1>{});
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, IC<0>{});
        if (kt + 1 < nkt) tile(kt + 1, IC<1>{});
    }
    if (a.dbq != nullptr) {               // rows that do not exist hold zeros
        f32x4 both[8];
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) both[d][r] = dqacc[0][d][r] + dqacc[1][d][r];
        block_colsum<FQ_THREADS>(both, a.alpha, reinterpret_cast<float*>(smem), a.dbq + h * 128, tid, lane);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = q0 + 16 * s;
        if (qrow < t) {
            bf16_t* drow = a.dq + (int64_t)b * a.g_batch + (int64_t)qrow * a.g_row + (int64_t)h * a.head;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(dqacc[s][d][r] * a.alpha);
                *reinterpret_cast<bf16x4*>(drow + 16 * d + 4 * g) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Per wave 16 keys (K and V fragments in registers); streams Q, dO and aux tiles of 64 queries.  S = Q K^T and dPd = dO V^T in
// the layout D[query 4g + r][key i16]; dV^T += dO^T Pd, dK^T += Q^T dS (contraction over the queries, operands as above).
// LDS: Q images [2][TILE] at 0, dO images [2][TILE] at 2 TILE, aux [2][AUX_BYTES], two reduction words.
template <int T> __device__ __forceinline__ void row_share4(unsigned bits, unsigned (&kb)[4]) {
    kb[0] = row_share<4 * T>(bits); kb[1] = row_share<4 * T + 1>(bits); kb[2] = row_share<4 * T + 2>(bits); kb[3] = row_share<4 * T + 3>(bits);
}

template <bool DROP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_bwd_dkv_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KB = 16 * NW;          // keys per workgroup: 128 (8 waves, one workgroup per CU) or 64 (4 waves, two per CU)
    int blk, h, b;
    if (!flash_item<KB>(a, blk, h, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = a.t;
    const int kb0 = blk * KB + wave * 16;
    const int key = kb0 + i16;
    unsigned char* qimg = smem;                         // [2][TILE]
    unsigned char* doimg = smem + 2 * TILE;             // [2][TILE]
    unsigned char* auximg = smem + 4 * TILE;            // [2][AUX_BYTES]
    const FragAddr fa = frag_addr(lane, (unsigned)(uintptr_t)(lds_void_t*)smem);
    const int64_t hb = (int64_t)b * a.batch + (int64_t)h * a.head;
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(a.k + hb), rs_v = make_rsrc(a.v + hb);
    const __amdgpu_buffer_rsrc_t rs_aux = make_rsrc(a.aux + (((int64_t)b * a.H + h) * t) * 4);
    const int rowst = (int)a.row, dorow = (int)a.do_row;
    const bool kvalid = key < t;
    const int64_t goff = (int64_t)b * a.g_batch + (int64_t)key * a.g_row + (int64_t)h * a.head;

    const int kmax = __float_as_int(a.aux[(((int64_t)b * a.H + h) * t) * 4 + 3]);      // written by the dQ kernel (row 0 always exists)
    if (kmax > 0 && blk * KB >= kmax) {     // every key of this block is masked: zero probability, zero gradients
        if (kvalid) {
            const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                *reinterpret_cast<bf16x4*>(a.dv + goff + 16 * d + 4 * g) = z;
                *reinterpret_cast<bf16x4*>(a.dk + goff + 16 * d + 4 * g) = z;
            }
        }
        return;
    }

    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc_rows(a.q + hb, t, rowst);
    const __amdgpu_buffer_rsrc_t rs_do = make_rsrc_rows(a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.head, t, dorow);
    const unsigned voff_q = stage_voff(rowst, wave, lane), voff_do = stage_voff(dorow, wave, lane);
    auto stage = [&](int qt, int buf) {
        stage_tile<NW>(rs_q, qimg + buf * TILE, 64 * qt, rowst, wave, voff_q);
        stage_tile<NW>(rs_do, doimg + buf * TILE, 64 * qt, dorow, wave, voff_do);
        if (wave == 0) {
            const int q = 64 * qt + lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_aux, (lds_void_t*)(auximg + buf * AUX_BYTES), 16,
                                                     (int)(q < t ? (unsigned)(q * 16) : OOB), 0, 0, 0);
        }
    };
    const int nqt = (t + 63) >> 6;
    stage(0, 0);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = ld16(rs_k, kvalid ? (unsigned)((key * rowst + 32 * ks + 8 * g) * 2) : OOB);
        vf[ks] = ld16(rs_v, kvalid ? (unsigned)((key * rowst + 32 * ks + 8 * g) * 2) : OOB);
    }
    const bool on = kvalid && a.key_mask[(int64_t)b * t + (kvalid ? key : 0)] != 0;
    // exp2 argument of this lane's key: s * cl + bl - m log2 e  (masked key: -1e4 log2 e; key that does not exist: NOKEY)
    const float cl = on ? a.alpha * LOG2E : 0.f;
    const float bl = !kvalid ? NOKEY : (on ? 0.f : MASKED_NAT * LOG2E);
    const float ml = on ? 1.f : 0.f;                   // masked_fill's backward
    const float scale = DROP ? 65536.f / (65536.f - (float)(uint32_t)(a.pdrop * 65536.f + 0.5f)) : 1.f;
    // keep-bits of the wave's 16 keys for query 64qt + 16(i16>>2) + 4g + (i16&3): element (T, r) = (i16>>2, i16&3) of this
    // lane's row of 16 lanes
    const int qsub = 16 * (i16 >> 2) + 4 * g + (i16 & 3);
    const uint16_t* keep = a.keep + ((((int64_t)b * a.H + h) * a.nkt + (kb0 >> 6)) * t) * 4 + ((kb0 >> 4) & 3);
    unsigned bits_next = (DROP && qsub < t) ? keep[(int64_t)qsub * 4] : 0;
    f32x4 dvacc[8], dkacc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) { dvacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dkacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto tile = [&](const int qt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr int QOFF = BUF * TILE, DOFF = 2 * TILE + BUF * TILE;
        const unsigned mybits = bits_next;
        if (qt + 1 < nqt) {
            stage(qt + 1, BUF ^ 1);
            const int qq = 64 * (qt + 1) + qsub;
            if (DROP) bits_next = qq < t ? keep[(int64_t)qq * 4] : 0;
        }
        const float4* aux = reinterpret_cast<const float4*>(auximg + BUF * AUX_BYTES);
        auto pair = [&](auto KPC) {
            constexpr int kp = decltype(KPC)::value;
            // the transposed fragments of this k-step depend only on the staged tile: their reads are issued BEFORE the score /
            // element-wise phase, so that the 16 MFMAs below do not run at the pace of one LDS round trip each (the compiler
            // otherwise keeps a single fragment of lookahead; the kernel has the registers: 2 waves per SIMD)
            bf16x8 fdo[8], fq[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) fdo[d] = tr_frag<DOFF + 8192 * kp>(fa, d);
#pragma unroll
            for (int d = 0; d < 8; ++d) fq[d] = tr_frag<QOFF + 8192 * kp>(fa, d);
            float pd[2][4], ds[2][4];
            auto one = [&](auto UC) {
                constexpr int u = decltype(UC)::value, T = 2 * kp + u;
                __builtin_amdgcn_sched_barrier(0);      // as in dq_tiles
                const f32x4 s = tile128<QOFF + 4096 * T>(fa, kf);
                const f32x4 dp = tile128<DOFF + 4096 * T>(fa, vf);
                unsigned kb[4] = {0, 0, 0, 0};
                if (DROP) row_share4<T>(mybits, kb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 ax = aux[16 * T + 4 * g + r];                         // {-m log2 e, 1/l, delta, -}; zeros for q >= t
                    const float pn = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], cl, bl) + ax.x) * ax.y;
                    float t1 = dp[r], pk = pn;
                    if (DROP) {
                        const int msk = keep_mask(kb[r], (unsigned)i16);
                        t1 = and_mask(t1, msk);
                        pk = and_mask(pn, msk);
                    }
                    pd[u][r] = pk;
                    ds[u][r] = (pn * ml) * __builtin_fmaf(t1, scale, -ax.z);
                }
            };
            one(IC<0>{}); one(IC<1>{});
            const bf16x8 pdb = pack8(pd[0], pd[1]);
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                dvacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdo[d], pdb, dvacc[d], 0, 0, 0);
                dkacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[d], dsb, dkacc[d], 0, 0, 0);
            }
        };
        pair(IC<0>{}); pair(IC<1>{});
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int qt = 0; qt < nqt; qt += 2) {
        tile(qt, IC<0>{});
        if (qt + 1 < nqt) tile(qt + 1, IC<1>{});
    }
    if (a.dbk != nullptr) block_colsum<64 * NW>(dkacc, a.alpha, reinterpret_cast<float*>(smem), a.dbk + h * 128, tid, lane);
    if (a.dbv != nullptr) block_colsum<64 * NW>(dvacc, scale, reinterpret_cast<float*>(smem) + 128, a.dbv + h * 128, tid, lane);
    if (kvalid) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            bf16x4 ov, ok;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ov[r] = (bf16_t)(dvacc[d][r] * scale); ok[r] = (bf16_t)(dkacc[d][r] * a.alpha); }
            *reinterpret_cast<bf16x4*>(a.dv + goff + 16 * d + 4 * g) = ov;
            *reinterpret_cast<bf16x4*>(a.dk + goff + 16 * d + 4 * g) = ok;
        }
    }
}

// keep-bits of one layer ahead of its forward kernel (the host runs this on a side stream beside the previous layer's GEMMs, whose
// vector ALUs are idle): thread = one 16-bit word [b][h][kt][q][g] = keys 64kt + 16g .. +15 of query q, the same Philox counters
// as flash_fwd_k<1> draws itself
__global__ __launch_bounds__(256) void flash_keep_bits_k(uint16_t* __restrict__ keep, int64_t p_batch, int B, int H, int t, int tp, int nkt,
        float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const int64_t n = (int64_t)B * H * nkt * t * 4;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < n; w += (int64_t)gridDim.x * 256) {
        const int g = (int)(w & 3);
        int64_t r = w >> 2;
        const int q = (int)(r % t); r /= t;
        const int kt = (int)(r % nkt); r /= nkt;
        const int h = (int)(r % H);
        const int b = (int)(r / H);
        keep[w] = (uint16_t)drop_bits16(dc, (uint64_t)(b * p_batch + ((int64_t)h * t + q) * tp + 64 * kt + 16 * g));
    }
}

int check_common(const char* who, const void* q, const void* k, const void* v, int64_t row, int64_t batch, int head, int B, int H, int t,
                 float p, const void* keep_bits) {
    FS2_REQUIRE(q && k && v, "%s: null argument", who);
    FS2_REQUIRE(t > 0 && t <= MASK_BYTES, "%s: need 0 < t <= 1024 (t=%d)", who, t);
    FS2_REQUIRE(B > 0 && H > 0 && (int64_t)B * H * ((t + 127) / 128) < (1 << 28), "%s: bad B/H", who);
    FS2_REQUIRE(row % 8 == 0 && batch % 8 == 0 && head % 8 == 0, "%s: strides must be multiples of 8 elements", who);
    FS2_REQUIRE(fs2_aligned16(q) && fs2_aligned16(k) && fs2_aligned16(v), "%s: pointers must be 16-byte aligned", who);
    FS2_REQUIRE((int64_t)(t + 64) * row * 2 < 0x7FFFFFF0LL, "%s: one (batch, head) slice exceeds 2 GiB", who);
    FS2_REQUIRE(p >= 0.f && p < 1.f && (p == 0.f || keep_bits != nullptr), "%s: dropout needs the keep-bits buffer", who);
    return FS2_OK;
}

int flash_grid(int B, int H, int t) { return 8 * ((B * H + 7) / 8) * ((t + 127) / 128); }

}  // namespace

extern "C" int64_t fs2_flash_attn_keep_words(int B, int H, int t) { return (int64_t)B * H * ((t + 63) / 64) * t * 4; }

extern "C" int fs2_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, int64_t batch_stride, int head_stride,
                                  const uint8_t* key_mask, const int32_t* key_info, void* o_out, int64_t o_row_stride, int64_t o_batch_stride,
                                  float* stats, uint16_t* keep_bits, int pregenerated, int64_t p_batch_stride, int B, int H, int t, int tp, float alpha,
                                  float p, const uint64_t* rng, uint32_t site, void* stream) {
    const int rc = check_common("fs2_flash_attn_fwd", q, k, v, row_stride, batch_stride, head_stride, B, H, t, p, keep_bits);
    if (rc != FS2_OK) return rc;
    FS2_REQUIRE(key_mask && o_out && stats, "fs2_flash_attn_fwd: null argument");
    FS2_REQUIRE(tp == (t + 7) / 8 * 8 && p_batch_stride % 8 == 0, "fs2_flash_attn_fwd: need tp = roundup8(t) and p_batch_stride %% 8 == 0");
    FS2_REQUIRE(o_row_stride % 4 == 0 && o_batch_stride % 4 == 0 && fs2_aligned16(o_out), "fs2_flash_attn_fwd: output rows must be 8-byte aligned");
    FS2_REQUIRE(p == 0.f || pregenerated || rng != nullptr, "fs2_flash_attn_fwd: dropout needs rng");
    FlashArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.row = row_stride; a.batch = batch_stride; a.head = head_stride;
    a.key_mask = key_mask; a.kinfo = key_info; a.O = (bf16_t*)o_out; a.o_row = o_row_stride; a.o_batch = o_batch_stride; a.stats = stats; a.keep = keep_bits;
    a.p_batch = p_batch_stride; a.B = B; a.H = H; a.t = t; a.tp = tp; a.nkt = (t + 63) / 64; a.alpha = alpha; a.pdrop = p; a.rng = rng; a.site = site;
    const int lds = 4 * TILE + MASK_BYTES + 16;
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    if (p > 0.f && pregenerated) hipLaunchKernelGGL(flash_fwd_k<2>, dim3(flash_grid(B, H, t)), dim3(512), lds, (hipStream_t)stream, a);
    else if (p > 0.f) hipLaunchKernelGGL(flash_fwd_k<1>, dim3(flash_grid(B, H, t)), dim3(512), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(flash_fwd_k<0>, dim3(flash_grid(B, H, t)), dim3(512), lds, (hipStream_t)stream, a);
    FS2_CHECK_LAUNCH("fs2_flash_attn_fwd");
    return FS2_OK;
}

extern "C" int fs2_flash_attn_mask_info(const uint8_t* key_mask, int B, int t, int32_t* info, void* stream) {
    FS2_REQUIRE(key_mask && info && B > 0 && t > 0 && t <= MASK_BYTES, "fs2_flash_attn_mask_info: bad arguments");
    hipLaunchKernelGGL(flash_mask_info_k, dim3(B), dim3(512), 0, (hipStream_t)stream, key_mask, t, info);
    FS2_CHECK_LAUNCH("fs2_flash_attn_mask_info");
    hipLaunchKernelGGL(flash_order_k, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, info);
    FS2_CHECK_LAUNCH("fs2_flash_attn_mask_info (order)");
    return FS2_OK;
}

extern "C" int fs2_flash_attn_keep_bits(uint16_t* keep_bits, int64_t p_batch_stride, int B, int H, int t, int tp, float p, const uint64_t* rng,
                                       uint32_t site, void* stream) {
    FS2_REQUIRE(keep_bits && rng && p > 0.f && p < 1.f, "fs2_flash_attn_keep_bits: need a buffer, rng and 0 < p < 1");
    FS2_REQUIRE(B > 0 && H > 0 && t > 0 && t <= MASK_BYTES && tp == (t + 7) / 8 * 8 && p_batch_stride % 8 == 0, "fs2_flash_attn_keep_bits: bad shape");
    const int nkt = (t + 63) / 64;
    const int64_t n = (int64_t)B * H * nkt * t * 4;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(flash_keep_bits_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, keep_bits, p_batch_stride, B, H, t, tp, nkt, p, rng, site);
    FS2_CHECK_LAUNCH("fs2_flash_attn_keep_bits");
    return FS2_OK;
}

extern "C" int fs2_flash_attn_bwd(const void* q, const void* k, const void* v, int64_t row_stride, int64_t batch_stride, int head_stride,
                                  const uint8_t* key_mask, const int32_t* key_info, const void* o_saved, int64_t o_row_stride,
                                  int64_t o_batch_stride,
                                  const void* d_out, int64_t do_row_stride, int64_t do_batch_stride, const float* stats,
                                  const uint16_t* keep_bits, float* aux, void* dq, void* dk, void* dv, int64_t g_row_stride,
                                  int64_t g_batch_stride, float* dbias_q, float* dbias_k, float* dbias_v, int B, int H, int t, float alpha,
                                  float p, void* stream) {
    const int rc = check_common("fs2_flash_attn_bwd", q, k, v, row_stride, batch_stride, head_stride, B, H, t, p, keep_bits);
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
    a.key_mask = key_mask; a.kinfo = key_info; a.O = (bf16_t*)const_cast<void*>(o_saved); a.o_row = o_row_stride; a.o_batch = o_batch_stride;
    a.stats = const_cast<float*>(stats); a.keep = const_cast<uint16_t*>(keep_bits);
    a.B = B; a.H = H; a.t = t; a.tp = (t + 7) / 8 * 8; a.nkt = (t + 63) / 64; a.alpha = alpha; a.pdrop = p;
    a.dO = (const bf16_t*)d_out; a.do_row = do_row_stride; a.do_batch = do_batch_stride; a.aux = aux;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.g_row = g_row_stride; a.g_batch = g_batch_stride;
    a.dbq = dbias_q; a.dbk = dbias_k; a.dbv = dbias_v;
    const int lds_q = 4 * TILE + MASK_BYTES + 16, lds_kv = 4 * TILE + 2 * AUX_BYTES + 16;
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dq_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dq_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    }
    const dim3 grid(flash_grid(B, H, t));
    // dK/dV: 64 keys per workgroup (4 waves, two workgroups per CU; default) or 128 (8 waves, one per CU; FS2_FLASH_DKV_WAVES=8).
    // Every non-empty key block costs the same (all queries), and at config 2 there are ~2.1 of the 128-key blocks per CU: three
    // rounds for two rounds' worth of work.  Half-size blocks leave a shorter tail (a lone workgroup on a CU also runs faster than
    // one of a pair): 149 -> 140 us per decoder layer, at twice the Q / dO staging traffic.
    static const int kv_waves = getenv("FS2_FLASH_DKV_WAVES") ? atoi(getenv("FS2_FLASH_DKV_WAVES")) : 4;
    const dim3 grid_kv(kv_waves == 4 ? 8 * ((B * H + 7) / 8) * ((t + 63) / 64) : flash_grid(B, H, t));
    if (p > 0.f) {
        hipLaunchKernelGGL(flash_bwd_dq_k<true>, grid, dim3(FQ_THREADS), lds_q, (hipStream_t)stream, a);
        if (kv_waves == 4) hipLaunchKernelGGL((flash_bwd_dkv_k<true, 4>), grid_kv, dim3(256), lds_kv, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((flash_bwd_dkv_k<true, 8>), grid_kv, dim3(512), lds_kv, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(flash_bwd_dq_k<false>, grid, dim3(FQ_THREADS), lds_q, (hipStream_t)stream, a);
        if (kv_waves == 4) hipLaunchKernelGGL((flash_bwd_dkv_k<false, 4>), grid_kv, dim3(256), lds_kv, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((flash_bwd_dkv_k<false, 8>), grid_kv, dim3(512), lds_kv, (hipStream_t)stream, a);
    }
    FS2_CHECK_LAUNCH("fs2_flash_attn_bwd");
    return FS2_OK;
}

// Flash-style attention for gfx950 (bf16, d_k = 64 / 96 / 128, any number of keys up to 16384): attention() of the reference
// (Models/modules.py:7-21) and its backward WITHOUT the (tq x tk) probability tensors in HBM -- the mode the trainer runs when the
// attention maps are not requested (hp.return_attn = False).  Self-attention of the FastSpeech2 stacks (tq = tk), the masked
// (no-peak = causal) self-attention and the rectangular encoder-decoder attention of the autoregressive Transformer-TTS decoder
// (Models/layers.py:108-118, masks of train.py:26-58: a key is masked when it is padding OR lies after the query).  Forward keeps per query row the running maximum and the sum of exponentials; backward recomputes
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
constexpr int MASK_MAX = 16384;         // longest key sequence: the key-mask row of a batch element lives in LDS (roundup64(tk) + 64 bytes)
constexpr int AUX_BYTES = 64 * 16;      // dK/dV kernel: {m, 1/l, delta, -} of the 64 queries of a tile

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFF0, 0x00020000);
}
// rows of DK bf16 at a stride of row_stride elements: a 16-byte read that starts past row rows-1 returns zeros
template <int DK>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_rows(const void* p, int rows, int row_stride) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (rows - 1) * row_stride * 2 + 2 * DK, 0x00020000);
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
// 16 x 16 product tile over the DK = 32 KS columns: rows r0.. of the LDS image (OFF = image offset + 256 r0) against the register
// fragments bf (B operand)
template <int OFF, int KS> __device__ __forceinline__ f32x4 tile128(const FragAddr& fa, const bf16x8 (&bf)[KS]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<OFF>(fa, ks), bf[ks], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ bf16x8 pack8(const float (&a)[4], const float (&b)[4]) {
    bf16x8 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) { o[c] = (bf16_t)a[c]; o[4 + c] = (bf16_t)b[c]; }
    return o;
}
// the same with the keep-bits of a dropout applied on the way: element r of `a` survives iff bit r of ba is set (b / bb alike).  Written
// in instructions -- one v_bfe_i32 + one v_and_b32 per element, one v_cvt_pk_bf16_f32 per PAIR: hipcc turns the plain form into
// v_and + v_cmp + v_cndmask + a single-element convert per element and a v_perm per pair (72 vector instructions per 16 x 64 tile
// against 40 here, in a kernel whose vector issue is 3x its matrix time)
template <int R> __device__ __forceinline__ float keep_elem(float v, unsigned bits) {
    int m;
    float o;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(R));
    asm("v_and_b32 %0, %1, %2" : "=v"(o) : "v"(v), "v"(m));
    return o;
}
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned o;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(lo), "v"(hi));
    return o;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 pack8_keep(const float (&a)[4], unsigned ba, const float (&b)[4], unsigned bb) {
    u32x4 o;
    o[0] = cvt_pk(keep_elem<0>(a[0], ba), keep_elem<1>(a[1], ba));
    o[1] = cvt_pk(keep_elem<2>(a[2], ba), keep_elem<3>(a[3], ba));
    o[2] = cvt_pk(keep_elem<0>(b[0], bb), keep_elem<1>(b[1], bb));
    o[3] = cvt_pk(keep_elem<2>(b[2], bb), keep_elem<3>(b[3], bb));
    return __builtin_bit_cast(bf16x8, o);
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
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x150 + N, 0xF, 0xF, true);        // (no `old` operand: hipcc otherwise emits a v_mov 0 per call)
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
template <int NT, int DT>
__device__ __forceinline__ void block_colsum(const f32x4 (&acc)[DT], float factor, float* lds, float* out, int tid, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
    if (tid < 16 * DT) lds[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float tot = row16_sum(acc[d][r]);
            if (i16 == 0) atomicAdd(&lds[16 * d + 4 * g + r], tot * factor);
        }
    __syncthreads();
    if (tid < 16 * DT) atomicAdd(out + tid, lds[tid]);
}

struct FlashArgs {
    const bf16_t *q, *k, *v;            // rows of one head: DK contiguous bf16 at base + b*batch + i*row + h*head
    int64_t row, batch;                 // element strides of the QUERY rows (q, and the dq rows are addressed with g_row / g_batch)
    int64_t kvrow, kvbatch;             // element strides of the KEY / VALUE rows (= row / batch for self-attention on one fused projection)
    int head;
    int tq, causal, mb;                 // queries (keys: t); causal: key j > query i is masked like a padded key; mb = bytes of the LDS mask row
    const uint8_t* key_mask;            // (B, t)
    const int32_t* kinfo;               // optional (B, 3): {kfull, kmax, the batch row of rank b by kmax} (fs2_flash_attn_mask_info), or nullptr
    bf16_t* O;                          // attention output (written by forward, read by backward), rows at O + b*o_batch + i*o_row + h*head
    int64_t o_row, o_batch;
    float* stats;                       // (B, H, tq, 2): {row maximum of the masked scaled scores, sum of exponentials}
    uint16_t* keep;                     // (B, H, nkt, tq, 4) keep-bits of the dropout: bit j of word [b][h][kt][q][g] = key 64kt + 16g + j
    int64_t p_batch;                    // batch stride of the virtual P tensor (dropout counters)
    int B, H, t, tp, nkt;
    float alpha, pdrop;
    const uint64_t* rng;
    uint32_t site;
    // backward
    const bf16_t* dO;                   // rows at dO + b*do_batch + i*do_row + h*head
    int64_t do_row, do_batch;
    float* aux;                         // (B, H, tq, 4) workspace: {-m log2 e, 1/l, delta = rowsum(dO * O), kmax bits}: dQ kernel -> dK/dV kernel
    bf16_t *dq, *dk, *dv;               // rows at dq + b*g_batch + i*g_row + h*head, dk / dv + b*gkv_batch + j*gkv_row + h*head
    int64_t g_row, g_batch, gkv_row, gkv_batch;
    float *dbq, *dbk, *dbv;             // optional bias gradients of the three projections (H*DK floats each): += column sums of dq / dk / dv
};

// Work item of this workgroup.  Workgroups are dealt round-robin to the 8 XCDs; the row blocks of one (batch, head) pair -- which
// stream the same K/V (or Q/dO) tiles -- are given to ONE XCD (one L2), and the pairs go round-robin over the XCDs so that a
// batch sorted by length does not leave one XCD with all the long sequences.  Grid = 8 * ceil(B H / 8) * nblk.
template <int BLK = 128>
__device__ __forceinline__ bool flash_item(const FlashArgs& a, const int nrows, int& blk, int& h, int& b) {
    const int nblk = (nrows + BLK - 1) / BLK;
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
// zero padded to mb bytes).  red: two LDS words.  Ends with a barrier.
template <int NT = 512>
__device__ __forceinline__ void scan_mask(const uint8_t* km_row, int t, int mb, unsigned char* lmask, int* red, int tid) {
    if (tid == 0) { red[0] = t; red[1] = 0; }
    __syncthreads();
    for (int j = tid; j < mb; j += NT) {
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
        for (int j = tid; j < a.mb; j += NT) lmask[j] = j < t ? km_row[j] : 0;
        kfull = a.kinfo[3 * b];
        kmax = a.kinfo[3 * b + 1];
    } else {
        scan_mask<NT>(a.key_mask + (int64_t)b * t, t, a.mb, lmask, red, tid);
        kfull = red[0];
        kmax = red[1];
    }
}
__global__ __launch_bounds__(512) void flash_mask_info_k(const uint8_t* __restrict__ key_mask, int t, int32_t* __restrict__ info) {
    __shared__ int red[2];
    scan_mask<512>(key_mask + (int64_t)blockIdx.x * t, t, (t + 63) / 64 * 64, nullptr, red, threadIdx.x);
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

// The reference's create_masks for the FastSpeech2 task (train_fastspeech2.py:55-82: mask = pos != pad on the (B, t) int64 positions of
// the collate function) and the two kernels above in ONE launch: block b compares and scans row b (all of a row's loads in flight at
// once), the block that takes the last ticket (a persistent word, zero before and after every launch) ranks the rows.
constexpr int INFO_MAXB = 1024;
__global__ __launch_bounds__(256) void pad_mask_info_k(const int64_t* __restrict__ pos, const int64_t ld, const int64_t pad, uint8_t* __restrict__ mask,
                                                       const int B, const int t, int32_t* __restrict__ info, unsigned* __restrict__ ticket) {
    __shared__ int red[2];
    __shared__ int kx[INFO_MAXB];
    __shared__ int last;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { red[0] = t; red[1] = 0; }
    __syncthreads();
    int first = t, lastk = 0;
    for (int j = threadIdx.x; j < t; j += 256) {
        const bool on = pos[(int64_t)b * ld + j] != pad;
        mask[(int64_t)b * t + j] = on ? 1 : 0;
        if (on) lastk = max(lastk, j + 1);
        else first = min(first, j);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        first = min(first, __shfl_xor(first, o));
        lastk = max(lastk, __shfl_xor(lastk, o));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&red[0], first); atomicMax(&red[1], lastk); }
    __syncthreads();
    if (threadIdx.x == 0) {
        info[3 * b] = red[0];
        __hip_atomic_store(info + 3 * b + 1, red[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1u) == (unsigned)B - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    for (int i = threadIdx.x; i < B; i += 256) kx[i] = __hip_atomic_load(info + 3 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {
        const int ki = kx[i];
        int rank = 0;
        for (int j2 = 0; j2 < B; ++j2) {
            const int kj = kx[j2];
            rank += (kj > ki || (kj == ki && j2 < i)) ? 1 : 0;
        }
        info[3 * rank + 2] = i;
    }
    if (threadIdx.x == 0) *ticket = 0u;
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
// Heads narrower than 128 columns keep the 256-byte image rows: the chunks past column DK are never read, their lanes fetch
// nothing (an offset beyond the descriptor: the DMA writes zeros).
template <int DK>
__device__ __forceinline__ unsigned stage_voff(int row_stride, int wave, int lane) {
    const int r = 4 * wave + (lane >> 4);
    const int lc = (lane & 15) ^ img_f(r);
    return 8 * lc < DK ? (unsigned)((r * row_stride + (lc << 3)) * 2) : OOB;
}

__device__ __forceinline__ float and_mask(float v, int msk) { return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & msk); }
__device__ __forceinline__ int keep_mask(unsigned bits, unsigned pos) { return __builtin_amdgcn_sbfe((int)bits, pos, 1u); }   // bit -> 0 / ~0

// dQ kernel shape: workgroup = 4 waves x 32 queries (two 16-query sub-tiles per wave: every K / V fragment read from LDS feeds two
// MFMAs; two independent workgroups per CU).  The forward kernel measured faster as 8 waves x 16 queries (4 waves per SIMD:
// 55.7 vs 60.6 us per launch averaged over the model's eight attention layers).
constexpr int FQ_WAVES = 4, FQ_THREADS = 256;
constexpr int NO_CAUSAL = 0x7FFFFFFF;

// masked (workgroup-uniform): the tile holds masked or non-existent keys -- keys with mask 0, and keys after the query `cq` (causal
// launches; NO_CAUSAL otherwise), get the raw value whose scaled score is -1e4 (masked_fill), keys >= t get NOKEY.  The MFMA work is
// common to both cases; only these fix-ups sit under a branch.
__device__ __forceinline__ void mask_fix(float (&v)[4], unsigned mk, int key0, int t, int cq, float masked_raw) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = (((mk >> (8 * r)) & 0xFFu) != 0 && key0 + r <= cq) ? v[r] : masked_raw;
        v[r] = (key0 + r < t) ? v[r] : NOKEY;
    }
}

// raw scores S^T of one 64-key tile (image at byte offset KOFF) against the wave's 16 queries: x[T][r] = key 64kt + 16T + 4g + r,
// query i16 (cq: this lane's query index in a causal launch).  The MFMA work is common to the masked and the mask-free case, only
// the element-wise fix-ups sit under the branch (two instantiated copies of a whole step cost ~100 spilled registers in the dQ kernel).
template <int KOFF, int KS>
__device__ __forceinline__ void score_tiles(const bool masked, const FragAddr& fa, const bf16x8 (&qf)[KS], const unsigned char* lmask, int kt,
                                            int t, int cq, float masked_raw, int lane, float (&x)[4][4], float& tmax) {
    const int g = lane >> 4;
    auto one = [&](auto TC) {
        constexpr int T = decltype(TC)::value;
        const f32x4 s = tile128<KOFF + 4096 * T, KS>(fa, qf);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[T][r] = s[r];
        if (masked) {
            const int key0 = 64 * kt + 16 * T + 4 * g;
            const unsigned mk = *reinterpret_cast<const unsigned*>(lmask + key0);        // key0 % 4 == 0; bytes >= t are 0
            mask_fix(x[T], mk, key0, t, cq, masked_raw);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, x[T][r]);
    };
    one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{});
}

// number of key tiles a block of queries [q_lo, q_hi] has to visit: tiles at or beyond kmax hold masked keys only (probability
// exp(-1e4 - m) = 0 in fp32 once the row has an unmasked key: kmax > 0); in a causal launch so do the tiles after the block's last
// query -- provided key 0 is unmasked (kfull > 0), which gives every query a visible key
__device__ __forceinline__ int key_tiles(const FlashArgs& a, int kfull, int kmax, int q_hi) {
    int nkt = kmax > 0 ? (kmax + 63) >> 6 : (a.t + 63) >> 6;
    if (a.causal && kfull > 0) nkt = min(nkt, (q_hi >> 6) + 1);
    return nkt;
}

// ------------------------------------------------------------------------------------------------ forward
// O = dropout(softmax(mask(alpha Q K^T))) V, stats = {m, l}.  Wave: 16 queries (columns of the transposed score tiles).
// LDS: K images [2][TILE] at 0, V images [2][TILE] at 2 TILE, key mask, two reduction words.
// DROP: 0 no dropout; 1 draw the keep-bits (Philox) and stash them; 2 read bits that fs2_flash_attn_keep_bits wrote beforehand
template <int DROP, int DK>
__global__ __launch_bounds__(512, 4) void flash_fwd_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = DK / 32, DT = DK / 16;
    int blk, h, b;
    if (!flash_item(a, a.tq, blk, h, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = a.t, tq = a.tq;
    const int qrow = blk * 128 + wave * 16 + i16;
    unsigned char* kimg = smem;                       // [2][TILE]
    unsigned char* vimg = smem + 2 * TILE;            // [2][TILE]
    unsigned char* lmask = smem + 4 * TILE;
    int* red = reinterpret_cast<int*>(smem + 4 * TILE + a.mb);
    const FragAddr fa = frag_addr(lane, (unsigned)(uintptr_t)(lds_void_t*)smem);
    const int64_t hbq = (int64_t)b * a.batch + (int64_t)h * a.head, hbk = (int64_t)b * a.kvbatch + (int64_t)h * a.head;
    const int rowst = (int)a.row, kvrow = (int)a.kvrow;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hbq), rs_k = make_rsrc_rows<DK>(a.k + hbk, t, kvrow), rs_v = make_rsrc_rows<DK>(a.v + hbk, t, kvrow);
    const unsigned voff0 = stage_voff<DK>(kvrow, wave, lane);
    stage_tile(rs_k, kimg, 0, kvrow, wave, voff0);
    stage_tile(rs_v, vimg, 0, kvrow, wave, voff0);
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = ld16(rs_q, qrow < tq ? (unsigned)((qrow * rowst + 32 * ks + 8 * g) * 2) : OOB);
    int kfull, kmax;
    mask_setup<512>(a, b, t, lmask, red, tid, kfull, kmax);
    const int nkt = key_tiles(a, kfull, kmax, min(blk * 128 + 127, tq - 1));
    const int cq = a.causal ? qrow : NO_CAUSAL;

    const DropCtx dc = drop_ctx(a.rng, a.site, a.pdrop);
    const int64_t prow = (int64_t)b * a.p_batch + ((int64_t)h * tq + (qrow < tq ? qrow : 0)) * a.tp;
    uint16_t* keep = a.keep + ((((int64_t)b * a.H + h) * a.nkt) * tq + (qrow < tq ? qrow : 0)) * 4 + g;
    const float c2 = a.alpha * LOG2E, masked_raw = MASKED_NAT / a.alpha;
    float m = NOKEY, l = 0.f;            // m: running maximum of the RAW scores
    f32x4 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned bits_next = DROP == 2 ? keep[0] : 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto tile = [&](const int kt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr int KOFF = BUF * TILE, VOFF = 2 * TILE + BUF * TILE;
        unsigned mybits = bits_next;
        if (kt + 1 < nkt) {
            stage_tile(rs_k, kimg + (BUF ^ 1) * TILE, 64 * (kt + 1), kvrow, wave, voff0);
            stage_tile(rs_v, vimg + (BUF ^ 1) * TILE, 64 * (kt + 1), kvrow, wave, voff0);
            if (DROP == 2) bits_next = keep[(int64_t)(kt + 1) * tq * 4];
        }
        if (DROP == 1) {                  // keep-bits of query i16, keys 64kt + 16g .. +15; stashed for the backward kernels
            mybits = drop_bits16(dc, (uint64_t)(prow + 64 * kt + 16 * g));
            if (qrow < tq) keep[(int64_t)kt * tq * 4] = (uint16_t)mybits;
        }
        float x[4][4];
        float tmax = NOKEY;
        score_tiles<KOFF, KS>(64 * (kt + 1) > kfull || (a.causal && 64 * (kt + 1) - 1 > blk * 128), fa, qf, lmask, kt, t, cq, masked_raw, lane, x, tmax);
        tmax = xor16_32_max(tmax);
        if (__any(tmax > m)) {           // a new row maximum somewhere in the wave: rescale
            const float m_new = fmaxf(m, tmax);
            const float corr = __builtin_amdgcn_exp2f((m - m_new) * c2);
            l *= corr;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[d][r] *= corr;
            m = m_new;
        }
        const float nm = -m * c2;
        unsigned bT[4] = {0, 0, 0, 0};
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            if (DROP != 0) bT[T] = (unsigned)__shfl((int)mybits, i16 + 16 * T, 64) >> (4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(x[T][r], c2, nm));
                l += pv;
                x[T][r] = pv;
            }
        }
        // ---- O^T += V^T P^T: k-step kp covers the keys of score tiles 2kp, 2kp+1 (in the accumulators' own order)
        const bf16x8 pb0 = DROP != 0 ? pack8_keep(x[0], bT[0], x[1], bT[1]) : pack8(x[0], x[1]);
        const bf16x8 pb1 = DROP != 0 ? pack8_keep(x[2], bT[2], x[3], bT[3]) : pack8(x[2], x[3]);
        // 2 DT V fragments, read one ahead: the reads of fragment n+1 are in flight while fragment n feeds its MFMA
        TrFrag vf[2];
        tr_issue<VOFF>(fa, 0, vf[0]);
#pragma unroll
        for (int n = 0; n < 2 * DT; ++n) {
            if (n + 1 < DT) tr_issue<VOFF>(fa, n + 1, vf[(n + 1) & 1]);
            else if (n + 1 < 2 * DT) tr_issue<VOFF + 8192>(fa, n + 1 - DT, vf[(n + 1) & 1]);
            const bf16x8 v8 = n + 1 < 2 * DT ? tr_wait<2>(vf[n & 1]) : tr_wait<0>(vf[n & 1]);
            oacc[n % DT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v8, n < DT ? pb0 : pb1, oacc[n % DT], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, IC<0>{});
        if (kt + 1 < nkt) tile(kt + 1, IC<1>{});
    }
    l = xor16_32_sum(l);
    if (qrow < tq) {
        const float inv = dc.scale / l;             // 1/(1-p) of the kept probabilities, applied once
        bf16_t* orow = a.O + (int64_t)b * a.o_batch + (int64_t)qrow * a.o_row + (int64_t)h * a.head;
#pragma unroll
        for (int d = 0; d < DT; ++d) {          // oacc[d][r] = O[qrow][16d + 4g + r]
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(oacc[d][r] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * d + 4 * g) = o;
        }
        if (g == 0) *reinterpret_cast<float2*>(a.stats + (((int64_t)b * a.H + h) * tq + qrow) * 2) = make_float2(m * a.alpha, l);
    }
}

// ------------------------------------------------------------------------------------------------ attention maps after the fact
// The (tq x tkp) maps the reference returns (dropout(softmax(mask(alpha Q K^T))), Models/modules.py:19-21) WITHOUT routing the arithmetic of
// the layer through them: the forward / backward kernels above run as in the map-free mode and this kernel writes the map of one layer
// from Q, K, the row statistics {m, l} and the stashed keep-bits of the forward.  Half of a forward's matrix work (no P V product), one
// pass of stores.  P[b][h][q][key] at pm + b*pm_batch + (h*tq + q)*tp + key, bf16; keys in [tk, tkp) and every tile the forward skipped
// (beyond the last unmasked key, or after the block's queries in a causal launch: probability exactly 0) are written as zeros.
// LDS: K images [2][TILE] at 0, key mask, two reduction words.
template <bool DROP, int DK>
__global__ __launch_bounds__(512, 4) void flash_probs_k(const FlashArgs a, bf16_t* __restrict__ pm, const int64_t pm_batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = DK / 32;
    int blk, h, b;
    if (!flash_item(a, a.tq, blk, h, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = a.t, tq = a.tq, tp = a.tp;
    const int qrow = blk * 128 + wave * 16 + i16;
    unsigned char* kimg = smem;                       // [2][TILE]
    unsigned char* lmask = smem + 2 * TILE;
    int* red = reinterpret_cast<int*>(smem + 2 * TILE + a.mb);
    const FragAddr fa = frag_addr(lane, (unsigned)(uintptr_t)(lds_void_t*)smem);
    const int64_t hbq = (int64_t)b * a.batch + (int64_t)h * a.head, hbk = (int64_t)b * a.kvbatch + (int64_t)h * a.head;
    const int rowst = (int)a.row, kvrow = (int)a.kvrow;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hbq), rs_k = make_rsrc_rows<DK>(a.k + hbk, t, kvrow);
    const unsigned voff0 = stage_voff<DK>(kvrow, wave, lane);
    stage_tile(rs_k, kimg, 0, kvrow, wave, voff0);
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = ld16(rs_q, qrow < tq ? (unsigned)((qrow * rowst + 32 * ks + 8 * g) * 2) : OOB);
    int kfull, kmax;
    mask_setup<512>(a, b, t, lmask, red, tid, kfull, kmax);
    const int nkt = key_tiles(a, kfull, kmax, min(blk * 128 + 127, tq - 1));      // the tiles the forward visited
    const int nkt_all = (tp + 63) >> 6;                                            // the tiles of the stored row
    const int cq = a.causal ? qrow : NO_CAUSAL;
    const bool qok = qrow < tq;
    const float2 st = qok ? *reinterpret_cast<const float2*>(a.stats + (((int64_t)b * a.H + h) * tq + qrow) * 2) : make_float2(0.f, 1.f);
    const float c2 = a.alpha * LOG2E, masked_raw = MASKED_NAT / a.alpha;
    const float nm = -st.x * LOG2E;                  // stats hold m * alpha
    const float scale = DROP ? 65536.f / (65536.f - (float)(uint32_t)(a.pdrop * 65536.f + 0.5f)) : 1.f;
    const float inv = scale / st.y;
    bf16_t* prow = pm + (int64_t)b * pm_batch + ((int64_t)h * tq + (qok ? qrow : 0)) * tp;
    const uint16_t* keep = a.keep + ((((int64_t)b * a.H + h) * a.nkt) * tq + (qok ? qrow : 0)) * 4;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto tile = [&](const int kt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr int KOFF = BUF * TILE;
        if (kt + 1 < nkt) stage_tile(rs_k, kimg + (BUF ^ 1) * TILE, 64 * (kt + 1), kvrow, wave, voff0);
        uint2 kw = make_uint2(0u, 0u);
        if (DROP && qok) kw = *reinterpret_cast<const uint2*>(keep + (int64_t)kt * tq * 4);      // the four 16-key words of this query and tile
        float x[4][4];
        float tmax = NOKEY;
        score_tiles<KOFF, KS>(64 * (kt + 1) > kfull || (a.causal && 64 * (kt + 1) - 1 > blk * 128), fa, qf, lmask, kt, t, cq, masked_raw, lane, x, tmax);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const unsigned w16 = ((T & 2) ? kw.y : kw.x) >> ((T & 1) * 16);
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(x[T][r], c2, nm)) * inv;
                if (DROP) pv = and_mask(pv, keep_mask(w16, (unsigned)(4 * g + r)));
                o[r] = (bf16_t)pv;
            }
            const int key0 = 64 * kt + 16 * T + 4 * g;
            if (qok && key0 < tp) *reinterpret_cast<bf16x4*>(prow + key0) = o;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, IC<0>{});
        if (kt + 1 < nkt) tile(kt + 1, IC<1>{});
    }
    // tiles the forward skipped: zeros
    if (qok) {
        const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        for (int kt = nkt; kt < nkt_all; ++kt)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const int key0 = 64 * kt + 16 * T + 4 * g;
                if (key0 < tp) *reinterpret_cast<bf16x4*>(prow + key0) = z;
            }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ aux)
// Same shape as the forward (4 waves x 32 queries); recomputes S^T and dPd^T = V dO^T per 64-key tile, dS^T = P (dPd keep / (1-p) -
// delta) (0 at masked keys: masked_fill's backward), dQ^T += K^T dS^T.  Also writes aux = {-m log2 e, 1/l, delta, 0} per query for
// the dK/dV kernel.
template <bool DROP, int DK>
__global__ __launch_bounds__(FQ_THREADS, 2) void flash_bwd_dq_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = DK / 32, DT = DK / 16;
    int blk, h, b;
    if (!flash_item(a, a.tq, blk, h, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = a.t, tq = a.tq;
    const int q0 = blk * 128 + wave * 32 + i16;
    unsigned char* kimg = smem;
    unsigned char* vimg = smem + 2 * TILE;
    unsigned char* lmask = smem + 4 * TILE;
    int* red = reinterpret_cast<int*>(smem + 4 * TILE + a.mb);
    const FragAddr fa = frag_addr(lane, (unsigned)(uintptr_t)(lds_void_t*)smem);
    const int64_t hbq = (int64_t)b * a.batch + (int64_t)h * a.head, hbk = (int64_t)b * a.kvbatch + (int64_t)h * a.head;
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(a.q + hbq);
    const __amdgpu_buffer_rsrc_t rs_do = make_rsrc(a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.head);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.O + (int64_t)b * a.o_batch + (int64_t)h * a.head);
    const int rowst = (int)a.row, kvrow = (int)a.kvrow;
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc_rows<DK>(a.k + hbk, t, kvrow), rs_v = make_rsrc_rows<DK>(a.v + hbk, t, kvrow);
    const unsigned voff0 = stage_voff<DK>(kvrow, wave, lane);
    stage_tile<FQ_WAVES>(rs_k, kimg, 0, kvrow, wave, voff0);
    stage_tile<FQ_WAVES>(rs_v, vimg, 0, kvrow, wave, voff0);

    bf16x8 qf[2][KS], dof[2][KS];
    float delta[2], nm2[2] = {0.f, 0.f}, linv[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = q0 + 16 * s;
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[s][ks] = ld16(rs_q, qrow < tq ? (unsigned)((qrow * rowst + 32 * ks + 8 * g) * 2) : OOB);
            dof[s][ks] = ld16(rs_do, qrow < tq ? (unsigned)((qrow * (int)a.do_row + 32 * ks + 8 * g) * 2) : OOB);
            const bf16x8 of = ld16(rs_o, qrow < tq ? (unsigned)((qrow * (int)a.o_row + 32 * ks + 8 * g) * 2) : OOB);
#pragma unroll
            for (int c = 0; c < 8; ++c) dl += (float)dof[s][ks][c] * (float)of[c];
        }
        delta[s] = xor16_32_sum(dl);
        if (qrow < tq) {
            const float2 st = *reinterpret_cast<const float2*>(a.stats + (((int64_t)b * a.H + h) * tq + qrow) * 2);
            nm2[s] = -st.x * LOG2E;
            linv[s] = 1.f / st.y;
        }
    }
    int kfull, kmax;
    mask_setup<FQ_THREADS>(a, b, t, lmask, red, tid, kfull, kmax);
    const int nkt = key_tiles(a, kfull, kmax, min(blk * 128 + 127, tq - 1));
    // per query for the dK/dV kernel: {-m log2 e, 1/l, delta, kmax of this batch row and "key 0 is unmasked" (as bits: saves that
    // kernel the mask scan)}
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = q0 + 16 * s;
        if (qrow < tq && g == 0)
            *reinterpret_cast<float4*>(a.aux + (((int64_t)b * a.H + h) * tq + qrow) * 4) =
                make_float4(nm2[s], linv[s], delta[s], __int_as_float(kmax | (kfull > 0 ? 0x40000000 : 0)));
    }
    const float scale = DROP ? 65536.f / (65536.f - (float)(uint32_t)(a.pdrop * 65536.f + 0.5f)) : 1.f;
    const int qc0 = q0 < tq ? q0 : 0, qc1 = q0 + 16 < tq ? q0 + 16 : 0;
    const int cq[2] = {a.causal ? q0 : NO_CAUSAL, a.causal ? q0 + 16 : NO_CAUSAL};
    const uint16_t* keep = a.keep + ((((int64_t)b * a.H + h) * a.nkt) * tq) * 4 + g;
    const float c2 = a.alpha * LOG2E, masked_raw = MASKED_NAT / a.alpha;
    f32x4 dqacc[2][DT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int d = 0; d < DT; ++d) dqacc[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned bits_next[2] = {0, 0};
    if (DROP) { bits_next[0] = keep[(int64_t)qc0 * 4]; bits_next[1] = keep[(int64_t)qc1 * 4]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto tile = [&](const int kt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr int KOFF = BUF * TILE, VOFF = 2 * TILE + BUF * TILE;
        const unsigned mybits[2] = {bits_next[0], bits_next[1]};
        if (kt + 1 < nkt) {
            stage_tile<FQ_WAVES>(rs_k, kimg + (BUF ^ 1) * TILE, 64 * (kt + 1), kvrow, wave, voff0);
            stage_tile<FQ_WAVES>(rs_v, vimg + (BUF ^ 1) * TILE, 64 * (kt + 1), kvrow, wave, voff0);
            if (DROP) {
                bits_next[0] = keep[((int64_t)(kt + 1) * tq + qc0) * 4];
                bits_next[1] = keep[((int64_t)(kt + 1) * tq + qc1) * 4];
            }
        }
        const bool masked = 64 * (kt + 1) > kfull || (a.causal && 64 * (kt + 1) - 1 > blk * 128);
        auto pair = [&](auto KPC) {
            constexpr int kp = decltype(KPC)::value;
            float ds[2][2][4];                               // [sub-tile][u][r]
            auto one = [&](auto UC) {
                constexpr int u = decltype(UC)::value, T = 2 * kp + u;
                __builtin_amdgcn_sched_barrier(0);          // keep the operand reads of later tiles from being hoisted (register pressure)
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa, pa = sa, pbb = sa;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
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
                    mask_fix(v[0], mk, key0, t, cq[0], masked_raw);
                    mask_fix(v[1], mk, key0, t, cq[1], masked_raw);
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
                        for (int r = 0; r < 4; ++r)
                            ds[s][u][r] = (((mk >> (8 * r)) & 0xFFu) != 0 && key0 + r <= cq[s]) ? ds[s][u][r] : 0.f;      // masked_fill's backward
                }
            };
            one(IC<0>{}); one(IC<1>{});
            const bf16x8 dsa = pack8(ds[0][0], ds[0][1]), dsb = pack8(ds[1][0], ds[1][1]);
            TrFrag kf_[2];                  // K fragments read one ahead of the MFMA pair they feed
            tr_issue<KOFF + 8192 * kp>(fa, 0, kf_[0]);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                if (d + 1 < DT) tr_issue<KOFF + 8192 * kp>(fa, d + 1, kf_[(d + 1) & 1]);
                const bf16x8 kfrag = d + 1 < DT ? tr_wait<2>(kf_[d & 1]) : tr_wait<0>(kf_[d & 1]);
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
        f32x4 both[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) both[d][r] = dqacc[0][d][r] + dqacc[1][d][r];
        block_colsum<FQ_THREADS, DT>(both, a.alpha, reinterpret_cast<float*>(smem), a.dbq + h * DK, tid, lane);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = q0 + 16 * s;
        if (qrow < tq) {
            bf16_t* drow = a.dq + (int64_t)b * a.g_batch + (int64_t)qrow * a.g_row + (int64_t)h * a.head;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
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

// CAUSAL: the launch masks key j > query i (a.causal); a template parameter because the visibility test costs four vector instructions per
// probability (a quarter of the kernel's element-wise work) that the self-attention of FastSpeech2 does not need
template <bool DROP, int NW, int DK, bool CAUSAL>
__global__ __launch_bounds__(64 * NW, 2) void flash_bwd_dkv_k(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = DK / 32, DT = DK / 16;
    constexpr int KB = 16 * NW;          // keys per workgroup: 64 (4 waves, two workgroups per CU)
    int blk, h, b;
    if (!flash_item<KB>(a, a.t, blk, h, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = a.t, tq = a.tq;
    const int kb0 = blk * KB + wave * 16;
    const int key = kb0 + i16;
    unsigned char* qimg = smem;                         // [2][TILE]
    unsigned char* doimg = smem + 2 * TILE;             // [2][TILE]
    unsigned char* auximg = smem + 4 * TILE;            // [2][AUX_BYTES]
    const FragAddr fa = frag_addr(lane, (unsigned)(uintptr_t)(lds_void_t*)smem);
    const int64_t hbq = (int64_t)b * a.batch + (int64_t)h * a.head, hbk = (int64_t)b * a.kvbatch + (int64_t)h * a.head;
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(a.k + hbk), rs_v = make_rsrc(a.v + hbk);
    const __amdgpu_buffer_rsrc_t rs_aux = make_rsrc(a.aux + (((int64_t)b * a.H + h) * tq) * 4);
    const int rowst = (int)a.row, kvrow = (int)a.kvrow, dorow = (int)a.do_row;
    const bool kvalid = key < t;
    const int64_t goff = (int64_t)b * a.gkv_batch + (int64_t)key * a.gkv_row + (int64_t)h * a.head;

    const int kbits = __float_as_int(a.aux[(((int64_t)b * a.H + h) * tq) * 4 + 3]);      // written by the dQ kernel (query 0 always exists)
    const int kmax = kbits & 0x3FFFFFFF;
    const bool key0_visible = (kbits & 0x40000000) != 0;
    if (kmax > 0 && blk * KB >= kmax) {     // every key of this block is masked: zero probability, zero gradients
        if (kvalid) {
            const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                *reinterpret_cast<bf16x4*>(a.dv + goff + 16 * d + 4 * g) = z;
                *reinterpret_cast<bf16x4*>(a.dk + goff + 16 * d + 4 * g) = z;
            }
        }
        return;
    }

    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc_rows<DK>(a.q + hbq, tq, rowst);
    const __amdgpu_buffer_rsrc_t rs_do = make_rsrc_rows<DK>(a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.head, tq, dorow);
    const unsigned voff_q = stage_voff<DK>(rowst, wave, lane), voff_do = stage_voff<DK>(dorow, wave, lane);
    auto stage = [&](int qt, int buf) {
        stage_tile<NW>(rs_q, qimg + buf * TILE, 64 * qt, rowst, wave, voff_q);
        stage_tile<NW>(rs_do, doimg + buf * TILE, 64 * qt, dorow, wave, voff_do);
        if (wave == 0) {
            const int q = 64 * qt + lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_aux, (lds_void_t*)(auximg + buf * AUX_BYTES), 16,
                                                     (int)(q < tq ? (unsigned)(q * 16) : OOB), 0, 0, 0);
        }
    };
    const int nqt = (tq + 63) >> 6;
    // causal: the queries before this block's first key do not see it (their probabilities are exactly 0 once key 0 is visible to them)
    const int qt0 = (CAUSAL && key0_visible) ? min((blk * KB) >> 6, nqt - 1) : 0;
    stage(qt0, 0);
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = ld16(rs_k, kvalid ? (unsigned)((key * kvrow + 32 * ks + 8 * g) * 2) : OOB);
        vf[ks] = ld16(rs_v, kvalid ? (unsigned)((key * kvrow + 32 * ks + 8 * g) * 2) : OOB);
    }
    const bool on = kvalid && a.key_mask[(int64_t)b * t + (kvalid ? key : 0)] != 0;
    // exp2 argument of this lane's key: s * cl + bl - m log2 e  (masked key: -1e4 log2 e; key that does not exist: NOKEY)
    const float cl = on ? a.alpha * LOG2E : 0.f;
    const float bl = !kvalid ? NOKEY : (on ? 0.f : MASKED_NAT * LOG2E);
    const float ml = on ? 1.f : 0.f;                   // masked_fill's backward
    const float bl_causal = kvalid ? MASKED_NAT * LOG2E : NOKEY;       // ... of a key that lies after the query (causal launches)
    const int ckey = CAUSAL ? key : -1;                // query >= ckey sees this key
    const float scale = DROP ? 65536.f / (65536.f - (float)(uint32_t)(a.pdrop * 65536.f + 0.5f)) : 1.f;
    // keep-bits of the wave's 16 keys for query 64qt + 16(i16>>2) + 4g + (i16&3): element (T, r) = (i16>>2, i16&3) of this
    // lane's row of 16 lanes
    const int qsub = 16 * (i16 >> 2) + 4 * g + (i16 & 3);
    const uint16_t* keep = a.keep + ((((int64_t)b * a.H + h) * a.nkt + (kb0 >> 6)) * tq) * 4 + ((kb0 >> 4) & 3);
    unsigned bits_next = (DROP && 64 * qt0 + qsub < tq) ? keep[(int64_t)(64 * qt0 + qsub) * 4] : 0;
    f32x4 dvacc[DT], dkacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { dvacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dkacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto tile = [&](const int qt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr int QOFF = BUF * TILE, DOFF = 2 * TILE + BUF * TILE;
        const unsigned mybits = bits_next;
        if (qt + 1 < nqt) {
            stage(qt + 1, BUF ^ 1);
            const int qq = 64 * (qt + 1) + qsub;
            if (DROP) bits_next = qq < tq ? keep[(int64_t)qq * 4] : 0;
        }
        const float4* aux = reinterpret_cast<const float4*>(auximg + BUF * AUX_BYTES);
        auto pair = [&](auto KPC) {
            constexpr int kp = decltype(KPC)::value;
            // the transposed fragments of this k-step depend only on the staged tile: their reads are issued BEFORE the score /
            // element-wise phase, so that the 2 DT MFMAs below do not run at the pace of one LDS round trip each (the compiler
            // otherwise keeps a single fragment of lookahead; the kernel has the registers: 2 waves per SIMD)
            bf16x8 fdo[DT], fq[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) fdo[d] = tr_frag<DOFF + 8192 * kp>(fa, d);
#pragma unroll
            for (int d = 0; d < DT; ++d) fq[d] = tr_frag<QOFF + 8192 * kp>(fa, d);
            float pd[2][4], ds[2][4];
            auto one = [&](auto UC) {
                constexpr int u = decltype(UC)::value, T = 2 * kp + u;
                __builtin_amdgcn_sched_barrier(0);      // as in dq_tiles
                const f32x4 s = tile128<QOFF + 4096 * T, KS>(fa, kf);
                const f32x4 dp = tile128<DOFF + 4096 * T, KS>(fa, vf);
                unsigned kb[4] = {0, 0, 0, 0};
                if (DROP) row_share4<T>(mybits, kb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 ax = aux[16 * T + 4 * g + r];                         // {-m log2 e, 1/l, delta, -}; zeros for q >= tq
                    const bool vis = !CAUSAL || 64 * qt + 16 * T + 4 * g + r >= ckey;
                    const float pn = __builtin_amdgcn_exp2f((vis ? __builtin_fmaf(s[r], cl, bl) : bl_causal) + ax.x) * ax.y;
                    float t1 = dp[r], pk = pn;
                    if (DROP) {
                        const int msk = keep_mask(kb[r], (unsigned)i16);
                        t1 = and_mask(t1, msk);
                        pk = and_mask(pn, msk);
                    }
                    pd[u][r] = pk;
                    ds[u][r] = (pn * (vis ? ml : 0.f)) * __builtin_fmaf(t1, scale, -ax.z);
                }
            };
            one(IC<0>{}); one(IC<1>{});
            const bf16x8 pdb = pack8(pd[0], pd[1]);
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                dvacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdo[d], pdb, dvacc[d], 0, 0, 0);
                dkacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[d], dsb, dkacc[d], 0, 0, 0);
            }
        };
        pair(IC<0>{}); pair(IC<1>{});
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int qt = qt0; qt < nqt; qt += 2) {
        tile(qt, IC<0>{});
        if (qt + 1 < nqt) tile(qt + 1, IC<1>{});
    }
    if (a.dbk != nullptr) block_colsum<64 * NW, DT>(dkacc, a.alpha, reinterpret_cast<float*>(smem), a.dbk + h * DK, tid, lane);
    if (a.dbv != nullptr) block_colsum<64 * NW, DT>(dvacc, scale, reinterpret_cast<float*>(smem) + 128, a.dbv + h * DK, tid, lane);
    if (kvalid) {
#pragma unroll
        for (int d = 0; d < DT; ++d) {
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

int check_desc(const char* who, const FS2FlashAttn& d, bool backward) {
    FS2_REQUIRE(d.q && d.k && d.v && d.key_mask && d.o && d.stats, "%s: null argument", who);
    FS2_REQUIRE(d.dk == 64 || d.dk == 96 || d.dk == 128, "%s: d_k must be 64, 96 or 128 (d_k=%d)", who, d.dk);
    FS2_REQUIRE(d.tq > 0 && d.tk > 0 && d.tk <= MASK_MAX, "%s: need tq > 0 and 0 < tk <= %d (tq=%d tk=%d)", who, MASK_MAX, d.tq, d.tk);
    FS2_REQUIRE(d.B > 0 && d.H > 0 && (int64_t)d.B * d.H * ((d.tq + 127) / 128) < (1 << 28) && (int64_t)d.B * d.H * ((d.tk + 63) / 64) < (1 << 28),
                "%s: bad B/H", who);
    FS2_REQUIRE(d.q_row_stride % 8 == 0 && d.q_batch_stride % 8 == 0 && d.kv_row_stride % 8 == 0 && d.kv_batch_stride % 8 == 0 && d.head_stride % 8 == 0,
                "%s: strides must be multiples of 8 elements", who);
    FS2_REQUIRE(fs2_aligned16(d.q) && fs2_aligned16(d.k) && fs2_aligned16(d.v), "%s: pointers must be 16-byte aligned", who);
    FS2_REQUIRE((int64_t)(d.tq + 64) * d.q_row_stride * 2 < 0x7FFFFFF0LL && (int64_t)(d.tk + 64) * d.kv_row_stride * 2 < 0x7FFFFFF0LL,
                "%s: one (batch, head) slice exceeds 2 GiB", who);
    FS2_REQUIRE(d.p >= 0.f && d.p < 1.f && (d.p == 0.f || d.keep_bits != nullptr), "%s: dropout needs the keep-bits buffer", who);
    FS2_REQUIRE(d.tkp == (d.tk + 7) / 8 * 8 && d.p_batch_stride % 8 == 0, "%s: need tkp = roundup8(tk) and p_batch_stride %% 8 == 0", who);
    FS2_REQUIRE(d.o_row_stride % 8 == 0 && d.o_batch_stride % 8 == 0 && fs2_aligned16(d.o), "%s: output rows must be 16-byte aligned", who);
    if (!backward) FS2_REQUIRE(d.p == 0.f || d.pregenerated || d.rng != nullptr, "%s: dropout needs rng", who);
    if (backward) {
        FS2_REQUIRE(d.d_out && d.aux && d.dq && d.dk_out && d.dv_out, "%s: null gradient argument", who);
        FS2_REQUIRE(d.do_row_stride % 8 == 0 && d.do_batch_stride % 8 == 0 && d.dq_row_stride % 4 == 0 && d.dq_batch_stride % 4 == 0 &&
                        d.dkv_row_stride % 4 == 0 && d.dkv_batch_stride % 4 == 0, "%s: strides must be multiples of 8 elements (gradients: 4)", who);
        FS2_REQUIRE(fs2_aligned16(d.d_out) && fs2_aligned16(d.aux) && fs2_aligned16(d.dq) && fs2_aligned16(d.dk_out) && fs2_aligned16(d.dv_out),
                    "%s: pointers must be 16-byte aligned", who);
        FS2_REQUIRE((int64_t)(d.tq + 64) * d.do_row_stride * 2 < 0x7FFFFFF0LL && (int64_t)(d.tq + 64) * d.o_row_stride * 2 < 0x7FFFFFF0LL,
                    "%s: one (batch, head) slice exceeds 2 GiB", who);
    }
    return FS2_OK;
}

FlashArgs to_args(const FS2FlashAttn& d) {
    FlashArgs a = {};
    a.q = (const bf16_t*)d.q; a.k = (const bf16_t*)d.k; a.v = (const bf16_t*)d.v;
    a.row = d.q_row_stride; a.batch = d.q_batch_stride; a.kvrow = d.kv_row_stride; a.kvbatch = d.kv_batch_stride; a.head = d.head_stride;
    a.tq = d.tq; a.t = d.tk; a.causal = d.causal ? 1 : 0; a.mb = (d.tk + 63) / 64 * 64 + 64;
    a.key_mask = d.key_mask; a.kinfo = d.key_info; a.O = (bf16_t*)d.o; a.o_row = d.o_row_stride; a.o_batch = d.o_batch_stride;
    a.stats = d.stats; a.keep = d.keep_bits; a.p_batch = d.p_batch_stride; a.B = d.B; a.H = d.H; a.tp = d.tkp; a.nkt = (d.tk + 63) / 64;
    a.alpha = d.alpha; a.pdrop = d.p; a.rng = d.rng; a.site = d.site;
    a.dO = (const bf16_t*)d.d_out; a.do_row = d.do_row_stride; a.do_batch = d.do_batch_stride; a.aux = d.aux;
    a.dq = (bf16_t*)d.dq; a.dk = (bf16_t*)d.dk_out; a.dv = (bf16_t*)d.dv_out;
    a.g_row = d.dq_row_stride; a.g_batch = d.dq_batch_stride; a.gkv_row = d.dkv_row_stride; a.gkv_batch = d.dkv_batch_stride;
    a.dbq = d.dbias_q; a.dbk = d.dbias_k; a.dbv = d.dbias_v;
    return a;
}

int flash_grid(int B, int H, int rows, int blk) { return 8 * ((B * H + 7) / 8) * ((rows + blk - 1) / blk); }

template <int DK>
int launch_fwd(const FlashArgs& a, int drop_mode, hipStream_t st) {
    const int lds = 4 * TILE + a.mb + 16;
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_k<0, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE + MASK_MAX + 128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_k<1, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE + MASK_MAX + 128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_k<2, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE + MASK_MAX + 128);
    }
    const dim3 grid(flash_grid(a.B, a.H, a.tq, 128));
    if (drop_mode == 2) hipLaunchKernelGGL((flash_fwd_k<2, DK>), grid, dim3(512), lds, st, a);
    else if (drop_mode == 1) hipLaunchKernelGGL((flash_fwd_k<1, DK>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((flash_fwd_k<0, DK>), grid, dim3(512), lds, st, a);
    FS2_CHECK_LAUNCH("fs2_flash_attn_fwd");
    return FS2_OK;
}

template <int DK>
int launch_bwd(const FlashArgs& a, hipStream_t st) {
    const int lds_q = 4 * TILE + a.mb + 16, lds_kv = 4 * TILE + 2 * AUX_BYTES + 16;
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dq_k<false, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE + MASK_MAX + 128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dq_k<true, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE + MASK_MAX + 128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<false, 4, DK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<true, 4, DK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<false, 4, DK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_bwd_dkv_k<true, 4, DK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    }
    // dK/dV: 64 keys per workgroup (4 waves, two workgroups per CU).  Every non-empty key block costs the same (all queries), and at
    // config 2 there are ~2.1 blocks of 128 keys per CU: three rounds for two rounds' worth of work.  Half-size blocks leave a
    // shorter tail (a lone workgroup on a CU also runs faster than one of a pair): 149 -> 140 us per decoder layer, at twice the
    // Q / dO staging traffic (the 8-wave / 128-key form was measured slower and is no longer compiled).
    const dim3 grid(flash_grid(a.B, a.H, a.tq, 128)), grid_kv(flash_grid(a.B, a.H, a.t, 64));
    if (a.pdrop > 0.f) {
        hipLaunchKernelGGL((flash_bwd_dq_k<true, DK>), grid, dim3(FQ_THREADS), lds_q, st, a);
        if (a.causal) hipLaunchKernelGGL((flash_bwd_dkv_k<true, 4, DK, true>), grid_kv, dim3(256), lds_kv, st, a);
        else hipLaunchKernelGGL((flash_bwd_dkv_k<true, 4, DK, false>), grid_kv, dim3(256), lds_kv, st, a);
    } else {
        hipLaunchKernelGGL((flash_bwd_dq_k<false, DK>), grid, dim3(FQ_THREADS), lds_q, st, a);
        if (a.causal) hipLaunchKernelGGL((flash_bwd_dkv_k<false, 4, DK, true>), grid_kv, dim3(256), lds_kv, st, a);
        else hipLaunchKernelGGL((flash_bwd_dkv_k<false, 4, DK, false>), grid_kv, dim3(256), lds_kv, st, a);
    }
    FS2_CHECK_LAUNCH("fs2_flash_attn_bwd");
    return FS2_OK;
}

template <int DK>
int launch_probs(const FlashArgs& a, bf16_t* pm, int64_t pm_batch, hipStream_t st) {
    const int lds = 2 * TILE + a.mb + 16;
    static Fs2PerDevice attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_probs_k<false, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE + MASK_MAX + 128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_probs_k<true, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE + MASK_MAX + 128);
    }
    const dim3 grid(flash_grid(a.B, a.H, a.tq, 128));
    if (a.pdrop > 0.f) hipLaunchKernelGGL((flash_probs_k<true, DK>), grid, dim3(512), lds, st, a, pm, pm_batch);
    else hipLaunchKernelGGL((flash_probs_k<false, DK>), grid, dim3(512), lds, st, a, pm, pm_batch);
    FS2_CHECK_LAUNCH("fs2_flash_attention_probs");
    return FS2_OK;
}

}  // namespace

extern "C" int64_t fs2_flash_attn_keep_words(int B, int H, int t) { return (int64_t)B * H * ((t + 63) / 64) * t * 4; }
extern "C" int64_t fs2_flash_attn_keep_words_rect(int B, int H, int tq, int tk) { return (int64_t)B * H * ((tk + 63) / 64) * tq * 4; }

extern "C" int fs2_flash_attention_fwd(const FS2FlashAttn* dp, void* stream) {
    FS2_REQUIRE(dp != nullptr, "fs2_flash_attention_fwd: null descriptor");
    const FS2FlashAttn& d = *dp;
    const int rc = check_desc("fs2_flash_attention_fwd", d, false);
    if (rc != FS2_OK) return rc;
    const FlashArgs a = to_args(d);
    const int mode = d.p > 0.f ? (d.pregenerated ? 2 : 1) : 0;
    if (d.dk == 128) return launch_fwd<128>(a, mode, (hipStream_t)stream);
    if (d.dk == 96) return launch_fwd<96>(a, mode, (hipStream_t)stream);
    return launch_fwd<64>(a, mode, (hipStream_t)stream);
}

extern "C" int fs2_flash_attention_probs(const FS2FlashAttn* dp, void* probs, int64_t probs_batch_stride, void* stream) {
    FS2_REQUIRE(dp != nullptr && probs != nullptr, "fs2_flash_attention_probs: null argument");
    const FS2FlashAttn& d = *dp;
    const int rc = check_desc("fs2_flash_attention_probs", d, false);
    if (rc != FS2_OK) return rc;
    FS2_REQUIRE(probs_batch_stride % 4 == 0 && ((uintptr_t)probs & 7) == 0, "fs2_flash_attention_probs: map rows must be 8-byte aligned");
    const FlashArgs a = to_args(d);
    if (d.dk == 128) return launch_probs<128>(a, (bf16_t*)probs, probs_batch_stride, (hipStream_t)stream);
    if (d.dk == 96) return launch_probs<96>(a, (bf16_t*)probs, probs_batch_stride, (hipStream_t)stream);
    return launch_probs<64>(a, (bf16_t*)probs, probs_batch_stride, (hipStream_t)stream);
}

extern "C" int fs2_flash_attention_bwd(const FS2FlashAttn* dp, void* stream) {
    FS2_REQUIRE(dp != nullptr, "fs2_flash_attention_bwd: null descriptor");
    const FS2FlashAttn& d = *dp;
    const int rc = check_desc("fs2_flash_attention_bwd", d, true);
    if (rc != FS2_OK) return rc;
    const FlashArgs a = to_args(d);
    if (d.dk == 128) return launch_bwd<128>(a, (hipStream_t)stream);
    if (d.dk == 96) return launch_bwd<96>(a, (hipStream_t)stream);
    return launch_bwd<64>(a, (hipStream_t)stream);
}

// ---- the round-2 entry points: self-attention, d_k = 128, q / k / v rows of one fused projection
extern "C" int fs2_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, int64_t batch_stride, int head_stride,
                                  const uint8_t* key_mask, const int32_t* key_info, void* o_out, int64_t o_row_stride, int64_t o_batch_stride,
                                  float* stats, uint16_t* keep_bits, int pregenerated, int64_t p_batch_stride, int B, int H, int t, int tp, float alpha,
                                  float p, const uint64_t* rng, uint32_t site, void* stream) {
    FS2FlashAttn d = {};
    d.q = q; d.k = k; d.v = v; d.q_row_stride = d.kv_row_stride = row_stride; d.q_batch_stride = d.kv_batch_stride = batch_stride; d.head_stride = head_stride;
    d.key_mask = key_mask; d.key_info = key_info; d.o = o_out; d.o_row_stride = o_row_stride; d.o_batch_stride = o_batch_stride; d.stats = stats;
    d.keep_bits = keep_bits; d.pregenerated = pregenerated; d.p_batch_stride = p_batch_stride; d.B = B; d.H = H; d.tq = d.tk = t; d.tkp = tp; d.dk = 128;
    d.alpha = alpha; d.p = p; d.rng = rng; d.site = site;
    return fs2_flash_attention_fwd(&d, stream);
}

extern "C" int fs2_flash_attn_mask_info(const uint8_t* key_mask, int B, int t, int32_t* info, void* stream) {
    FS2_REQUIRE(key_mask && info && B > 0 && t > 0 && t <= MASK_MAX, "fs2_flash_attn_mask_info: bad arguments");
    hipLaunchKernelGGL(flash_mask_info_k, dim3(B), dim3(512), 0, (hipStream_t)stream, key_mask, t, info);
    FS2_CHECK_LAUNCH("fs2_flash_attn_mask_info");
    hipLaunchKernelGGL(flash_order_k, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, info);
    FS2_CHECK_LAUNCH("fs2_flash_attn_mask_info (order)");
    return FS2_OK;
}

extern "C" int fs2_pad_mask_info(const int64_t* pos, int64_t ld, int64_t pad, int B, int t, uint8_t* mask, int32_t* info, uint32_t* ticket, void* stream) {
    FS2_REQUIRE(pos && mask && info && ticket && B > 0 && B <= INFO_MAXB && t > 0 && t <= MASK_MAX && ld >= t,
                "fs2_pad_mask_info: need pos, mask, info, ticket, 0 < B <= %d and 0 < t <= %d (B=%d t=%d)", INFO_MAXB, MASK_MAX, B, t);
    hipLaunchKernelGGL(pad_mask_info_k, dim3(B), dim3(256), 0, (hipStream_t)stream, pos, ld, pad, mask, B, t, info, ticket);
    FS2_CHECK_LAUNCH("fs2_pad_mask_info");
    return FS2_OK;
}

extern "C" int fs2_flash_attn_keep_bits(uint16_t* keep_bits, int64_t p_batch_stride, int B, int H, int t, int tp, float p, const uint64_t* rng,
                                       uint32_t site, void* stream) {
    FS2_REQUIRE(keep_bits && rng && p > 0.f && p < 1.f, "fs2_flash_attn_keep_bits: need a buffer, rng and 0 < p < 1");
    FS2_REQUIRE(B > 0 && H > 0 && t > 0 && t <= MASK_MAX && tp == (t + 7) / 8 * 8 && p_batch_stride % 8 == 0, "fs2_flash_attn_keep_bits: bad shape");
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
    FS2FlashAttn d = {};
    d.q = q; d.k = k; d.v = v; d.q_row_stride = d.kv_row_stride = row_stride; d.q_batch_stride = d.kv_batch_stride = batch_stride; d.head_stride = head_stride;
    d.key_mask = key_mask; d.key_info = key_info; d.o = const_cast<void*>(o_saved); d.o_row_stride = o_row_stride; d.o_batch_stride = o_batch_stride;
    d.stats = const_cast<float*>(stats); d.keep_bits = const_cast<uint16_t*>(keep_bits); d.B = B; d.H = H; d.tq = d.tk = t; d.tkp = (t + 7) / 8 * 8; d.dk = 128;
    d.alpha = alpha; d.p = p;
    d.d_out = d_out; d.do_row_stride = do_row_stride; d.do_batch_stride = do_batch_stride; d.aux = aux; d.dq = dq; d.dk_out = dk; d.dv_out = dv;
    d.dq_row_stride = d.dkv_row_stride = g_row_stride; d.dq_batch_stride = d.dkv_batch_stride = g_batch_stride;
    d.dbias_q = dbias_q; d.dbias_k = dbias_k; d.dbias_v = dbias_v;
    return fs2_flash_attention_bwd(&d, stream);
}

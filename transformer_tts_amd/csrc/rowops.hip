// Row-wise (one 64-lane wave per row) HBM-bound kernels of the FastSpeech2 path for gfx950:
// LayerNorm family with fused residual / dropout, attention softmax, positional-encoding add,
// the variance-predictor head and BatchNorm+tanh.  Every lane moves 16-byte (f32) / 8-byte (bf16)
// groups of 4 consecutive channels, so a wave-instruction covers 1 KiB / 512 B contiguous bytes.
// A block is 4 waves; waves stride over rows, and per-channel reductions over rows (dgamma, dbeta,
// BatchNorm sums) are kept in registers, combined across the 4 waves in LDS and flushed with one
// float atomic per channel per block.
#include <stdlib.h>
#include "fs2_common.h"

namespace {

constexpr int ROW_BLOCK = 256;  // 4 waves
constexpr int ROW_WAVES = 4;
constexpr int RED_BLOCK = 1024; // 16 waves (row-reducing kernels)
constexpr int RED_WAVES = 16;
static inline int red_grid(int64_t rows) {
    // every block ends with one contended float atomic per channel: at most 256 blocks, and at least 4 rows per wave
    // (M = 6144 ran 384 blocks of one row per wave: 19 us for 25 MB, most of it the atomic tail; whole step, A/B on one box:
    // cap 128: 9.69 ms, 256: 9.33, 512: 9.39, 1024: 9.52)
    int64_t b = (rows + 4 * RED_WAVES - 1) / (4 * RED_WAVES);
    static const int cap = getenv("FS2_RED_BLOCKS") ? atoi(getenv("FS2_RED_BLOCKS")) : 256;      // (measurement switch)
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

static inline int row_grid(int64_t rows) {
    int64_t b = (rows + ROW_WAVES - 1) / ROW_WAVES;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

#define ROW_LOOP_W(M, W)                                                                     \
    const int lane = threadIdx.x & 63;                                                       \
    const int wave = threadIdx.x >> 6;                                                       \
    for (int64_t row = (int64_t)blockIdx.x * (W) + wave; row < (M); row += (int64_t)gridDim.x * (W))
#define ROW_LOOP(M) ROW_LOOP_W(M, ROW_WAVES)
// Kernels that reduce over rows (dgamma, dbeta, bias gradients, BatchNorm sums) run 16-wave blocks: every block
// ends with one float atomic per channel onto the SAME few hundred addresses, and contended atomics serialise
// (MI355X_MICROARCH.md, "Global float atomics": ~14x slower when every workgroup adds into one row) -- so the
// number of blocks, not the bytes, sets that tail: <= 512 blocks instead of 2048.
#define RED_LOOP(M) ROW_LOOP_W(M, RED_WAVES)

// column of group g for this lane, and whether it is inside the row
#define GCOL(g) (4 * (lane + 64 * (g)))

template <int NG, typename T>
__device__ __forceinline__ void row_load(const T* p, int d, int lane, float4 (&v)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) v[g] = (GCOL(g) < d) ? load4<T>(p + GCOL(g)) : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int NG, typename T>
__device__ __forceinline__ void row_store(T* p, int d, int lane, const float4 (&v)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
        if (GCOL(g) < d) store4<T>(p + GCOL(g), v[g]);
}
// ---- fp8 copy of a kernel's bf16 row output, written by the kernel itself (fs2_q8_next: the launch behind it): codes with the scale
//      of the amax this tensor had one step ago, the new amax reduced per thread -> workgroup -> one atomic (fs2_quantize_fp8_repair
//      re-quantises when the binade moved: the codes equal fs2_amax + fs2_quantize_fp8 of the stored tensor, always)
struct Q8Arg { unsigned char* q; float* state; const float* prev; int bf8; };
struct Q8Run { float scale, fmax, amax; };
__device__ __forceinline__ Q8Run q8_begin(const Q8Arg& qa) {
    Q8Run r = {1.f, 448.f, 0.f};
    if (qa.q != nullptr) {
        float inv;
        r.scale = fs2_pow2_scale(qa.prev[0], qa.bf8 ? 15 : 8, &inv);
        r.fmax = qa.bf8 ? 57344.f : 448.f;
    }
    return r;
}
template <int NG>
__device__ __forceinline__ void row_store_q8(const Q8Arg& qa, Q8Run& r, int64_t rowoff, int d, int lane, const float4 (&v)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (GCOL(g) < d) {
            const float e[4] = {(float)(bf16_t)v[g].x, (float)(bf16_t)v[g].y, (float)(bf16_t)v[g].z, (float)(bf16_t)v[g].w};   // as stored
            float c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = __uint_as_float(__float_as_uint(e[k]) & 0x7FFFFFFFu);
                r.amax = (a == a) ? fmaxf(r.amax, a) : r.amax;
                c[k] = fminf(fmaxf(e[k] * r.scale, -r.fmax), r.fmax);
            }
            int w = 0;
            if (qa.bf8) { w = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], w, true); }
            else { w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true); }
            *reinterpret_cast<int*>(qa.q + rowoff + GCOL(g)) = w;
        }
    }
}
// one atomic per workgroup; `slots` = LDS floats the caller can spare (>= blockDim / 64), every thread of the block calls it
__device__ __forceinline__ void q8_finish(const Q8Arg& qa, const Q8Run& r, float* slots) {
    if (qa.q == nullptr) return;
    const float m = wave_max(r.amax);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = slots[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) b = fmaxf(b, slots[w]);
        if (b > 0.f) atomicMax(reinterpret_cast<unsigned*>(qa.state), __float_as_uint(b));
    }
}
// Raw (unconverted) groups of 4 channels: the row-reducing backward kernels request the operands of the NEXT row of a wave before
// they process the current one (two rows in flight per wave: a wave otherwise sits out one full memory round trip per row, and
// 16 waves per CU x 5 KB per row did not cover the HBM latency: 3.1 TB/s for the fused FeedForward-tail backward).  The bf16 -> f32
// conversion happens at the use, so the wait for a row's data sits there and not at the request.
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef bf16x4 type; };
template <typename T> __device__ __forceinline__ float4 cvt4(typename Raw4<T>::type v);
template <> __device__ __forceinline__ float4 cvt4<float>(float4 v) { return v; }
template <> __device__ __forceinline__ float4 cvt4<bf16_t>(bf16x4 v) { return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }
template <int NG, typename T>
__device__ __forceinline__ void row_load_raw(const T* p, int d, int lane, typename Raw4<T>::type (&v)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = GCOL(g) < d ? GCOL(g) : 0;          // (columns past the row re-read column 0: no branch around the load; unused)
        v[g] = *reinterpret_cast<const typename Raw4<T>::type*>(p + c);
    }
}
template <int NG, typename T>
__device__ __forceinline__ void row_cvt(const typename Raw4<T>::type (&r)[NG], int d, int lane, float4 (&v)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) v[g] = (GCOL(g) < d) ? cvt4<T>(r[g]) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float sum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// mean and 1/sqrt(var+eps) of a row held in registers (two-pass, as torch's LayerNorm does)
template <int NG>
__device__ __forceinline__ void row_stats(const float4 (&v)[NG], int d, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) s += sum4(v[g]);  // out-of-row groups are zero
    mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
        if (GCOL(g) < d) {
            float4 c = make_float4(v[g].x - mean, v[g].y - mean, v[g].z - mean, v[g].w - mean);
            q += sum4(mul4(c, c));
        }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
}

// flush per-lane per-channel partial sums: LDS across the 4 waves, then one atomic per channel
template <int NG>
__device__ __forceinline__ void flush_channel_sums(const float4 (&acc)[NG], float* out, int d, float* lds /* [waves][NG*256] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) *reinterpret_cast<float4*>(lds + wave * NG * 256 + GCOL(g)) = acc[g];
    __syncthreads();
    for (int c = threadIdx.x; c < NG * 256; c += blockDim.x) {
        if (c < d) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += lds[w * NG * 256 + c];
            atomicAdd(out + c, s);
        }
    }
}

// ================================================================ LayerNorm (+dropout)
template <typename TX, typename TY, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void layernorm_fwd_k(const TX* __restrict__ x, const float* __restrict__ gamma,
        const float* __restrict__ beta, TY* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
        int64_t M, int d, float eps, float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 gm[NG], bt[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
        row_load<NG, float>(beta, d, lane, bt);
    }
    ROW_LOOP(M) {
        float4 v[NG];
        row_load<NG, TX>(x + row * d, d, lane, v);
        float mu, rs;
        row_stats<NG>(v, d, lane, eps, mu, rs);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 o;
            o.x = (v[g].x - mu) * rs * gm[g].x + bt[g].x; o.y = (v[g].y - mu) * rs * gm[g].y + bt[g].y;
            o.z = (v[g].z - mu) * rs * gm[g].z + bt[g].z; o.w = (v[g].w - mu) * rs * gm[g].w + bt[g].w;
            if (dc.on && GCOL(g) < d) o = mul4(o, drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
            v[g] = o;
        }
        row_store<NG, TY>(y + row * d, d, lane, v);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <typename TDY, typename TX, typename TDX, int NG>
__global__ __launch_bounds__(RED_BLOCK) void layernorm_bwd_k(const TDY* __restrict__ dy, const TX* __restrict__ x,
        const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
        TDX* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t M, int d, float p,
        const uint64_t* rng, uint32_t site, int relu_mask, int dx_accumulate, float* __restrict__ dcolsum) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 gm[NG], ag[NG], ab[NG], ac[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
#pragma unroll
        for (int g = 0; g < NG; ++g) ag[g] = ab[g] = ac[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invd = 1.f / (float)d;
    RED_LOOP(M) {
        float4 g_[NG], xv[NG], old[NG];
        row_load<NG, TDY>(dy + row * d, d, lane, g_);
        row_load<NG, TX>(x + row * d, d, lane, xv);
        if (dx_accumulate) row_load<NG, TDX>(dx + row * d, d, lane, old);      // requested with the others, used after the reduction
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
        unsigned pos[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            pos[g] = 0xFu;
            if (GCOL(g) < d) {
                if (dc.on) g_[g] = mul4(g_[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
                if (relu_mask)  // the LayerNorm input was relu(z): remember where z > 0
                    pos[g] = (xv[g].x > 0.f ? 1u : 0u) | (xv[g].y > 0.f ? 2u : 0u) | (xv[g].z > 0.f ? 4u : 0u) | (xv[g].w > 0.f ? 8u : 0u);
                float4 xh = make_float4((xv[g].x - mu) * rs, (xv[g].y - mu) * rs, (xv[g].z - mu) * rs, (xv[g].w - mu) * rs);
                ag[g] = add4(ag[g], mul4(g_[g], xh));
                ab[g] = add4(ab[g], g_[g]);
                float4 dg = mul4(g_[g], gm[g]);
                c1 += sum4(dg);
                c2 += sum4(mul4(dg, xh));
                g_[g] = dg;   // dy * gamma
                xv[g] = xh;
            }
        }
        c1 = wave_sum(c1) * invd;
        c2 = wave_sum(c2) * invd;
        float4 o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            o[g].x = rs * (g_[g].x - c1 - xv[g].x * c2); o[g].y = rs * (g_[g].y - c1 - xv[g].y * c2);
            o[g].z = rs * (g_[g].z - c1 - xv[g].z * c2); o[g].w = rs * (g_[g].w - c1 - xv[g].w * c2);
            if (relu_mask) {
                o[g].x = (pos[g] & 1u) ? o[g].x : 0.f; o[g].y = (pos[g] & 2u) ? o[g].y : 0.f;
                o[g].z = (pos[g] & 4u) ? o[g].z : 0.f; o[g].w = (pos[g] & 8u) ? o[g].w : 0.f;
            }
            ac[g] = add4(ac[g], o[g]);
        }
        if (dx_accumulate) {
#pragma unroll
            for (int g = 0; g < NG; ++g) o[g] = add4(o[g], old[g]);
        }
        row_store<NG, TDX>(dx + row * d, d, lane, o);
    }
    flush_channel_sums<NG>(ag, dgamma, d, red);
    flush_channel_sums<NG>(ab, dbeta, d, red);
    if (dcolsum != nullptr) flush_channel_sums<NG>(ac, dcolsum, d, red);
}

// ================================================================ s = r + dropout(a); y = LN(s)
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void add_ln_fwd_k(const float* __restrict__ r, const T* __restrict__ a,
        float* __restrict__ s, const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
        float* __restrict__ mean, float* __restrict__ rstd, int64_t M, int d, float eps, float p, const uint64_t* rng,
        uint32_t site, const Q8Arg qa) {
    __shared__ float q8s[ROW_BLOCK / 64];
    const DropCtx dc = drop_ctx(rng, site, p);
    Q8Run qr = q8_begin(qa);
    float4 gm[NG], bt[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
        row_load<NG, float>(beta, d, lane, bt);
    }
    ROW_LOOP(M) {
        float4 v[NG], av[NG];
        row_load<NG, float>(r + row * d, d, lane, v);
        row_load<NG, T>(a + row * d, d, lane, av);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (dc.on && GCOL(g) < d) av[g] = mul4(av[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
            v[g] = add4(v[g], av[g]);
        }
        row_store<NG, float>(s + row * d, d, lane, v);
        float mu, rs;
        row_stats<NG>(v, d, lane, eps, mu, rs);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g].x = (v[g].x - mu) * rs * gm[g].x + bt[g].x; v[g].y = (v[g].y - mu) * rs * gm[g].y + bt[g].y;
            v[g].z = (v[g].z - mu) * rs * gm[g].z + bt[g].z; v[g].w = (v[g].w - mu) * rs * gm[g].w + bt[g].w;
        }
        row_store<NG, T>(y + row * d, d, lane, v);
        if (qa.q != nullptr) row_store_q8<NG>(qa, qr, row * d, d, lane, v);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
    q8_finish(qa, qr, q8s);
}

template <typename T, int NG>
__global__ __launch_bounds__(RED_BLOCK) void add_ln_bwd_k(const float* __restrict__ ds_down, const T* __restrict__ dy,
        const float* __restrict__ s, const float* __restrict__ gamma, const float* __restrict__ mean,
        const float* __restrict__ rstd, float* __restrict__ dr, T* __restrict__ da, float* __restrict__ dgamma,
        float* __restrict__ dbeta, int64_t M, int d, float p, const uint64_t* rng, uint32_t site,
        float* __restrict__ dcolsum, const Q8Arg qa) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc = drop_ctx(rng, site, p);
    Q8Run qr = q8_begin(qa);
    float4 gm[NG], ag[NG], ab[NG], ac[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
#pragma unroll
        for (int g = 0; g < NG; ++g) ag[g] = ab[g] = ac[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invd = 1.f / (float)d;
    RED_LOOP(M) {
        float4 g_[NG], xv[NG], dn[NG];
        row_load<NG, T>(dy + row * d, d, lane, g_);
        row_load<NG, float>(s + row * d, d, lane, xv);
        if (ds_down != nullptr) row_load<NG, float>(ds_down + row * d, d, lane, dn);     // requested with the others, used after the reduction
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (GCOL(g) < d) {
                float4 xh = make_float4((xv[g].x - mu) * rs, (xv[g].y - mu) * rs, (xv[g].z - mu) * rs, (xv[g].w - mu) * rs);
                ag[g] = add4(ag[g], mul4(g_[g], xh));
                ab[g] = add4(ab[g], g_[g]);
                float4 dg = mul4(g_[g], gm[g]);
                c1 += sum4(dg);
                c2 += sum4(mul4(dg, xh));
                g_[g] = dg;
                xv[g] = xh;
            }
        }
        c1 = wave_sum(c1) * invd;
        c2 = wave_sum(c2) * invd;
        float4 o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            o[g].x = rs * (g_[g].x - c1 - xv[g].x * c2); o[g].y = rs * (g_[g].y - c1 - xv[g].y * c2);
            o[g].z = rs * (g_[g].z - c1 - xv[g].z * c2); o[g].w = rs * (g_[g].w - c1 - xv[g].w * c2);
            if (ds_down != nullptr) o[g] = add4(o[g], dn[g]);
        }
        row_store<NG, float>(dr + row * d, d, lane, o);
#pragma unroll
        for (int g = 0; g < NG; ++g)
        {
            if (dc.on && GCOL(g) < d) o[g] = mul4(o[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
            ac[g] = add4(ac[g], o[g]);
        }
        row_store<NG, T>(da + row * d, d, lane, o);
        if (qa.q != nullptr) row_store_q8<NG>(qa, qr, row * d, d, lane, o);
    }
    flush_channel_sums<NG>(ag, dgamma, d, red);
    flush_channel_sums<NG>(ab, dbeta, d, red);
    if (dcolsum != nullptr) flush_channel_sums<NG>(ac, dcolsum, d, red);
    q8_finish(qa, qr, red);
}

// ================================================================ y = LN(dropout(f2 + h))
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void ffn_ln_fwd_k(const T* __restrict__ f2, const T* __restrict__ h,
        const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean,
        float* __restrict__ rstd, int64_t M, int d, float eps, float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 gm[NG], bt[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
        row_load<NG, float>(beta, d, lane, bt);
    }
    ROW_LOOP(M) {
        float4 v[NG], hv[NG];
        row_load<NG, T>(f2 + row * d, d, lane, v);
        row_load<NG, T>(h + row * d, d, lane, hv);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g] = add4(v[g], hv[g]);
            if (dc.on && GCOL(g) < d) v[g] = mul4(v[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
        }
        float mu, rs;
        row_stats<NG>(v, d, lane, eps, mu, rs);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g].x = (v[g].x - mu) * rs * gm[g].x + bt[g].x; v[g].y = (v[g].y - mu) * rs * gm[g].y + bt[g].y;
            v[g].z = (v[g].z - mu) * rs * gm[g].z + bt[g].z; v[g].w = (v[g].w - mu) * rs * gm[g].w + bt[g].w;
        }
        row_store<NG, T>(y + row * d, d, lane, v);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <typename T, int NG>
__global__ __launch_bounds__(RED_BLOCK) void ffn_ln_bwd_k(const T* __restrict__ dy, const T* __restrict__ f2,
        const T* __restrict__ h, const float* __restrict__ gamma, const float* __restrict__ mean,
        const float* __restrict__ rstd, T* __restrict__ gout, float* __restrict__ dgamma, float* __restrict__ dbeta,
        int64_t M, int d, float p, const uint64_t* rng, uint32_t site, float* __restrict__ dcolsum) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 gm[NG], ag[NG], ab[NG], ac[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
#pragma unroll
        for (int g = 0; g < NG; ++g) ag[g] = ab[g] = ac[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invd = 1.f / (float)d;
    RED_LOOP(M) {
        float4 g_[NG], xv[NG], hv[NG], ds[NG];
        row_load<NG, T>(dy + row * d, d, lane, g_);
        row_load<NG, T>(f2 + row * d, d, lane, xv);
        row_load<NG, T>(h + row * d, d, lane, hv);
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            ds[g] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (GCOL(g) < d) {
                if (dc.on) ds[g] = drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2);
                float4 u = mul4(add4(xv[g], hv[g]), ds[g]);
                float4 xh = make_float4((u.x - mu) * rs, (u.y - mu) * rs, (u.z - mu) * rs, (u.w - mu) * rs);
                ag[g] = add4(ag[g], mul4(g_[g], xh));
                ab[g] = add4(ab[g], g_[g]);
                float4 dg = mul4(g_[g], gm[g]);
                c1 += sum4(dg);
                c2 += sum4(mul4(dg, xh));
                g_[g] = dg;
                xv[g] = xh;
            }
        }
        c1 = wave_sum(c1) * invd;
        c2 = wave_sum(c2) * invd;
        float4 o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            o[g].x = rs * (g_[g].x - c1 - xv[g].x * c2) * ds[g].x; o[g].y = rs * (g_[g].y - c1 - xv[g].y * c2) * ds[g].y;
            o[g].z = rs * (g_[g].z - c1 - xv[g].z * c2) * ds[g].z; o[g].w = rs * (g_[g].w - c1 - xv[g].w * c2) * ds[g].w;
            ac[g] = add4(ac[g], o[g]);
        }
        row_store<NG, T>(gout + row * d, d, lane, o);
    }
    flush_channel_sums<NG>(ag, dgamma, d, red);
    flush_channel_sums<NG>(ab, dbeta, d, red);
    if (dcolsum != nullptr) flush_channel_sums<NG>(ac, dcolsum, d, red);
}

// ================================================================ FeedForward tail + the residual add and LayerNorm that follow it
// EncoderLayer.forward after the second convolution (Models/modules.py:85-87, Models/layers.py:40,31 / encoder.py:112):
//   yff = LN1(dropout1(f2 + h));  s = r + dropout2(yff);  y = LN2(s)
// = fs2_ffn_ln_fwd followed by fs2_add_ln_fwd without the round trip of yff through HBM (one row pass instead of two).  yff is
// rounded to the compute dtype exactly where the two-kernel form stores it, so both forms agree to the last place.
template <typename T> __device__ __forceinline__ float4 round_to(float4 v);
template <> __device__ __forceinline__ float4 round_to<float>(float4 v) { return v; }
template <> __device__ __forceinline__ float4 round_to<bf16_t>(float4 v) {
    return make_float4((float)(bf16_t)v.x, (float)(bf16_t)v.y, (float)(bf16_t)v.z, (float)(bf16_t)v.w);
}
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void ffn_tail_fwd_k(const T* __restrict__ f2, const T* __restrict__ h,
        const float* __restrict__ r, const float* __restrict__ gamma1, const float* __restrict__ beta1,
        const float* __restrict__ gamma2, const float* __restrict__ beta2, float* __restrict__ s, T* __restrict__ y,
        float* __restrict__ mean1, float* __restrict__ rstd1, float* __restrict__ mean2, float* __restrict__ rstd2, int64_t M, int d,
        float eps, float p, const uint64_t* rng, uint32_t site1, uint32_t site2, const Q8Arg qa) {
    __shared__ float q8s[ROW_BLOCK / 64];
    const DropCtx dc1 = drop_ctx(rng, site1, p), dc2 = drop_ctx(rng, site2, p);
    Q8Run qr = q8_begin(qa);
    float4 g1[NG], b1[NG], g2[NG], b2[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma1, d, lane, g1); row_load<NG, float>(beta1, d, lane, b1);
        row_load<NG, float>(gamma2, d, lane, g2); row_load<NG, float>(beta2, d, lane, b2);
    }
    ROW_LOOP(M) {
        float4 v[NG], hv[NG], rv[NG];
        row_load<NG, T>(f2 + row * d, d, lane, v);
        row_load<NG, T>(h + row * d, d, lane, hv);
        row_load<NG, float>(r + row * d, d, lane, rv);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g] = add4(v[g], hv[g]);
            if (dc1.on && GCOL(g) < d) v[g] = mul4(v[g], drop_scale4(dc1, (uint64_t)(row * d + GCOL(g)) >> 2));
        }
        float mu, rs;
        row_stats<NG>(v, d, lane, eps, mu, rs);
        if (lane == 0) { mean1[row] = mu; rstd1[row] = rs; }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g].x = (v[g].x - mu) * rs * g1[g].x + b1[g].x; v[g].y = (v[g].y - mu) * rs * g1[g].y + b1[g].y;
            v[g].z = (v[g].z - mu) * rs * g1[g].z + b1[g].z; v[g].w = (v[g].w - mu) * rs * g1[g].w + b1[g].w;
            v[g] = round_to<T>(v[g]);                        // yff as the two-kernel form stores it
            if (dc2.on && GCOL(g) < d) v[g] = mul4(v[g], drop_scale4(dc2, (uint64_t)(row * d + GCOL(g)) >> 2));
            v[g] = (GCOL(g) < d) ? add4(rv[g], v[g]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        row_store<NG, float>(s + row * d, d, lane, v);
        row_stats<NG>(v, d, lane, eps, mu, rs);
        if (lane == 0) { mean2[row] = mu; rstd2[row] = rs; }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g].x = (v[g].x - mu) * rs * g2[g].x + b2[g].x; v[g].y = (v[g].y - mu) * rs * g2[g].y + b2[g].y;
            v[g].z = (v[g].z - mu) * rs * g2[g].z + b2[g].z; v[g].w = (v[g].w - mu) * rs * g2[g].w + b2[g].w;
        }
        row_store<NG, T>(y + row * d, d, lane, v);
        if (qa.q != nullptr) row_store_q8<NG>(qa, qr, row * d, d, lane, v);
    }
    q8_finish(qa, qr, q8s);
}

// backward of the same: fs2_add_ln_bwd (LN2, residual, dropout2) followed by fs2_ffn_ln_bwd (LN1, dropout1) in one row pass; the
// gradient handed from one to the other is rounded to the compute dtype where the two-kernel form stores it.
template <typename T, int NG>
__global__ __launch_bounds__(RED_BLOCK) void ffn_tail_bwd_k(const float* __restrict__ ds_down, const T* __restrict__ dy,
        const float* __restrict__ s, const float* __restrict__ gamma2, const float* __restrict__ mean2, const float* __restrict__ rstd2,
        const T* __restrict__ f2, const T* __restrict__ h, const float* __restrict__ gamma1, const float* __restrict__ mean1,
        const float* __restrict__ rstd1, float* __restrict__ dr, T* __restrict__ gout, float* __restrict__ dgamma2,
        float* __restrict__ dbeta2, float* __restrict__ dgamma1, float* __restrict__ dbeta1, float* __restrict__ dcolsum, int64_t M,
        int d, float p, const uint64_t* rng, uint32_t site1, uint32_t site2, const Q8Arg qa) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc1 = drop_ctx(rng, site1, p), dc2 = drop_ctx(rng, site2, p);
    Q8Run qr = q8_begin(qa);
    float4 gm2[NG], gm1[NG], ag2[NG], ab2[NG], ag1[NG], ab1[NG], ac[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma2, d, lane, gm2);
        row_load<NG, float>(gamma1, d, lane, gm1);
#pragma unroll
        for (int g = 0; g < NG; ++g) ag2[g] = ab2[g] = ag1[g] = ab1[g] = ac[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invd = 1.f / (float)d;
    // operands of one row, as loaded
    struct RowOps {
        typename Raw4<T>::type dy[NG], f2[NG], h[NG];
        float4 s[NG], dn[NG];
        float mu2, rs2, mu1, rs1;
    };
    const bool has_dn = ds_down != nullptr;
    auto request = [&](RowOps& o, int64_t row, int lane) __attribute__((always_inline)) {
        row_load_raw<NG, T>(dy + row * d, d, lane, o.dy);
        row_load_raw<NG, float>(s + row * d, d, lane, o.s);
        if (has_dn) row_load_raw<NG, float>(ds_down + row * d, d, lane, o.dn);
        row_load_raw<NG, T>(f2 + row * d, d, lane, o.f2);
        row_load_raw<NG, T>(h + row * d, d, lane, o.h);
        o.mu2 = mean2[row]; o.rs2 = rstd2[row]; o.mu1 = mean1[row]; o.rs1 = rstd1[row];
    };
    auto process = [&](const RowOps& in, int64_t row, int lane) __attribute__((always_inline)) {
        // ---- LN2 backward + residual + dropout2'  (add_ln_bwd_k)
        float4 g_[NG], xv[NG], dn[NG], fv[NG], hv[NG];
        row_cvt<NG, T>(in.dy, d, lane, g_);
        row_cvt<NG, float>(in.s, d, lane, xv);
        if (has_dn) row_cvt<NG, float>(in.dn, d, lane, dn);
        row_cvt<NG, T>(in.f2, d, lane, fv);
        row_cvt<NG, T>(in.h, d, lane, hv);
        float mu = in.mu2, rs = in.rs2;
        const float mu1 = in.mu1, rs1 = in.rs1;
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (GCOL(g) < d) {
                float4 xh = make_float4((xv[g].x - mu) * rs, (xv[g].y - mu) * rs, (xv[g].z - mu) * rs, (xv[g].w - mu) * rs);
                ag2[g] = add4(ag2[g], mul4(g_[g], xh));
                ab2[g] = add4(ab2[g], g_[g]);
                float4 dg = mul4(g_[g], gm2[g]);
                c1 += sum4(dg);
                c2 += sum4(mul4(dg, xh));
                g_[g] = dg;
                xv[g] = xh;
            }
        }
        c1 = wave_sum(c1) * invd;
        c2 = wave_sum(c2) * invd;
        float4 o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            o[g].x = rs * (g_[g].x - c1 - xv[g].x * c2); o[g].y = rs * (g_[g].y - c1 - xv[g].y * c2);
            o[g].z = rs * (g_[g].z - c1 - xv[g].z * c2); o[g].w = rs * (g_[g].w - c1 - xv[g].w * c2);
            if (has_dn) o[g] = add4(o[g], dn[g]);
        }
        row_store<NG, float>(dr + row * d, d, lane, o);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (dc2.on && GCOL(g) < d) o[g] = mul4(o[g], drop_scale4(dc2, (uint64_t)(row * d + GCOL(g)) >> 2));
            o[g] = round_to<T>(o[g]);                        // d(yff) as the two-kernel form stores it
        }
        // ---- LN1 backward + dropout1'  (ffn_ln_bwd_k with dy = o)
        float4 ds[NG];
        mu = mu1; rs = rs1;
        c1 = 0.f; c2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            ds[g] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (GCOL(g) < d) {
                if (dc1.on) ds[g] = drop_scale4(dc1, (uint64_t)(row * d + GCOL(g)) >> 2);
                float4 u = mul4(add4(fv[g], hv[g]), ds[g]);
                float4 xh = make_float4((u.x - mu) * rs, (u.y - mu) * rs, (u.z - mu) * rs, (u.w - mu) * rs);
                ag1[g] = add4(ag1[g], mul4(o[g], xh));
                ab1[g] = add4(ab1[g], o[g]);
                float4 dg = mul4(o[g], gm1[g]);
                c1 += sum4(dg);
                c2 += sum4(mul4(dg, xh));
                o[g] = dg;
                fv[g] = xh;
            }
        }
        c1 = wave_sum(c1) * invd;
        c2 = wave_sum(c2) * invd;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            o[g].x = rs * (o[g].x - c1 - fv[g].x * c2) * ds[g].x; o[g].y = rs * (o[g].y - c1 - fv[g].y * c2) * ds[g].y;
            o[g].z = rs * (o[g].z - c1 - fv[g].z * c2) * ds[g].z; o[g].w = rs * (o[g].w - c1 - fv[g].w * c2) * ds[g].w;
            ac[g] = add4(ac[g], o[g]);
        }
        row_store<NG, T>(gout + row * d, d, lane, o);
        if (qa.q != nullptr) row_store_q8<NG>(qa, qr, row * d, d, lane, o);
    };
    {
        // two rows in flight per wave: the operands of row r + stride are requested before row r is processed.  Requests past the
        // last row are clamped to it (always issued: a load under a branch makes hipcc drain vmcnt at every use) and not processed.
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int64_t stride = (int64_t)gridDim.x * RED_WAVES, last = M - 1;
        int64_t row = (int64_t)blockIdx.x * RED_WAVES + wave;
        if (row < M) {
            RowOps ra, rb;
            request(ra, row, lane);
            for (; row < M; row += 2 * stride) {
                const int64_t r1 = row + stride, r2 = row + 2 * stride;
                request(rb, r1 < M ? r1 : last, lane);
                process(ra, row, lane);
                request(ra, r2 < M ? r2 : last, lane);
                if (r1 < M) process(rb, r1, lane);
            }
        }
    }
    flush_channel_sums<NG>(ag2, dgamma2, d, red);
    flush_channel_sums<NG>(ab2, dbeta2, d, red);
    flush_channel_sums<NG>(ag1, dgamma1, d, red);
    flush_channel_sums<NG>(ab1, dbeta1, d, red);
    if (dcolsum != nullptr) flush_channel_sums<NG>(ac, dcolsum, d, red);
    q8_finish(qa, qr, red);
}

// ================================================================ attention softmax (in place) + dropout
// rows = (b, h, i); S row = base + b*batch_stride + (h*t + i)*tp; keys j < t; pad columns [t,tp) -> 0
// (tq query rows per head against t keys: tq == t for self-attention; causal: key j of query i is also masked when
//  j > i -- the decoder self-attention of the autoregressive model, reference train.py:26-58)
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void softmax_fwd_k(T* __restrict__ s, T* __restrict__ pd,
        const uint8_t* __restrict__ key_mask, int B, int H, int tq, int t, int tp, int64_t batch_stride, int causal, float p,
        const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const int64_t rows = (int64_t)B * H * tq;
    ROW_LOOP(rows) {
        const int b = (int)(row / ((int64_t)H * tq));
        const int64_t inb = row - (int64_t)b * H * tq;   // h*tq + i
        const int64_t off = b * batch_stride + inb * tp;
        const int jmax = causal ? (int)(inb % tq) : 0x7fffffff;      // last key a causal row may look at
        const uint8_t* km = key_mask + (int64_t)b * t;
        float4 v[NG];
        row_load<NG, T>(s + off, tp, lane, v);
        float mx = -3.0e38f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float* e = &v[g].x;
            unsigned mk4 = 0x01010101u;               // 4 key-mask bytes in one load when the group is fully inside
            const bool whole = GCOL(g) + 4 <= t && ((reinterpret_cast<uintptr_t>(km) + GCOL(g)) & 3) == 0;
            if (whole) mk4 = *reinterpret_cast<const unsigned*>(km + GCOL(g));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = GCOL(g) + c;
                if (j < t) {
                    const bool keep = (whole ? ((mk4 >> (8 * c)) & 0xFFu) != 0 : km[j] != 0) && j <= jmax;
                    if (!keep) e[c] = -1e4f;          // masked_fill(mask == 0, -1e4)
                    mx = fmaxf(mx, e[c]);
                } else e[c] = -3.0e38f;
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float* e = &v[g].x;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = GCOL(g) + c;
                e[c] = (j < t) ? __expf(e[c] - mx) : 0.f;
                sum += e[c];
            }
        }
        const float inv = 1.f / wave_sum(sum);
        float4 o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g] = scale4(v[g], inv);
            o[g] = v[g];
            if (dc.on && GCOL(g) < tp) o[g] = mul4(o[g], drop_scale4(dc, (uint64_t)(off + GCOL(g)) >> 2));
        }
        row_store<NG, T>(s + off, tp, lane, v);
        if (pd != s) row_store<NG, T>(pd + off, tp, lane, o);
    }
}

template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void softmax_bwd_k(T* __restrict__ dp, int64_t dp_stride, const T* __restrict__ ps,
        int64_t p_stride, int B, int H, int tq, int t, int tp, float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const int64_t rows = (int64_t)B * H * tq;
    ROW_LOOP(rows) {
        const int b = (int)(row / ((int64_t)H * tq));
        const int64_t inb = row - (int64_t)b * H * tq;
        const int64_t off = b * dp_stride + inb * tp;       // in the dP / dS buffer
        const int64_t poff = b * p_stride + inb * tp;       // in the saved-probabilities buffer (= the forward's offsets)
        float4 g_[NG], pv[NG];
        row_load<NG, T>(dp + off, tp, lane, g_);
        row_load<NG, T>(ps + poff, tp, lane, pv);
        float dot = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // pad columns [t,tp) of dP were never written by the GEMM: force them (and P's) to 0
            float* ge = &g_[g].x; float* pe_ = &pv[g].x;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (GCOL(g) + c >= t) { ge[c] = 0.f; pe_[c] = 0.f; }
            if (dc.on && GCOL(g) < tp) g_[g] = mul4(g_[g], drop_scale4(dc, (uint64_t)(poff + GCOL(g)) >> 2));
            dot += sum4(mul4(g_[g], pv[g]));
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            g_[g].x = pv[g].x * (g_[g].x - dot); g_[g].y = pv[g].y * (g_[g].y - dot);
            g_[g].z = pv[g].z * (g_[g].z - dot); g_[g].w = pv[g].w * (g_[g].w - dot);
        }
        row_store<NG, T>(dp + off, tp, lane, g_);
    }
}

// ---------------------------------------------------------------- bf16 fast path: 16-byte accesses, R rows per wave
// The t ~ 925 probability tensors are the largest streams of the step (165 MB per decoder layer); with 8-byte lane
// accesses and one row in flight per wave the kernels above reached ~2 TB/s.  Here a lane owns 8 consecutive keys per
// group (GCOL8), a wave issues the loads of R rows before it reduces the first one, and the key mask comes from
// aligned 32-bit words.  Same arithmetic, same Philox counters (element offset >> 2) as the generic kernels.
#define GCOL8(g) (8 * (lane + 64 * (g)))

__device__ __forceinline__ void unpack8(const bf16x8& v, float (&e)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) e[c] = (float)v[c];
}
__device__ __forceinline__ bf16x8 pack8(const float (&e)[8]) {
    bf16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (bf16_t)e[c];
    return o;
}
// bytes km[0..8) as a 64-bit word (byte c in bits 8c..8c+7); reads the aligned words that cover them
__device__ __forceinline__ uint64_t load_mask8(const uint8_t* km) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(km);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const unsigned sh = (unsigned)(a & 3) * 8;
    const uint32_t w0 = w[0], w1 = w[1];
    if (sh == 0) return ((uint64_t)w1 << 32) | w0;
    const uint32_t w2 = w[2];
    const uint32_t lo = (w0 >> sh) | (w1 << (32 - sh));
    const uint32_t hi = (w1 >> sh) | (w2 << (32 - sh));
    return ((uint64_t)hi << 32) | lo;
}

template <int NG8, int R>
__global__ __launch_bounds__(ROW_BLOCK) void softmax_fwd8_k(bf16_t* __restrict__ s, bf16_t* __restrict__ pd,
        const uint8_t* __restrict__ key_mask, int B, int H, int t, int tp, int64_t batch_stride, float p,
        const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const int64_t rows = (int64_t)B * H * t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row0 = ((int64_t)blockIdx.x * ROW_WAVES + wave) * R; row0 < rows; row0 += (int64_t)gridDim.x * ROW_WAVES * R) {
        bf16x8 raw[R][NG8];
        uint64_t mk[R][NG8];
        int64_t off[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r < rows ? row0 + r : rows - 1;      // a clamped duplicate; not stored
            const int b = (int)(row / ((int64_t)H * t));
            off[r] = b * batch_stride + (row - (int64_t)b * H * t) * tp;
            const uint8_t* km = key_mask + (int64_t)b * t;
#pragma unroll
            for (int g = 0; g < NG8; ++g) {
                const int col = GCOL8(g);
                if (col < tp) raw[r][g] = *reinterpret_cast<const bf16x8*>(s + off[r] + col);
                mk[r][g] = 0;
                if (col + 8 <= t) mk[r][g] = load_mask8(km + col);
                else
                    for (int c = 0; c < 8; ++c)
                        if (col + c < t && km[col + c] != 0) mk[r][g] |= (uint64_t)0xFF << (8 * c);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float e[NG8][8];
            float mx = -3.0e38f;
#pragma unroll
            for (int g = 0; g < NG8; ++g) {
                const int col = GCOL8(g);
                bf16x8 rw = {};
                if (col < tp) rw = raw[r][g];
                // branch-free: invalid columns (>= t) become -3e38 (exp -> 0), masked keys -1e4
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t mb = (uint32_t)(mk[r][g] >> (8 * c)) & 0xFFu;
                    float x = (float)rw[c];
                    x = mb != 0 ? x : -1e4f;              // masked_fill(mask == 0, -1e4) on keys
                    x = (col + c < t) ? x : -3.0e38f;
                    e[g][c] = x;
                    mx = fmaxf(mx, x);
                }
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < NG8; ++g)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    e[g][c] = __expf(e[g][c] - mx);
                    sum += e[g][c];
                }
            const float inv = 1.f / wave_sum(sum);
            if (row0 + r < rows) {
#pragma unroll
                for (int g = 0; g < NG8; ++g) {
                    const int col = GCOL8(g);
                    if (col >= tp) continue;
#pragma unroll
                    for (int c = 0; c < 8; ++c) e[g][c] *= inv;
                    *reinterpret_cast<bf16x8*>(s + off[r] + col) = pack8(e[g]);
                    if (pd != s) {
                        if (dc.on) {
                            float ds[8];
                            drop_scale8(dc, (uint64_t)(off[r] + col) >> 3, ds);
#pragma unroll
                            for (int c = 0; c < 8; ++c) e[g][c] *= ds[c];
                        }
                        *reinterpret_cast<bf16x8*>(pd + off[r] + col) = pack8(e[g]);
                    }
                }
            }
        }
    }
}

template <int NG8, int R>
__global__ __launch_bounds__(ROW_BLOCK) void softmax_bwd8_k(bf16_t* __restrict__ dp, int64_t dp_stride,
        const bf16_t* __restrict__ ps, int64_t p_stride, int B, int H, int t, int tp, float p, const uint64_t* rng,
        uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const int64_t rows = (int64_t)B * H * t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row0 = ((int64_t)blockIdx.x * ROW_WAVES + wave) * R; row0 < rows; row0 += (int64_t)gridDim.x * ROW_WAVES * R) {
        bf16x8 rg[R][NG8], rp[R][NG8];
        int64_t off[R], poff[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
            const int b = (int)(row / ((int64_t)H * t));
            const int64_t inb = row - (int64_t)b * H * t;
            off[r] = b * dp_stride + inb * tp;
            poff[r] = b * p_stride + inb * tp;
#pragma unroll
            for (int g = 0; g < NG8; ++g) {
                const int col = GCOL8(g);
                if (col < tp) {
                    rg[r][g] = *reinterpret_cast<const bf16x8*>(dp + off[r] + col);
                    rp[r][g] = *reinterpret_cast<const bf16x8*>(ps + poff[r] + col);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float ge[NG8][8], pe_[NG8][8];
            float dot = 0.f;
#pragma unroll
            for (int g = 0; g < NG8; ++g) {
                const int col = GCOL8(g);
                bf16x8 ag = {}, ap = {};
                if (col < tp) { ag = rg[r][g]; ap = rp[r][g]; }
                // pad columns [t,tp) of dP were never written by the GEMM: force them (and P's) to 0 (selects, no branches)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    ge[g][c] = (col + c < t) ? (float)ag[c] : 0.f;
                    pe_[g][c] = (col + c < t) ? (float)ap[c] : 0.f;
                }
                if (dc.on && col < tp) {
                    float ds[8];
                    drop_scale8(dc, (uint64_t)(poff[r] + col) >> 3, ds);
#pragma unroll
                    for (int c = 0; c < 8; ++c) ge[g][c] *= ds[c];
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) dot += ge[g][c] * pe_[g][c];
            }
            dot = wave_sum(dot);
            if (row0 + r < rows) {
#pragma unroll
                for (int g = 0; g < NG8; ++g) {
                    const int col = GCOL8(g);
                    if (col >= tp) continue;
#pragma unroll
                    for (int c = 0; c < 8; ++c) ge[g][c] = pe_[g][c] * (ge[g][c] - dot);
                    *reinterpret_cast<bf16x8*>(dp + off[r] + col) = pack8(ge[g]);
                }
            }
        }
    }
}

// ================================================================ positional encoding add
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void pe_add_fwd_k(const T* __restrict__ a, const float* __restrict__ pe,
        const float* __restrict__ alpha, float* __restrict__ out, int64_t M, int t, int d, float p,
        const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const float al = alpha[0];
    ROW_LOOP(M) {
        const int pos = (int)(row % t);
        float4 v[NG], pv[NG];
        row_load<NG, T>(a + row * d, d, lane, v);
        row_load<NG, float>(pe + (int64_t)pos * d, d, lane, pv);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g] = add4(v[g], scale4(pv[g], al));
            if (dc.on && GCOL(g) < d) v[g] = mul4(v[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
        }
        row_store<NG, float>(out + row * d, d, lane, v);
    }
}

template <typename T, int NG>
__global__ __launch_bounds__(RED_BLOCK) void pe_add_bwd_k(const float* __restrict__ dout, const float* __restrict__ pe,
        T* __restrict__ da, float* __restrict__ dalpha, int64_t M, int t, int d, float p, const uint64_t* rng,
        uint32_t site, float* __restrict__ dcolsum) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc = drop_ctx(rng, site, p);
    float acc = 0.f;
    float4 ac[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) ac[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    RED_LOOP(M) {
        const int pos = (int)(row % t);
        float4 v[NG], pv[NG];
        row_load<NG, float>(dout + row * d, d, lane, v);
        row_load<NG, float>(pe + (int64_t)pos * d, d, lane, pv);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (dc.on && GCOL(g) < d) v[g] = mul4(v[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
            acc += sum4(mul4(v[g], pv[g]));
            ac[g] = add4(ac[g], v[g]);
        }
        if (da != nullptr) row_store<NG, T>(da + row * d, d, lane, v);
    }
    // one atomic per BLOCK on the single dalpha word (8192 per-wave atomics on one address serialise: ~100 us)
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        atomicAdd(dalpha, t);
    }
    if (dcolsum != nullptr) flush_channel_sums<NG>(ac, dcolsum, d, red);
}

// ================================================================ head of an FFT stack: x = dropout(a + alpha pe[t]);  y = LN(x)
// (Models/modules.py:107-111 followed by the first layer's norm_1, Models/layers.py:31) in one row pass each way -- with ids != NULL
// the rows of `a` are gathered from the embedding table (nn.Embedding, Models/encoder.py:55,84): three launches of the encoder's
// forward become one.  x is stored in fp32 (the residual stream), y in the compute dtype, mean / rstd for the backward pass.
template <typename TA, typename TY, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void pe_add_ln_fwd_k(const TA* __restrict__ a, const int64_t* __restrict__ ids,
        const float* __restrict__ pe, const float* __restrict__ alpha, const float* __restrict__ gamma, const float* __restrict__ beta,
        float* __restrict__ x, TY* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t M, int t, int d, float eps,
        float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const float al = alpha[0];
    float4 gm[NG], bt[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
        row_load<NG, float>(beta, d, lane, bt);
    }
    ROW_LOOP(M) {
        const int pos = (int)(row % t);
        const int64_t src = ids != nullptr ? ids[row] : row;
        float4 v[NG], pv[NG];
        row_load<NG, TA>(a + src * d, d, lane, v);
        row_load<NG, float>(pe + (int64_t)pos * d, d, lane, pv);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g] = add4(v[g], scale4(pv[g], al));
            if (dc.on && GCOL(g) < d) v[g] = mul4(v[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
        }
        row_store<NG, float>(x + row * d, d, lane, v);
        float mu, rs;
        row_stats<NG>(v, d, lane, eps, mu, rs);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 o;
            o.x = (v[g].x - mu) * rs * gm[g].x + bt[g].x; o.y = (v[g].y - mu) * rs * gm[g].y + bt[g].y;
            o.z = (v[g].z - mu) * rs * gm[g].z + bt[g].z; o.w = (v[g].w - mu) * rs * gm[g].w + bt[g].w;
            v[g] = o;
        }
        row_store<NG, TY>(y + row * d, d, lane, v);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// backward of the same: LayerNorm backward of dy (+ the residual stream's gradient ds), then the positional encoder's: da = the
// gradient of `a` (dropout replayed), dalpha += sum da pe, dcolsum (optional) += column sums of da, dgamma / dbeta of the norm.
template <typename TDY, typename TDA, int NG>
__global__ __launch_bounds__(RED_BLOCK) void ln_pe_add_bwd_k(const TDY* __restrict__ dy, const float* __restrict__ x,
        const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ ds,
        const float* __restrict__ pe, TDA* __restrict__ da, float* __restrict__ dgamma, float* __restrict__ dbeta,
        float* __restrict__ dalpha, float* __restrict__ dcolsum, int64_t M, int t, int d, float p, const uint64_t* rng, uint32_t site) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 gm[NG], ag[NG], ab[NG], ac[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm);
#pragma unroll
        for (int g = 0; g < NG; ++g) ag[g] = ab[g] = ac[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invd = 1.f / (float)d;
    float acc = 0.f;
    RED_LOOP(M) {
        const int pos = (int)(row % t);
        float4 g_[NG], xv[NG], old[NG], pv[NG];
        row_load<NG, TDY>(dy + row * d, d, lane, g_);
        row_load<NG, float>(x + row * d, d, lane, xv);
        if (ds != nullptr) row_load<NG, float>(ds + row * d, d, lane, old);
        row_load<NG, float>(pe + (int64_t)pos * d, d, lane, pv);
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (GCOL(g) < d) {
                float4 xh = make_float4((xv[g].x - mu) * rs, (xv[g].y - mu) * rs, (xv[g].z - mu) * rs, (xv[g].w - mu) * rs);
                ag[g] = add4(ag[g], mul4(g_[g], xh));
                ab[g] = add4(ab[g], g_[g]);
                float4 dg = mul4(g_[g], gm[g]);
                c1 += sum4(dg);
                c2 += sum4(mul4(dg, xh));
                g_[g] = dg;   // dy * gamma
                xv[g] = xh;
            }
        }
        c1 = wave_sum(c1) * invd;
        c2 = wave_sum(c2) * invd;
        float4 o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            o[g].x = rs * (g_[g].x - c1 - xv[g].x * c2); o[g].y = rs * (g_[g].y - c1 - xv[g].y * c2);
            o[g].z = rs * (g_[g].z - c1 - xv[g].z * c2); o[g].w = rs * (g_[g].w - c1 - xv[g].w * c2);
            if (ds != nullptr) o[g] = add4(o[g], old[g]);
            if (GCOL(g) >= d) o[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dc.on && GCOL(g) < d) o[g] = mul4(o[g], drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
            acc += sum4(mul4(o[g], pv[g]));
            ac[g] = add4(ac[g], o[g]);
        }
        row_store<NG, TDA>(da + row * d, d, lane, o);
    }
    flush_channel_sums<NG>(ag, dgamma, d, red);
    flush_channel_sums<NG>(ab, dbeta, d, red);
    if (dcolsum != nullptr) flush_channel_sums<NG>(ac, dcolsum, d, red);
    // one atomic per BLOCK on the single dalpha word
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tsum = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tsum += red[w];
        atomicAdd(dalpha, tsum);
    }
}

// ================================================================ Linear(d -> 1) + masked_fill(0)
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void linear1_fwd_k(const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ b, const uint8_t* __restrict__ mask, float* __restrict__ out, int64_t M, int d) {
    float4 wv[NG];
    { const int lane = threadIdx.x & 63; row_load<NG, float>(w, d, lane, wv); }
    const float bias = b[0];
    ROW_LOOP(M) {
        float4 v[NG];
        row_load<NG, T>(x + row * d, d, lane, v);
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) s += sum4(mul4(v[g], wv[g]));
        s = wave_sum(s);
        if (lane == 0) out[row] = mask[row] ? s + bias : 0.f;
    }
}

template <typename T, int NG>
__global__ __launch_bounds__(RED_BLOCK) void linear1_bwd_k(const float* __restrict__ dout, const T* __restrict__ x,
        const float* __restrict__ w, const uint8_t* __restrict__ mask, T* __restrict__ dx, float* __restrict__ dw,
        float* __restrict__ db, int64_t M, int d) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    float4 wv[NG], aw[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(w, d, lane, wv);
#pragma unroll
        for (int g = 0; g < NG; ++g) aw[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ab = 0.f;
    RED_LOOP(M) {
        const float dout_row = dout[row];             // unconditional: both loads leave together
        const float go = mask[row] ? dout_row : 0.f;
        float4 v[NG], o[NG];
        row_load<NG, T>(x + row * d, d, lane, v);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            aw[g] = add4(aw[g], scale4(v[g], go));
            o[g] = scale4(wv[g], go);
        }
        row_store<NG, T>(dx + row * d, d, lane, o);
        if (lane == 0) ab += go;
    }
    flush_channel_sums<NG>(aw, dw, d, red);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ab;      // one atomic per block on the single db word
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        atomicAdd(db, t);
    }
}

// ================================================================ LayerNorm(+dropout) followed by Linear(d -> 1) + mask
// The tail of a VariancePredictor (Models/varianceadaptor.py:226-231): out = mask(Linear(dropout(LN(x)))).  = fs2_layernorm_fwd
// followed by fs2_linear1_fwd without storing the normalised rows (they are recomputed in the backward from x and the row
// statistics); rounded to T where the two-kernel form stores them.
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void ln_linear1_fwd_k(const T* __restrict__ x, const float* __restrict__ gamma,
        const float* __restrict__ beta, const float* __restrict__ w, const float* __restrict__ b, const uint8_t* __restrict__ mask,
        float* __restrict__ out, float* __restrict__ mean, float* __restrict__ rstd, int64_t M, int d, float eps, float p,
        const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 gm[NG], bt[NG], wv[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm); row_load<NG, float>(beta, d, lane, bt); row_load<NG, float>(w, d, lane, wv);
    }
    const float bias = b[0];
    ROW_LOOP(M) {
        float4 v[NG];
        row_load<NG, T>(x + row * d, d, lane, v);
        float mu, rs;
        row_stats<NG>(v, d, lane, eps, mu, rs);
        float sdot = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 o;
            o.x = (v[g].x - mu) * rs * gm[g].x + bt[g].x; o.y = (v[g].y - mu) * rs * gm[g].y + bt[g].y;
            o.z = (v[g].z - mu) * rs * gm[g].z + bt[g].z; o.w = (v[g].w - mu) * rs * gm[g].w + bt[g].w;
            if (dc.on && GCOL(g) < d) o = mul4(o, drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2));
            o = round_to<T>(o);
            if (GCOL(g) < d) sdot += sum4(mul4(o, wv[g]));
        }
        sdot = wave_sum(sdot);
        if (lane == 0) { out[row] = mask[row] ? sdot + bias : 0.f; mean[row] = mu; rstd[row] = rs; }
    }
}

// backward of the same: fs2_linear1_bwd followed by fs2_layernorm_bwd (relu_mask: the LayerNorm input was relu(z)) in one row pass
template <typename T, int NG>
__global__ __launch_bounds__(RED_BLOCK) void ln_linear1_bwd_k(const float* __restrict__ dout, const T* __restrict__ x,
        const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
        const float* __restrict__ w, const uint8_t* __restrict__ mask, T* __restrict__ dx, float* __restrict__ dgamma,
        float* __restrict__ dbeta, float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dcolsum, int64_t M, int d, float p,
        const uint64_t* rng, uint32_t site, int relu_mask) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 gm[NG], bt[NG], wv[NG], ag[NG], ab[NG], ac[NG], aw[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(gamma, d, lane, gm); row_load<NG, float>(beta, d, lane, bt); row_load<NG, float>(w, d, lane, wv);
#pragma unroll
        for (int g = 0; g < NG; ++g) ag[g] = ab[g] = ac[g] = aw[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invd = 1.f / (float)d;
    float abias = 0.f;
    RED_LOOP(M) {
        const float dout_row = dout[row];             // unconditional: both loads leave together
        const float go = mask[row] ? dout_row : 0.f;
        float4 xv[NG], g_[NG];
        row_load<NG, T>(x + row * d, d, lane, xv);
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
        unsigned pos[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            pos[g] = 0xFu;
            g_[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (GCOL(g) < d) {
                float4 ds = make_float4(1.f, 1.f, 1.f, 1.f);
                if (dc.on) ds = drop_scale4(dc, (uint64_t)(row * d + GCOL(g)) >> 2);
                if (relu_mask)
                    pos[g] = (xv[g].x > 0.f ? 1u : 0u) | (xv[g].y > 0.f ? 2u : 0u) | (xv[g].z > 0.f ? 4u : 0u) | (xv[g].w > 0.f ? 8u : 0u);
                float4 xh = make_float4((xv[g].x - mu) * rs, (xv[g].y - mu) * rs, (xv[g].z - mu) * rs, (xv[g].w - mu) * rs);
                // the forward's normalised row (what the Linear saw), recomputed: weight gradient of the Linear
                float4 n = make_float4(xh.x * gm[g].x + bt[g].x, xh.y * gm[g].y + bt[g].y, xh.z * gm[g].z + bt[g].z, xh.w * gm[g].w + bt[g].w);
                n = round_to<T>(mul4(n, ds));
                aw[g] = add4(aw[g], scale4(n, go));
                // gradient reaching the LayerNorm output: w * go (rounded as the two-kernel form stores it), through the dropout
                float4 gy = mul4(round_to<T>(scale4(wv[g], go)), ds);
                ag[g] = add4(ag[g], mul4(gy, xh));
                ab[g] = add4(ab[g], gy);
                float4 dg = mul4(gy, gm[g]);
                c1 += sum4(dg);
                c2 += sum4(mul4(dg, xh));
                g_[g] = dg;
                xv[g] = xh;
            }
        }
        c1 = wave_sum(c1) * invd;
        c2 = wave_sum(c2) * invd;
        float4 o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            o[g].x = rs * (g_[g].x - c1 - xv[g].x * c2); o[g].y = rs * (g_[g].y - c1 - xv[g].y * c2);
            o[g].z = rs * (g_[g].z - c1 - xv[g].z * c2); o[g].w = rs * (g_[g].w - c1 - xv[g].w * c2);
            if (relu_mask) {
                o[g].x = (pos[g] & 1u) ? o[g].x : 0.f; o[g].y = (pos[g] & 2u) ? o[g].y : 0.f;
                o[g].z = (pos[g] & 4u) ? o[g].z : 0.f; o[g].w = (pos[g] & 8u) ? o[g].w : 0.f;
            }
            ac[g] = add4(ac[g], o[g]);
        }
        row_store<NG, T>(dx + row * d, d, lane, o);
        if (lane == 0) abias += go;
    }
    flush_channel_sums<NG>(ag, dgamma, d, red);
    flush_channel_sums<NG>(ab, dbeta, d, red);
    flush_channel_sums<NG>(aw, dw, d, red);
    if (dcolsum != nullptr) flush_channel_sums<NG>(ac, dcolsum, d, red);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = abias;      // one atomic per block on the single db word
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int wv_ = 0; wv_ < (int)(blockDim.x >> 6); ++wv_) t += red[wv_];
        atomicAdd(db, t);
    }
}

// ================================================================ BatchNorm(batch stats) + tanh + dropout
template <typename T, int NG>
__global__ __launch_bounds__(RED_BLOCK) void colstats_k(const T* __restrict__ x, int64_t M, int C, float* __restrict__ sums) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    float4 a1[NG], a2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) a1[g] = a2[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    RED_LOOP(M) {
        float4 v[NG];
        row_load<NG, T>(x + row * C, C, lane, v);
#pragma unroll
        for (int g = 0; g < NG; ++g) { a1[g] = add4(a1[g], v[g]); a2[g] = add4(a2[g], mul4(v[g], v[g])); }
    }
    flush_channel_sums<NG>(a1, sums, C, red);
    flush_channel_sums<NG>(a2, sums + C, C, red);
}

__global__ void bn_finalize_k(const float* __restrict__ sums, float count, const float* __restrict__ count_dev, float eps, float momentum,
        float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
        float* __restrict__ running_var, int64_t* __restrict__ nbt, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (count_dev != nullptr) count = count_dev[0];
    if (c < C) {
        const double mud = (double)sums[c] / (double)count;
        const float mu = (float)mud;
        float var = (float)((double)sums[C + c] / (double)count - mud * mud);   // biased batch variance
        var = fmaxf(var, 0.f);
        mean[c] = mu;
        rstd[c] = 1.0f / sqrtf(var + eps);
        if (running_mean != nullptr) {
            const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
    if (c == 0 && nbt != nullptr) nbt[0] += 1;
}

// tanh(x) = sign(x) (1 - t) / (1 + t) with t = exp(-2|x|): one v_exp_f32 and one v_rcp_f32 instead of libm's tanhf (which made the
// BatchNorm+tanh kernels ALU-bound); odd polynomial below |x| = 0.04 where 1 - t cancels.  |error| <= 2e-7 absolute, <= 3e-7 relative.
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_exp2f(ax * -2.8853900817779268f);
    const float big = (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
    const float x2 = x * x;
    const float small = ax * __builtin_fmaf(x2, __builtin_fmaf(x2, 0.13333333f, -0.33333334f), 1.f);
    return copysignf(ax < 0.04f ? small : big, x);
}

template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void bn_tanh_fwd_k(const T* __restrict__ x, const float* __restrict__ mean,
        const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
        T* __restrict__ y, int64_t M, int C, float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 sc[NG], sh[NG];
    {
        const int lane = threadIdx.x & 63;
        float4 mu[NG], rs[NG], gm[NG], bt[NG];
        row_load<NG, float>(mean, C, lane, mu); row_load<NG, float>(rstd, C, lane, rs);
        row_load<NG, float>(gamma, C, lane, gm); row_load<NG, float>(beta, C, lane, bt);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            sc[g] = mul4(rs[g], gm[g]);
            sh[g] = make_float4(bt[g].x - mu[g].x * sc[g].x, bt[g].y - mu[g].y * sc[g].y,
                                bt[g].z - mu[g].z * sc[g].z, bt[g].w - mu[g].w * sc[g].w);
        }
    }
    ROW_LOOP(M) {
        float4 v[NG];
        row_load<NG, T>(x + row * C, C, lane, v);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g].x = tanh_fast(v[g].x * sc[g].x + sh[g].x); v[g].y = tanh_fast(v[g].y * sc[g].y + sh[g].y);
            v[g].z = tanh_fast(v[g].z * sc[g].z + sh[g].z); v[g].w = tanh_fast(v[g].w * sc[g].w + sh[g].w);
            if (dc.on && GCOL(g) < C) v[g] = mul4(v[g], drop_scale4(dc, (uint64_t)(row * C + GCOL(g)) >> 2));
        }
        row_store<NG, T>(y + row * C, C, lane, v);
    }
}

// bn_finalize_k + bn_tanh_fwd_k in one launch (the training forward of a post-net layer: four launches less per step): every wave
// derives mean / rstd of its channels from the column sums the producing GEMM's epilogue left (the same arithmetic as bn_finalize_k),
// wave 0 of block 0 stores them for the backward pass and updates the running statistics.
template <typename T, int NG>
__global__ __launch_bounds__(ROW_BLOCK) void bn_stats_tanh_fwd_k(const T* __restrict__ x, const float* __restrict__ sums, float count,
        const float* __restrict__ count_dev, float eps, float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
        T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
        float* __restrict__ running_var, int64_t* __restrict__ nbt, int64_t M, int C, float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 sc[NG], sh[NG];
    {
        const int lane = threadIdx.x & 63;
        if (count_dev != nullptr) count = count_dev[0];
        float4 s1[NG], s2[NG], gm[NG], bt[NG], mu[NG], rs[NG], vr[NG];
        row_load<NG, float>(sums, C, lane, s1); row_load<NG, float>(sums + C, C, lane, s2);
        row_load<NG, float>(gamma, C, lane, gm); row_load<NG, float>(beta, C, lane, bt);
        const double icount = 1.0 / (double)count;       // (one division per lane; bn_finalize_k divides per channel: equal to ~1e-16 relative)
        auto fin = [&](float a, float b, float& m, float& r, float& v) {
            const double mud = (double)a * icount;
            m = (float)mud;
            v = fmaxf((float)((double)b * icount - mud * mud), 0.f);      // biased batch variance
            r = __builtin_amdgcn_rsqf(v + eps);
            r = r * (1.5f - 0.5f * (v + eps) * r * r);                    // one Newton step: rsq to full fp32 precision
        };
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            fin(s1[g].x, s2[g].x, mu[g].x, rs[g].x, vr[g].x); fin(s1[g].y, s2[g].y, mu[g].y, rs[g].y, vr[g].y);
            fin(s1[g].z, s2[g].z, mu[g].z, rs[g].z, vr[g].z); fin(s1[g].w, s2[g].w, mu[g].w, rs[g].w, vr[g].w);
            sc[g] = mul4(rs[g], gm[g]);
            sh[g] = make_float4(bt[g].x - mu[g].x * sc[g].x, bt[g].y - mu[g].y * sc[g].y,
                                bt[g].z - mu[g].z * sc[g].z, bt[g].w - mu[g].w * sc[g].w);
        }
        if (blockIdx.x == 0 && threadIdx.x < 64) {
            row_store<NG, float>(mean, C, lane, mu);
            row_store<NG, float>(rstd, C, lane, rs);
            if (running_mean != nullptr) {
                float4 rm[NG], rv[NG];
                row_load<NG, float>(running_mean, C, lane, rm); row_load<NG, float>(running_var, C, lane, rv);
                const float ub = count > 1.f ? count / (count - 1.f) : 1.f;
                auto upd = [&](float old, float now) { return (1.f - momentum) * old + momentum * now; };
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    rm[g] = make_float4(upd(rm[g].x, mu[g].x), upd(rm[g].y, mu[g].y), upd(rm[g].z, mu[g].z), upd(rm[g].w, mu[g].w));
                    rv[g] = make_float4(upd(rv[g].x, vr[g].x * ub), upd(rv[g].y, vr[g].y * ub), upd(rv[g].z, vr[g].z * ub), upd(rv[g].w, vr[g].w * ub));
                }
                row_store<NG, float>(running_mean, C, lane, rm);
                row_store<NG, float>(running_var, C, lane, rv);
            }
            if (threadIdx.x == 0 && nbt != nullptr) nbt[0] += 1;
        }
    }
    ROW_LOOP(M) {
        float4 v[NG];
        row_load<NG, T>(x + row * C, C, lane, v);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v[g].x = tanh_fast(v[g].x * sc[g].x + sh[g].x); v[g].y = tanh_fast(v[g].y * sc[g].y + sh[g].y);
            v[g].z = tanh_fast(v[g].z * sc[g].z + sh[g].z); v[g].w = tanh_fast(v[g].w * sc[g].w + sh[g].w);
            if (dc.on && GCOL(g) < C) v[g] = mul4(v[g], drop_scale4(dc, (uint64_t)(row * C + GCOL(g)) >> 2));
        }
        row_store<NG, T>(y + row * C, C, lane, v);
    }
}

// MODE 0: accumulate red[0..C) += sum dz, red[C..2C) += sum dz*xhat.   MODE 1: write dx.
template <typename T, int NG, int MODE>
__global__ __launch_bounds__(RED_BLOCK) void bn_tanh_bwd_k(const T* __restrict__ dy, const T* __restrict__ x,
        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
        const float* __restrict__ beta, float* __restrict__ red_io, float count, const float* __restrict__ count_dev, T* __restrict__ dx,
        float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t M, int C, float p, const uint64_t* rng,
        uint32_t site, float* __restrict__ dcolsum) {
    __shared__ __attribute__((aligned(16))) float red[RED_WAVES * NG * 256];
    const DropCtx dc = drop_ctx(rng, site, p);
    float4 mu[NG], rs[NG], gm[NG], bt[NG], a1[NG], a2[NG], r0[NG], r1[NG];
    {
        const int lane = threadIdx.x & 63;
        row_load<NG, float>(mean, C, lane, mu); row_load<NG, float>(rstd, C, lane, rs);
        row_load<NG, float>(gamma, C, lane, gm); row_load<NG, float>(beta, C, lane, bt);
#pragma unroll
        for (int g = 0; g < NG; ++g) a1[g] = a2[g] = r0[g] = r1[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1) {
            row_load<NG, float>(red_io, C, lane, r0);
            row_load<NG, float>(red_io + C, C, lane, r1);
            const float ic = 1.f / (count_dev != nullptr ? count_dev[0] : count);
#pragma unroll
            for (int g = 0; g < NG; ++g) { r0[g] = scale4(r0[g], ic); r1[g] = scale4(r1[g], ic); }
        }
    }
    RED_LOOP(M) {
        float4 g_[NG], xv[NG];
        row_load<NG, T>(dy + row * C, C, lane, g_);
        row_load<NG, T>(x + row * C, C, lane, xv);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (dc.on && GCOL(g) < C) g_[g] = mul4(g_[g], drop_scale4(dc, (uint64_t)(row * C + GCOL(g)) >> 2));
            float4 xh = make_float4((xv[g].x - mu[g].x) * rs[g].x, (xv[g].y - mu[g].y) * rs[g].y,
                                    (xv[g].z - mu[g].z) * rs[g].z, (xv[g].w - mu[g].w) * rs[g].w);
            float4 th = make_float4(tanh_fast(xh.x * gm[g].x + bt[g].x), tanh_fast(xh.y * gm[g].y + bt[g].y),
                                    tanh_fast(xh.z * gm[g].z + bt[g].z), tanh_fast(xh.w * gm[g].w + bt[g].w));
            float4 dz = make_float4(g_[g].x * (1.f - th.x * th.x), g_[g].y * (1.f - th.y * th.y),
                                    g_[g].z * (1.f - th.z * th.z), g_[g].w * (1.f - th.w * th.w));
            if (MODE == 0) {
                a1[g] = add4(a1[g], dz);
                a2[g] = add4(a2[g], mul4(dz, xh));
            } else {
                g_[g].x = gm[g].x * rs[g].x * (dz.x - r0[g].x - xh.x * r1[g].x);
                g_[g].y = gm[g].y * rs[g].y * (dz.y - r0[g].y - xh.y * r1[g].y);
                g_[g].z = gm[g].z * rs[g].z * (dz.z - r0[g].z - xh.z * r1[g].z);
                g_[g].w = gm[g].w * rs[g].w * (dz.w - r0[g].w - xh.w * r1[g].w);
                if (GCOL(g) < C) a1[g] = add4(a1[g], g_[g]);      // column sums of dx (bias gradient of the producing conv)
            }
        }
        if (MODE == 1) row_store<NG, T>(dx + row * C, C, lane, g_);
    }
    if (MODE == 0) {
        flush_channel_sums<NG>(a1, red_io, C, red);
        flush_channel_sums<NG>(a2, red_io + C, C, red);
    } else {
        if (dcolsum != nullptr) flush_channel_sums<NG>(a1, dcolsum, C, red);
        if (blockIdx.x == 0 && dgamma != nullptr) {
            // dbeta += sum dz, dgamma += sum dz*xhat (already reduced over rows -- and over ranks -- in red_io)
            for (int c = threadIdx.x; c < C; c += blockDim.x) {
                atomicAdd(dbeta + c, red_io[c]);
                atomicAdd(dgamma + c, red_io[C + c]);
            }
        }
    }
}

}  // namespace

// ================================================================ host launchers
#define NG_DISPATCH(d, NGV, ...)                                             \
    do {                                                                     \
        if ((d) <= 256) { constexpr int NGV = 1; __VA_ARGS__; }              \
        else if ((d) <= 512) { constexpr int NGV = 2; __VA_ARGS__; }         \
        else if ((d) <= 1024) { constexpr int NGV = 4; __VA_ARGS__; }        \
        else { constexpr int NGV = 8; __VA_ARGS__; }                         \
    } while (0)

#define CHECK_ROW(name, d, maxd)                                                                     \
    FS2_REQUIRE((d) > 0 && (d) % 4 == 0 && (d) <= (maxd), "%s: row length %d must be a multiple of 4 and <= %d", name, (int)(d), (int)(maxd))
#define CHECK_DT(name, dt) FS2_REQUIRE((dt) == FS2_F32 || (dt) == FS2_BF16, "%s: bad dtype %d", name, (int)(dt))

// ---- one-shot request (per host thread): the next fs2_add_ln_fwd / fs2_add_ln_bwd / fs2_ffn_tail_fwd / fs2_ffn_tail_bwd launch also writes
//      the fp8 copy of its bf16 row output (y / da / y / g)
namespace {
thread_local Q8Arg g_q8_next = {nullptr, nullptr, nullptr, 0};
Q8Arg take_q8(int dtype) {
    Q8Arg a = g_q8_next;
    g_q8_next = Q8Arg{nullptr, nullptr, nullptr, 0};
    if (dtype != FS2_BF16) a.q = nullptr;
    return a;
}
}  // namespace
extern "C" int fs2_q8_next(void* q8, float* state, const float* prev, int bf8) {
    FS2_REQUIRE(q8 != nullptr && state != nullptr && prev != nullptr, "fs2_q8_next: null argument");
    FS2_REQUIRE((reinterpret_cast<uintptr_t>(q8) & 3u) == 0, "fs2_q8_next: q8 must be 4-byte aligned");
    g_q8_next = Q8Arg{reinterpret_cast<unsigned char*>(q8), state, prev, bf8 ? 1 : 0};
    return FS2_OK;
}

extern "C" int fs2_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                 float* mean, float* rstd, int64_t M, int d, float eps, float p, const uint64_t* rng,
                                 uint32_t site, void* stream) {
    CHECK_ROW("fs2_layernorm_fwd", d, 2048); CHECK_DT("fs2_layernorm_fwd", x_dtype); CHECK_DT("fs2_layernorm_fwd", y_dtype);
    FS2_REQUIRE(!(x_dtype == FS2_BF16 && y_dtype == FS2_F32), "fs2_layernorm_fwd: bf16 -> f32 is not provided");
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_layernorm_fwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, {
        if (x_dtype == FS2_F32 && y_dtype == FS2_F32)
            hipLaunchKernelGGL((layernorm_fwd_k<float, float, NG>), grid, block, 0, st, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, d, eps, p, rng, site);
        else if (x_dtype == FS2_F32)
            hipLaunchKernelGGL((layernorm_fwd_k<float, bf16_t, NG>), grid, block, 0, st, (const float*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, d, eps, p, rng, site);
        else
            hipLaunchKernelGGL((layernorm_fwd_k<bf16_t, bf16_t, NG>), grid, block, 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, d, eps, p, rng, site);
    });
    FS2_CHECK_LAUNCH("fs2_layernorm_fwd");
    return FS2_OK;
}

extern "C" int fs2_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                                 const float* mean, const float* rstd, void* dx, int dx_dtype, float* dgamma,
                                 float* dbeta, int64_t M, int d, float p, const uint64_t* rng, uint32_t site,
                                 int relu_mask, int dx_accumulate, float* dcolsum, void* stream) {
    CHECK_ROW("fs2_layernorm_bwd", d, 1024);
    FS2_REQUIRE(x_dtype == dx_dtype, "fs2_layernorm_bwd: dx dtype must equal x dtype");
    FS2_REQUIRE(!(dy_dtype == FS2_F32 && x_dtype == FS2_BF16), "fs2_layernorm_bwd: f32 dy with bf16 x is not provided");
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_layernorm_bwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, {
        if constexpr (NG <= 4) {
            if (dy_dtype == FS2_F32)
                hipLaunchKernelGGL((layernorm_bwd_k<float, float, float, NG>), grid, block, 0, st, (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, dgamma, dbeta, M, d, p, rng, site, relu_mask, dx_accumulate, dcolsum);
            else if (x_dtype == FS2_F32)
                hipLaunchKernelGGL((layernorm_bwd_k<bf16_t, float, float, NG>), grid, block, 0, st, (const bf16_t*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, dgamma, dbeta, M, d, p, rng, site, relu_mask, dx_accumulate, dcolsum);
            else
                hipLaunchKernelGGL((layernorm_bwd_k<bf16_t, bf16_t, bf16_t, NG>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)dx, dgamma, dbeta, M, d, p, rng, site, relu_mask, dx_accumulate, dcolsum);
        }
    });
    FS2_CHECK_LAUNCH("fs2_layernorm_bwd");
    return FS2_OK;
}

#define T_DISPATCH(dtype, T, ...)                                     \
    do {                                                              \
        if ((dtype) == FS2_F32) { typedef float T; __VA_ARGS__; }     \
        else { typedef bf16_t T; __VA_ARGS__; }                       \
    } while (0)

extern "C" int fs2_add_ln_fwd(const float* r, const void* a, int dtype, float* s, const float* gamma, const float* beta,
                              void* y, float* mean, float* rstd, int64_t M, int d, float eps, float p,
                              const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_ROW("fs2_add_ln_fwd", d, 1024); CHECK_DT("fs2_add_ln_fwd", dtype);
    const Q8Arg qa = take_q8(dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_add_ln_fwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((add_ln_fwd_k<T, NG>), grid, block, 0, st, r, (const T*)a, s, gamma, beta, (T*)y, mean, rstd, M, d, eps, p, rng, site, qa);
    }); } });
    FS2_CHECK_LAUNCH("fs2_add_ln_fwd");
    return FS2_OK;
}

extern "C" int fs2_add_ln_bwd(const float* ds_down, const void* dy, int dtype, const float* s, const float* gamma,
                              const float* mean, const float* rstd, float* dr, void* da, float* dgamma, float* dbeta,
                              int64_t M, int d, float p, const uint64_t* rng, uint32_t site, float* dcolsum, void* stream) {
    CHECK_ROW("fs2_add_ln_bwd", d, 1024); CHECK_DT("fs2_add_ln_bwd", dtype);
    const Q8Arg qa = take_q8(dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_add_ln_bwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((add_ln_bwd_k<T, NG>), grid, block, 0, st, ds_down, (const T*)dy, s, gamma, mean, rstd, dr, (T*)da, dgamma, dbeta, M, d, p, rng, site, dcolsum, qa);
    }); } });
    FS2_CHECK_LAUNCH("fs2_add_ln_bwd");
    return FS2_OK;
}

extern "C" int fs2_ffn_ln_fwd(const void* f2, const void* h, int dtype, const float* gamma, const float* beta, void* y,
                              float* mean, float* rstd, int64_t M, int d, float eps, float p, const uint64_t* rng,
                              uint32_t site, void* stream) {
    CHECK_ROW("fs2_ffn_ln_fwd", d, 1024); CHECK_DT("fs2_ffn_ln_fwd", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_ffn_ln_fwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((ffn_ln_fwd_k<T, NG>), grid, block, 0, st, (const T*)f2, (const T*)h, gamma, beta, (T*)y, mean, rstd, M, d, eps, p, rng, site);
    }); } });
    FS2_CHECK_LAUNCH("fs2_ffn_ln_fwd");
    return FS2_OK;
}

extern "C" int fs2_ffn_ln_bwd(const void* dy, const void* f2, const void* h, int dtype, const float* gamma,
                              const float* mean, const float* rstd, void* g, float* dgamma, float* dbeta, int64_t M,
                              int d, float p, const uint64_t* rng, uint32_t site, float* dcolsum, void* stream) {
    CHECK_ROW("fs2_ffn_ln_bwd", d, 1024); CHECK_DT("fs2_ffn_ln_bwd", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_ffn_ln_bwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((ffn_ln_bwd_k<T, NG>), grid, block, 0, st, (const T*)dy, (const T*)f2, (const T*)h, gamma, mean, rstd, (T*)g, dgamma, dbeta, M, d, p, rng, site, dcolsum);
    }); } });
    FS2_CHECK_LAUNCH("fs2_ffn_ln_bwd");
    return FS2_OK;
}

extern "C" int fs2_ffn_tail_fwd(const void* f2, const void* h, int dtype, const float* r, const float* gamma1, const float* beta1,
                                const float* gamma2, const float* beta2, float* s, void* y, float* mean1, float* rstd1, float* mean2,
                                float* rstd2, int64_t M, int d, float eps, float p, const uint64_t* rng, uint32_t site1, uint32_t site2,
                                void* stream) {
    CHECK_ROW("fs2_ffn_tail_fwd", d, 1024); CHECK_DT("fs2_ffn_tail_fwd", dtype);
    const Q8Arg qa = take_q8(dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_ffn_tail_fwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((ffn_tail_fwd_k<T, NG>), grid, block, 0, st, (const T*)f2, (const T*)h, r, gamma1, beta1, gamma2, beta2, s, (T*)y, mean1, rstd1, mean2, rstd2, M, d, eps, p, rng, site1, site2, qa);
    }); } });
    FS2_CHECK_LAUNCH("fs2_ffn_tail_fwd");
    return FS2_OK;
}

extern "C" int fs2_ffn_tail_bwd(const float* ds_down, const void* dy, int dtype, const float* s, const float* gamma2, const float* mean2,
                                const float* rstd2, const void* f2, const void* h, const float* gamma1, const float* mean1,
                                const float* rstd1, float* dr, void* g, float* dgamma2, float* dbeta2, float* dgamma1, float* dbeta1,
                                float* dcolsum, int64_t M, int d, float p, const uint64_t* rng, uint32_t site1, uint32_t site2,
                                void* stream) {
    CHECK_ROW("fs2_ffn_tail_bwd", d, 1024); CHECK_DT("fs2_ffn_tail_bwd", dtype);
    const Q8Arg qa = take_q8(dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_ffn_tail_bwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((ffn_tail_bwd_k<T, NG>), grid, block, 0, st, ds_down, (const T*)dy, s, gamma2, mean2, rstd2, (const T*)f2, (const T*)h, gamma1, mean1, rstd1, dr, (T*)g, dgamma2, dbeta2, dgamma1, dbeta1, dcolsum, M, d, p, rng, site1, site2, qa);
    }); } });
    FS2_CHECK_LAUNCH("fs2_ffn_tail_bwd");
    return FS2_OK;
}

extern "C" int fs2_softmax_fwd(void* s, void* pd, int dtype, const uint8_t* key_mask, int B, int H, int t, int tp,
                               int64_t batch_stride, float p, const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_DT("fs2_softmax_fwd", dtype);
    FS2_REQUIRE(t > 0 && tp >= t && tp % 8 == 0 && tp <= 2048, "fs2_softmax_fwd: need 0 < t <= tp <= 2048, tp %% 8 == 0 (t=%d tp=%d)", t, tp);
    FS2_REQUIRE(batch_stride % 4 == 0, "fs2_softmax_fwd: batch_stride must be a multiple of 4");
    FS2_REQUIRE(p == 0.f || (rng != nullptr && pd != s), "fs2_softmax_fwd: dropout needs rng and a separate p_drop buffer");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid((int64_t)B * H * t)), block(ROW_BLOCK);
    if (dtype == FS2_BF16 && batch_stride % 8 == 0 && fs2_aligned16(s) && fs2_aligned16(pd) &&
        (reinterpret_cast<uintptr_t>(key_mask) & 3) == 0) {
        constexpr int R = 2;
        dim3 grid8(row_grid(((int64_t)B * H * t + R - 1) / R));
#define SM8(NG8) hipLaunchKernelGGL((softmax_fwd8_k<NG8, R>), grid8, block, 0, st, (bf16_t*)s, (bf16_t*)pd, key_mask, B, H, t, tp, batch_stride, p, rng, site)
        if (tp <= 512) SM8(1); else if (tp <= 1024) SM8(2); else if (tp <= 1536) SM8(3); else SM8(4);
#undef SM8
        FS2_CHECK_LAUNCH("fs2_softmax_fwd");
        return FS2_OK;
    }
    NG_DISPATCH(tp, NG, { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((softmax_fwd_k<T, NG>), grid, block, 0, st, (T*)s, (T*)pd, key_mask, B, H, t, t, tp, batch_stride, 0, p, rng, site);
    }); });
    FS2_CHECK_LAUNCH("fs2_softmax_fwd");
    return FS2_OK;
}

extern "C" int fs2_softmax_bwd(void* dp, int64_t dp_batch_stride, const void* ps, int64_t p_batch_stride, int dtype, int B,
                               int H, int t, int tp, float p, const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_DT("fs2_softmax_bwd", dtype);
    FS2_REQUIRE(t > 0 && tp >= t && tp % 8 == 0 && tp <= 2048, "fs2_softmax_bwd: need 0 < t <= tp <= 2048, tp %% 8 == 0");
    FS2_REQUIRE(dp_batch_stride % 4 == 0 && p_batch_stride % 4 == 0, "fs2_softmax_bwd: batch strides must be multiples of 4");
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_softmax_bwd: dropout needs rng");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid((int64_t)B * H * t)), block(ROW_BLOCK);
    if (dtype == FS2_BF16 && dp_batch_stride % 8 == 0 && p_batch_stride % 8 == 0 && fs2_aligned16(dp) && fs2_aligned16(ps)) {
        constexpr int R = 2;
        dim3 grid8(row_grid(((int64_t)B * H * t + R - 1) / R));
#define SM8(NG8) hipLaunchKernelGGL((softmax_bwd8_k<NG8, R>), grid8, block, 0, st, (bf16_t*)dp, dp_batch_stride, (const bf16_t*)ps, p_batch_stride, B, H, t, tp, p, rng, site)
        if (tp <= 512) SM8(1); else if (tp <= 1024) SM8(2); else if (tp <= 1536) SM8(3); else SM8(4);
#undef SM8
        FS2_CHECK_LAUNCH("fs2_softmax_bwd");
        return FS2_OK;
    }
    NG_DISPATCH(tp, NG, { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((softmax_bwd_k<T, NG>), grid, block, 0, st, (T*)dp, dp_batch_stride, (const T*)ps, p_batch_stride, B, H, t, t, tp, p, rng, site);
    }); });
    FS2_CHECK_LAUNCH("fs2_softmax_bwd");
    return FS2_OK;
}

// rectangular / causal form: tq query rows per head, tk keys (row stride tkp); the decoder self-attention (causal) and the
// encoder-decoder attention (tq != tk) of the autoregressive model
extern "C" int fs2_softmax_rect_fwd(void* s, void* pd, int dtype, const uint8_t* key_mask, int B, int H, int tq, int tk, int tkp,
                                    int64_t batch_stride, int causal, float p, const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_DT("fs2_softmax_rect_fwd", dtype);
    FS2_REQUIRE(tq > 0 && tk > 0 && tkp >= tk && tkp % 8 == 0 && tkp <= 2048, "fs2_softmax_rect_fwd: need 0 < tk <= tkp <= 2048, tkp %% 8 == 0 (tk=%d tkp=%d)", tk, tkp);
    FS2_REQUIRE(!causal || tq == tk, "fs2_softmax_rect_fwd: the causal mask needs tq == tk");
    FS2_REQUIRE(batch_stride % 4 == 0, "fs2_softmax_rect_fwd: batch_stride must be a multiple of 4");
    FS2_REQUIRE(p == 0.f || (rng != nullptr && pd != s), "fs2_softmax_rect_fwd: dropout needs rng and a separate p_drop buffer");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid((int64_t)B * H * tq)), block(ROW_BLOCK);
    NG_DISPATCH(tkp, NG, { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((softmax_fwd_k<T, NG>), grid, block, 0, st, (T*)s, (T*)pd, key_mask, B, H, tq, tk, tkp, batch_stride, causal, p, rng, site);
    }); });
    FS2_CHECK_LAUNCH("fs2_softmax_rect_fwd");
    return FS2_OK;
}

extern "C" int fs2_softmax_rect_bwd(void* dp, int64_t dp_batch_stride, const void* ps, int64_t p_batch_stride, int dtype, int B,
                                    int H, int tq, int tk, int tkp, float p, const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_DT("fs2_softmax_rect_bwd", dtype);
    FS2_REQUIRE(tq > 0 && tk > 0 && tkp >= tk && tkp % 8 == 0 && tkp <= 2048, "fs2_softmax_rect_bwd: need 0 < tk <= tkp <= 2048, tkp %% 8 == 0");
    FS2_REQUIRE(dp_batch_stride % 4 == 0 && p_batch_stride % 4 == 0, "fs2_softmax_rect_bwd: batch strides must be multiples of 4");
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_softmax_rect_bwd: dropout needs rng");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid((int64_t)B * H * tq)), block(ROW_BLOCK);
    NG_DISPATCH(tkp, NG, { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((softmax_bwd_k<T, NG>), grid, block, 0, st, (T*)dp, dp_batch_stride, (const T*)ps, p_batch_stride, B, H, tq, tk, tkp, p, rng, site);
    }); });
    FS2_CHECK_LAUNCH("fs2_softmax_rect_bwd");
    return FS2_OK;
}

extern "C" int fs2_pe_add_fwd(const void* a, int a_dtype, const float* pe, const float* alpha, float* out, int B, int t,
                              int d, float p, const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_ROW("fs2_pe_add_fwd", d, 1024); CHECK_DT("fs2_pe_add_fwd", a_dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_pe_add_fwd: dropout needs rng");
    const int64_t M = (int64_t)B * t;
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(a_dtype, T, {
        hipLaunchKernelGGL((pe_add_fwd_k<T, NG>), grid, block, 0, st, (const T*)a, pe, alpha, out, M, t, d, p, rng, site);
    }); } });
    FS2_CHECK_LAUNCH("fs2_pe_add_fwd");
    return FS2_OK;
}

extern "C" int fs2_pe_add_ln_fwd(const void* a, int a_dtype, const int64_t* ids, const float* pe, const float* alpha, const float* gamma,
                                 const float* beta, float* x, void* y, int y_dtype, float* mean, float* rstd, int B, int t, int d,
                                 float eps, float p, const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_ROW("fs2_pe_add_ln_fwd", d, 1024); CHECK_DT("fs2_pe_add_ln_fwd", a_dtype); CHECK_DT("fs2_pe_add_ln_fwd", y_dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_pe_add_ln_fwd: dropout needs rng");
    FS2_REQUIRE(a && pe && alpha && gamma && beta && x && y && mean && rstd, "fs2_pe_add_ln_fwd: null operand");
    FS2_REQUIRE(ids == nullptr || a_dtype == FS2_F32, "fs2_pe_add_ln_fwd: the embedding table is fp32");
    const int64_t M = (int64_t)B * t;
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(a_dtype, TA, { T_DISPATCH(y_dtype, TY, {
        hipLaunchKernelGGL((pe_add_ln_fwd_k<TA, TY, NG>), grid, block, 0, st, (const TA*)a, ids, pe, alpha, gamma, beta, x, (TY*)y, mean, rstd,
                           M, t, d, eps, p, rng, site);
    }); }); } });
    FS2_CHECK_LAUNCH("fs2_pe_add_ln_fwd");
    return FS2_OK;
}

extern "C" int fs2_ln_pe_add_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 const float* ds, const float* pe, void* da, int da_dtype, float* dgamma, float* dbeta, float* dalpha,
                                 float* dcolsum, int B, int t, int d, float p, const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_ROW("fs2_ln_pe_add_bwd", d, 1024); CHECK_DT("fs2_ln_pe_add_bwd", dy_dtype); CHECK_DT("fs2_ln_pe_add_bwd", da_dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_ln_pe_add_bwd: dropout needs rng");
    FS2_REQUIRE(dy && x && gamma && mean && rstd && pe && da && dgamma && dbeta && dalpha, "fs2_ln_pe_add_bwd: null operand");
    const int64_t M = (int64_t)B * t;
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dy_dtype, TDY, { T_DISPATCH(da_dtype, TDA, {
        hipLaunchKernelGGL((ln_pe_add_bwd_k<TDY, TDA, NG>), grid, block, 0, st, (const TDY*)dy, x, gamma, mean, rstd, ds, pe, (TDA*)da, dgamma,
                           dbeta, dalpha, dcolsum, M, t, d, p, rng, site);
    }); }); } });
    FS2_CHECK_LAUNCH("fs2_ln_pe_add_bwd");
    return FS2_OK;
}

extern "C" int fs2_pe_add_bwd(const float* dout, const float* pe, void* da, int da_dtype, float* dalpha, int B, int t,
                              int d, float p, const uint64_t* rng, uint32_t site, float* dcolsum, void* stream) {
    CHECK_ROW("fs2_pe_add_bwd", d, 1024); CHECK_DT("fs2_pe_add_bwd", da_dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_pe_add_bwd: dropout needs rng");
    const int64_t M = (int64_t)B * t;
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(da_dtype, T, {
        hipLaunchKernelGGL((pe_add_bwd_k<T, NG>), grid, block, 0, st, dout, pe, (T*)da, dalpha, M, t, d, p, rng, site, dcolsum);
    }); } });
    FS2_CHECK_LAUNCH("fs2_pe_add_bwd");
    return FS2_OK;
}

extern "C" int fs2_linear1_fwd(const void* x, int dtype, const float* w, const float* b, const uint8_t* mask, float* out,
                               int64_t M, int d, void* stream) {
    CHECK_ROW("fs2_linear1_fwd", d, 1024); CHECK_DT("fs2_linear1_fwd", dtype);
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((linear1_fwd_k<T, NG>), grid, block, 0, st, (const T*)x, w, b, mask, out, M, d);
    }); } });
    FS2_CHECK_LAUNCH("fs2_linear1_fwd");
    return FS2_OK;
}

extern "C" int fs2_linear1_bwd(const float* dout, const void* x, int dtype, const float* w, const uint8_t* mask, void* dx,
                               float* dw, float* db, int64_t M, int d, void* stream) {
    CHECK_ROW("fs2_linear1_bwd", d, 1024); CHECK_DT("fs2_linear1_bwd", dtype);
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((linear1_bwd_k<T, NG>), grid, block, 0, st, dout, (const T*)x, w, mask, (T*)dx, dw, db, M, d);
    }); } });
    FS2_CHECK_LAUNCH("fs2_linear1_bwd");
    return FS2_OK;
}

extern "C" int fs2_ln_linear1_fwd(const void* x, int dtype, const float* gamma, const float* beta, const float* w, const float* b,
                                  const uint8_t* mask, float* out, float* mean, float* rstd, int64_t M, int d, float eps, float p,
                                  const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_ROW("fs2_ln_linear1_fwd", d, 1024); CHECK_DT("fs2_ln_linear1_fwd", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_ln_linear1_fwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((ln_linear1_fwd_k<T, NG>), grid, block, 0, st, (const T*)x, gamma, beta, w, b, mask, out, mean, rstd, M, d, eps, p, rng, site);
    }); } });
    FS2_CHECK_LAUNCH("fs2_ln_linear1_fwd");
    return FS2_OK;
}

extern "C" int fs2_ln_linear1_bwd(const float* dout, const void* x, int dtype, const float* gamma, const float* beta, const float* mean,
                                  const float* rstd, const float* w, const uint8_t* mask, void* dx, float* dgamma, float* dbeta, float* dw,
                                  float* db, float* dcolsum, int64_t M, int d, float p, const uint64_t* rng, uint32_t site, int relu_mask,
                                  void* stream) {
    CHECK_ROW("fs2_ln_linear1_bwd", d, 1024); CHECK_DT("fs2_ln_linear1_bwd", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_ln_linear1_bwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(d, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((ln_linear1_bwd_k<T, NG>), grid, block, 0, st, dout, (const T*)x, gamma, beta, mean, rstd, w, mask, (T*)dx, dgamma, dbeta, dw, db, dcolsum, M, d, p, rng, site, relu_mask);
    }); } });
    FS2_CHECK_LAUNCH("fs2_ln_linear1_bwd");
    return FS2_OK;
}

extern "C" int fs2_colstats(const void* x, int dtype, int64_t M, int C, float* sums, void* stream) {
    CHECK_ROW("fs2_colstats", C, 1024); CHECK_DT("fs2_colstats", dtype);
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(C, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((colstats_k<T, NG>), grid, block, 0, st, (const T*)x, M, C, sums);
    }); } });
    FS2_CHECK_LAUNCH("fs2_colstats");
    return FS2_OK;
}

extern "C" int fs2_bn_finalize(const float* sums, float count, const float* count_dev, float eps, float momentum,
                               float* mean, float* rstd, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, int C, void* stream) {
    FS2_REQUIRE(C > 0 && (count > 0.f || count_dev != nullptr), "fs2_bn_finalize: bad C/count");
    hipLaunchKernelGGL(bn_finalize_k, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count, count_dev, eps,
                       momentum, mean, rstd, running_mean, running_var, num_batches_tracked, C);
    FS2_CHECK_LAUNCH("fs2_bn_finalize");
    return FS2_OK;
}

extern "C" int fs2_bn_tanh_fwd(const void* x, int dtype, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, void* y, int64_t M, int C, float p, const uint64_t* rng,
                               uint32_t site, void* stream) {
    CHECK_ROW("fs2_bn_tanh_fwd", C, 1024); CHECK_DT("fs2_bn_tanh_fwd", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_bn_tanh_fwd: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(C, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((bn_tanh_fwd_k<T, NG>), grid, block, 0, st, (const T*)x, mean, rstd, gamma, beta, (T*)y, M, C, p, rng, site);
    }); } });
    FS2_CHECK_LAUNCH("fs2_bn_tanh_fwd");
    return FS2_OK;
}

extern "C" int fs2_bn_stats_tanh_fwd(const void* x, int dtype, const float* sums, float count, const float* count_dev, float eps,
                                     float momentum, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, int64_t M, int C, float p,
                                     const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_ROW("fs2_bn_stats_tanh_fwd", C, 1024); CHECK_DT("fs2_bn_stats_tanh_fwd", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_bn_stats_tanh_fwd: dropout needs rng");
    FS2_REQUIRE(sums && mean && rstd && (count > 0.f || count_dev != nullptr), "fs2_bn_stats_tanh_fwd: need sums, mean, rstd and a row count");
    FS2_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "fs2_bn_stats_tanh_fwd: running_mean and running_var come together");
    FS2_REQUIRE(M > 0, "fs2_bn_stats_tanh_fwd: no rows");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(row_grid(M)), block(ROW_BLOCK);
    NG_DISPATCH(C, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((bn_stats_tanh_fwd_k<T, NG>), grid, block, 0, st, (const T*)x, sums, count, count_dev, eps, momentum, gamma, beta,
                           (T*)y, mean, rstd, running_mean, running_var, num_batches_tracked, M, C, p, rng, site);
    }); } });
    FS2_CHECK_LAUNCH("fs2_bn_stats_tanh_fwd");
    return FS2_OK;
}

extern "C" int fs2_bn_tanh_bwd_reduce(const void* dy, const void* x, int dtype, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, float* red, int64_t M, int C, float p,
                                      const uint64_t* rng, uint32_t site, void* stream) {
    CHECK_ROW("fs2_bn_tanh_bwd_reduce", C, 1024); CHECK_DT("fs2_bn_tanh_bwd_reduce", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_bn_tanh_bwd_reduce: dropout needs rng");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(C, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((bn_tanh_bwd_k<T, NG, 0>), grid, block, 0, st, (const T*)dy, (const T*)x, mean, rstd, gamma, beta, red, 1.f, (const float*)nullptr, (T*)nullptr, (float*)nullptr, (float*)nullptr, M, C, p, rng, site, (float*)nullptr);
    }); } });
    FS2_CHECK_LAUNCH("fs2_bn_tanh_bwd_reduce");
    return FS2_OK;
}

extern "C" int fs2_bn_tanh_bwd_apply(const void* dy, const void* x, int dtype, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, const float* red, float count,
                                     const float* count_dev, void* dx, float* dgamma, float* dbeta, int64_t M, int C,
                                     float p, const uint64_t* rng, uint32_t site, float* dcolsum, void* stream) {
    CHECK_ROW("fs2_bn_tanh_bwd_apply", C, 1024); CHECK_DT("fs2_bn_tanh_bwd_apply", dtype);
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_bn_tanh_bwd_apply: dropout needs rng");
    FS2_REQUIRE(count > 0.f || count_dev != nullptr, "fs2_bn_tanh_bwd_apply: count must be positive");
    if (M <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(red_grid(M)), block(RED_BLOCK);
    NG_DISPATCH(C, NG, { if constexpr (NG <= 4) { T_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((bn_tanh_bwd_k<T, NG, 1>), grid, block, 0, st, (const T*)dy, (const T*)x, mean, rstd, gamma, beta, const_cast<float*>(red), count, count_dev, (T*)dx, dgamma, dbeta, M, C, p, rng, site, dcolsum);
    }); } });
    FS2_CHECK_LAUNCH("fs2_bn_tanh_bwd_apply");
    return FS2_OK;
}

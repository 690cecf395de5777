// Per-tensor fp8 quantisation for the fp8 operand mode of fs2_gemm (BASELINE.json configs[4]): OCP e4m3 / e5m2 as gfx950
// implements them (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32, round to nearest even).  Current scaling with a power-of-two
// scale: multiplying by it is exact, so the only rounding is the fp8 conversion itself.
#include "fs2_common.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float absf_bits(float v) { return __uint_as_float(__float_as_uint(v) & 0x7FFFFFFFu); }

// 16 bytes per lane per load (8 bf16 / 4 f32): scalar 2-byte loads run these streaming kernels at less than half the rate
template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int N = 4; typedef float4 type; };
template <> struct Vec16<bf16_t> { static constexpr int N = 8; typedef bf16x8 type; };
template <typename T> __device__ __forceinline__ void unpack16(const typename Vec16<T>::type& v, float* e);
template <> __device__ __forceinline__ void unpack16<float>(const float4& v, float* e) { e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
template <> __device__ __forceinline__ void unpack16<bf16_t>(const bf16x8& v, float* e) {
#pragma unroll
    for (int c = 0; c < 8; ++c) e[c] = (float)v[c];
}

template <typename T>
__global__ __launch_bounds__(TPB) void amax_k(const T* __restrict__ x, int64_t n, float* __restrict__ state) {
    constexpr int VN = Vec16<T>::N;
    float m = 0.f;
    const int64_t nv = n / VN;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * TPB) {
        float e[VN];
        unpack16<T>(reinterpret_cast<const typename Vec16<T>::type*>(x)[i], e);
#pragma unroll
        for (int c = 0; c < VN; ++c) { const float v = absf_bits(e[c]); m = (v == v) ? fmaxf(m, v) : m; }   // NaNs do not define the range
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nv * VN)) {
        const float v = absf_bits(to_f32<T>(x[nv * VN + threadIdx.x]));
        m = (v == v) ? fmaxf(m, v) : m;
    }
    m = wave_max(m);
    // ONE atomic per workgroup on the tensor's single state word (with one per wave, 8192 atomics on one address took ~60 us of a
    // 66 us launch: same-address float atomics serialise at the memory side); non-negative floats order like their bit patterns
    __shared__ float wm[TPB / 64];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = wm[0];
#pragma unroll
        for (int w = 1; w < TPB / 64; ++w) b = fmaxf(b, wm[w]);
        if (b > 0.f) atomicMax(reinterpret_cast<unsigned*>(state), __float_as_uint(b));
    }
}

__device__ __forceinline__ float pow2_scale(float amax, int log2max, float* inv) { return fs2_pow2_scale(amax, log2max, inv); }

// REPAIR: the codes in dst were written by a producer (FS2Gemm.q8) with the scale of prev[0]; nothing to do but 1/scale when the true
// amax state[0] gives the same scale
template <typename T, bool BF8, bool REPAIR = false>
__global__ __launch_bounds__(TPB) void quant_k(const T* __restrict__ x, unsigned* __restrict__ dst, int64_t n, float* __restrict__ state,
                                               const float* __restrict__ prev = nullptr) {
    constexpr int LOG2MAX = BF8 ? 15 : 8;        // e5m2: max 57344 >= 2^15; e4m3: max 448 >= 2^8
    constexpr float FMAX = BF8 ? 57344.f : 448.f;
    float inv;
    const float scale = pow2_scale(state[0], LOG2MAX, &inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) state[1] = inv;
    if constexpr (REPAIR) {
        float inv_spec;
        if (pow2_scale(prev[0], LOG2MAX, &inv_spec) == scale) return;
    }
    // 16 source elements -> one 16-byte store of fp8 codes per iteration (two or four 16-byte loads)
    constexpr int VN = Vec16<T>::N;
    const int64_t n16 = n >> 4;
    auto pack = [&](const float* v) {
        int word = 0;
        float c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = fminf(fmaxf(v[j] * scale, -FMAX), FMAX);   // saturate (|t| < 2^LOG2MAX <= FMAX anyway; NaN -> NaN)
        if constexpr (BF8) {
            word = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], word, false);
            word = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], word, true);
        } else {
            word = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], word, false);
            word = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], word, true);
        }
        return (unsigned)word;
    };
    for (int64_t q = (int64_t)blockIdx.x * TPB + threadIdx.x; q < n16; q += (int64_t)gridDim.x * TPB) {
        float e[16];
#pragma unroll
        for (int j = 0; j < 16 / VN; ++j) unpack16<T>(reinterpret_cast<const typename Vec16<T>::type*>(x)[q * (16 / VN) + j], e + j * VN);
        uint4 o;
        o.x = pack(e); o.y = pack(e + 4); o.z = pack(e + 8); o.w = pack(e + 12);
        reinterpret_cast<uint4*>(dst)[q] = o;
    }
    // tail (< 16 elements): one word per thread of the first block
    const int64_t t0 = n16 << 4;
    if (blockIdx.x == 0 && threadIdx.x < 4 && t0 + 4 * threadIdx.x < n) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int64_t i = t0 + 4 * threadIdx.x + c; v[c] = i < n ? to_f32<T>(x[i]) : 0.f; }
        dst[(t0 >> 2) + threadIdx.x] = pack(v);
    }
}

// ---- many tensors in two launches (the weight shadows of a model, once per optimizer step): block b of the grid works on chunk
//      b - block_begin of the tensor whose [block_begin, block_begin + nblocks) contains b; chunks of 2048 x 16 elements
constexpr int CHUNK16 = 2048;
template <int PASS, bool BF8>
__global__ __launch_bounds__(TPB) void quant_batched_k(const FS2QuantDesc* __restrict__ tab, int nt) {
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FS2QuantDesc d = tab[lo];
    const int cb = (int)blockIdx.x - d.block_begin;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.src);
    const int64_t n16 = d.n >> 4;
    const int64_t q0 = (int64_t)cb * CHUNK16, q1 = q0 + CHUNK16 < n16 ? q0 + CHUNK16 : n16;
    const int64_t t0 = n16 << 4;
    if constexpr (PASS == 0) {
        float m = 0.f;
        for (int64_t q = q0 + threadIdx.x; q < q1; q += TPB) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[2 * q + j];
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float a = absf_bits((float)v[c]); m = (a == a) ? fmaxf(m, a) : m; }
            }
        }
        if (cb == 0 && t0 + threadIdx.x < d.n) { const float a = absf_bits((float)x[t0 + threadIdx.x]); m = (a == a) ? fmaxf(m, a) : m; }
        m = wave_max(m);
        __shared__ float wm[TPB / 64];
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = wm[0];
#pragma unroll
            for (int w = 1; w < TPB / 64; ++w) b = fmaxf(b, wm[w]);
            if (b > 0.f) atomicMax(reinterpret_cast<unsigned*>(d.state), __float_as_uint(b));
        }
    } else {
        constexpr int LOG2MAX = BF8 ? 15 : 8;
        constexpr float FMAX = BF8 ? 57344.f : 448.f;
        float inv;
        const float scale = pow2_scale(d.state[0], LOG2MAX, &inv);
        if (cb == 0 && threadIdx.x == 0) d.state[1] = inv;
        auto pack = [&](const float* v) {
            int word = 0;
            float c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = fminf(fmaxf(v[j] * scale, -FMAX), FMAX);
            if constexpr (BF8) {
                word = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], word, false);
                word = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], word, true);
            } else {
                word = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], word, false);
                word = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], word, true);
            }
            return (unsigned)word;
        };
        unsigned* dst = reinterpret_cast<unsigned*>(d.dst);
        for (int64_t q = q0 + threadIdx.x; q < q1; q += TPB) {
            float e[16];
#pragma unroll
            for (int j = 0; j < 2; ++j) unpack16<bf16_t>(reinterpret_cast<const bf16x8*>(x)[2 * q + j], e + 8 * j);
            uint4 o;
            o.x = pack(e); o.y = pack(e + 4); o.z = pack(e + 8); o.w = pack(e + 12);
            reinterpret_cast<uint4*>(dst)[q] = o;
        }
        if (cb == 0 && threadIdx.x < 4 && t0 + 4 * threadIdx.x < d.n) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int64_t i = t0 + 4 * threadIdx.x + c; v[c] = i < d.n ? (float)x[i] : 0.f; }
            dst[(t0 >> 2) + threadIdx.x] = pack(v);
        }
    }
}

inline int flat_grid(int64_t n, int cap = 2048) {
    int64_t b = (n + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int fs2_amax(const void* src, int src_dtype, int64_t n, float* state, void* stream) {
    FS2_REQUIRE(src_dtype == FS2_F32 || src_dtype == FS2_BF16, "fs2_amax: bad dtype %d", src_dtype);
    FS2_REQUIRE(n > 0 && src && state, "fs2_amax: bad arguments");
    FS2_REQUIRE(fs2_aligned16(src), "fs2_amax: src must be 16-byte aligned");
    if (src_dtype == FS2_F32) hipLaunchKernelGGL((amax_k<float>), dim3(flat_grid((n + 3) >> 2, 1024)), dim3(TPB), 0, (hipStream_t)stream, (const float*)src, n, state);
    else hipLaunchKernelGGL((amax_k<bf16_t>), dim3(flat_grid((n + 7) >> 3, 1024)), dim3(TPB), 0, (hipStream_t)stream, (const bf16_t*)src, n, state);
    FS2_CHECK_LAUNCH("fs2_amax");
    return FS2_OK;
}

extern "C" int fs2_quantize_fp8(const void* src, int src_dtype, void* dst, int bf8, int64_t n, float* state, void* stream) {
    FS2_REQUIRE(src_dtype == FS2_F32 || src_dtype == FS2_BF16, "fs2_quantize_fp8: bad dtype %d", src_dtype);
    FS2_REQUIRE(n > 0 && src && dst && state, "fs2_quantize_fp8: bad arguments");
    FS2_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 3u) == 0, "fs2_quantize_fp8: dst must be 4-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    FS2_REQUIRE(fs2_aligned16(src) && fs2_aligned16(dst), "fs2_quantize_fp8: src and dst must be 16-byte aligned");
    dim3 grid(flat_grid((n + 15) >> 4)), block(TPB);
    if (src_dtype == FS2_F32) {
        if (bf8) hipLaunchKernelGGL((quant_k<float, true>), grid, block, 0, st, (const float*)src, (unsigned*)dst, n, state, (const float*)nullptr);
        else hipLaunchKernelGGL((quant_k<float, false>), grid, block, 0, st, (const float*)src, (unsigned*)dst, n, state, (const float*)nullptr);
    } else {
        if (bf8) hipLaunchKernelGGL((quant_k<bf16_t, true>), grid, block, 0, st, (const bf16_t*)src, (unsigned*)dst, n, state, (const float*)nullptr);
        else hipLaunchKernelGGL((quant_k<bf16_t, false>), grid, block, 0, st, (const bf16_t*)src, (unsigned*)dst, n, state, (const float*)nullptr);
    }
    FS2_CHECK_LAUNCH("fs2_quantize_fp8");
    return FS2_OK;
}

extern "C" int fs2_quantize_fp8_repair(const void* src, int src_dtype, void* dst, int bf8, int64_t n, float* state, const float* prev, void* stream) {
    FS2_REQUIRE(src_dtype == FS2_BF16, "fs2_quantize_fp8_repair: bf16 sources only (dtype %d)", src_dtype);
    FS2_REQUIRE(n > 0 && src && dst && state && prev, "fs2_quantize_fp8_repair: bad arguments");
    FS2_REQUIRE(fs2_aligned16(src) && fs2_aligned16(dst), "fs2_quantize_fp8_repair: src and dst must be 16-byte aligned");
    dim3 grid(flat_grid((n + 15) >> 4, 128)), block(TPB);      // (a no-op in the common case: few workgroups)
    if (bf8) hipLaunchKernelGGL((quant_k<bf16_t, true, true>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)src, (unsigned*)dst, n, state, prev);
    else hipLaunchKernelGGL((quant_k<bf16_t, false, true>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)src, (unsigned*)dst, n, state, prev);
    FS2_CHECK_LAUNCH("fs2_quantize_fp8_repair");
    return FS2_OK;
}

extern "C" int fs2_quantize_fp8_batched(const FS2QuantDesc* table, int n, int total_blocks, int bf8, void* stream) {
    FS2_REQUIRE(table != nullptr && n >= 1 && total_blocks >= 1, "fs2_quantize_fp8_batched: empty table");
    hipStream_t st = (hipStream_t)stream;
    if (bf8) {
        hipLaunchKernelGGL((quant_batched_k<0, true>), dim3(total_blocks), dim3(TPB), 0, st, table, n);
        hipLaunchKernelGGL((quant_batched_k<1, true>), dim3(total_blocks), dim3(TPB), 0, st, table, n);
    } else {
        hipLaunchKernelGGL((quant_batched_k<0, false>), dim3(total_blocks), dim3(TPB), 0, st, table, n);
        hipLaunchKernelGGL((quant_batched_k<1, false>), dim3(total_blocks), dim3(TPB), 0, st, table, n);
    }
    FS2_CHECK_LAUNCH("fs2_quantize_fp8_batched");
    return FS2_OK;
}

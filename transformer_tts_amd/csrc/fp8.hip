// Per-tensor fp8 quantisation for the fp8 operand mode of fs2_gemm (BASELINE.json configs[4]): OCP e4m3 / e5m2 as gfx950
// implements them (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32, round to nearest even).  Current scaling with a power-of-two
// scale: multiplying by it is exact, so the only rounding is the fp8 conversion itself.
#include "common.cuh"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float absf_bits(float v) { return __uint_as_float(__float_as_uint(v) & 0x7FFFFFFFu); }

template <typename T>
__global__ __launch_bounds__(TPB) void amax_k(const T* __restrict__ x, int64_t n, float* __restrict__ state) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float v = absf_bits(to_f32<T>(x[i]));
        m = (v == v) ? fmaxf(m, v) : m;          // NaNs do not define the range
    }
    m = wave_max(m);
    // non-negative floats order like their bit patterns: integer atomicMax
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(state), __float_as_uint(m));
}

// scale = 2^k with amax * 2^k < 2^LOG2MAX: k = LOG2MAX - 1 - exponent(amax)   (amax in [2^e, 2^(e+1)))
__device__ __forceinline__ float pow2_scale(float amax, int log2max, float* inv) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) { *inv = 1.f; return 1.f; }
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu) - 127;
    if (e < -126) e = -126;                       // subnormal amax
    int k = log2max - 1 - e;
    k = k > 126 ? 126 : (k < -126 ? -126 : k);
    *inv = __uint_as_float((unsigned)(127 - k) << 23);
    return __uint_as_float((unsigned)(127 + k) << 23);
}

template <typename T, bool BF8>
__global__ __launch_bounds__(TPB) void quant_k(const T* __restrict__ x, unsigned* __restrict__ dst, int64_t n, float* __restrict__ state) {
    constexpr int LOG2MAX = BF8 ? 15 : 8;        // e5m2: max 57344 >= 2^15; e4m3: max 448 >= 2^8
    constexpr float FMAX = BF8 ? 57344.f : 448.f;
    float inv;
    const float scale = pow2_scale(state[0], LOG2MAX, &inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) state[1] = inv;
    const int64_t nw = (n + 3) >> 2;              // one 32-bit word = 4 fp8 per iteration
    for (int64_t w = (int64_t)blockIdx.x * TPB + threadIdx.x; w < nw; w += (int64_t)gridDim.x * TPB) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t i = 4 * w + c;
            float t = i < n ? to_f32<T>(x[i]) * scale : 0.f;
            t = fminf(fmaxf(t, -FMAX), FMAX);     // saturate (the scale keeps |t| < 2^LOG2MAX <= FMAX anyway; NaN -> NaN)
            v[c] = t;
        }
        int word = 0;
        if constexpr (BF8) {
            word = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], word, false);
            word = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], word, true);
        } else {
            word = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], word, false);
            word = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], word, true);
        }
        dst[w] = (unsigned)word;
    }
}

inline int flat_grid(int64_t n) {
    int64_t b = (n + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int fs2_amax(const void* src, int src_dtype, int64_t n, float* state, void* stream) {
    FS2_REQUIRE(src_dtype == FS2_F32 || src_dtype == FS2_BF16, "fs2_amax: bad dtype %d", src_dtype);
    FS2_REQUIRE(n > 0 && src && state, "fs2_amax: bad arguments");
    if (src_dtype == FS2_F32) hipLaunchKernelGGL((amax_k<float>), dim3(flat_grid(n)), dim3(TPB), 0, (hipStream_t)stream, (const float*)src, n, state);
    else hipLaunchKernelGGL((amax_k<bf16_t>), dim3(flat_grid(n)), dim3(TPB), 0, (hipStream_t)stream, (const bf16_t*)src, n, state);
    FS2_CHECK_LAUNCH("fs2_amax");
    return FS2_OK;
}

extern "C" int fs2_quantize_fp8(const void* src, int src_dtype, void* dst, int bf8, int64_t n, float* state, void* stream) {
    FS2_REQUIRE(src_dtype == FS2_F32 || src_dtype == FS2_BF16, "fs2_quantize_fp8: bad dtype %d", src_dtype);
    FS2_REQUIRE(n > 0 && src && dst && state, "fs2_quantize_fp8: bad arguments");
    FS2_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 3u) == 0, "fs2_quantize_fp8: dst must be 4-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(flat_grid((n + 3) >> 2)), block(TPB);
    if (src_dtype == FS2_F32) {
        if (bf8) hipLaunchKernelGGL((quant_k<float, true>), grid, block, 0, st, (const float*)src, (unsigned*)dst, n, state);
        else hipLaunchKernelGGL((quant_k<float, false>), grid, block, 0, st, (const float*)src, (unsigned*)dst, n, state);
    } else {
        if (bf8) hipLaunchKernelGGL((quant_k<bf16_t, true>), grid, block, 0, st, (const bf16_t*)src, (unsigned*)dst, n, state);
        else hipLaunchKernelGGL((quant_k<bf16_t, false>), grid, block, 0, st, (const bf16_t*)src, (unsigned*)dst, n, state);
    }
    FS2_CHECK_LAUNCH("fs2_quantize_fp8");
    return FS2_OK;
}

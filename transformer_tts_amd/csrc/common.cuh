// Shared device/host helpers for the gfx950 kernels of libfs2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "fs2_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define FS2_WAVE 64

// ---------------------------------------------------------------- host-side error reporting
void fs2_set_error(const char* fmt, ...);
#define FS2_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            fs2_set_error(__VA_ARGS__);   \
            return FS2_EINVAL;            \
        }                                 \
    } while (0)
#define FS2_CHECK_LAUNCH(name)                                                    \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            fs2_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return FS2_ELAUNCH;                                                   \
        }                                                                         \
    } while (0)

static inline bool fs2_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------- element access by dtype
template <typename T> struct DType;
template <> struct DType<float> { static constexpr int code = FS2_F32; };
template <> struct DType<bf16_t> { static constexpr int code = FS2_BF16; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// load / store 4 consecutive elements (16-B aligned for float, 8-B for bf16) as float4
template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ---------------------------------------------------------------- wave / block reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- Philox4x32-7 dropout stream
// Seven rounds is the Crush-resistant minimum of Salmon et al. (SC'11); the 32x32->64 products compile to one
// quarter-rate v_mad_u64_u32 each, so a call costs 14 of them (the 10-round mul_hi/mul_lo form cost 40 multiplies
// and made the attention softmax ALU-bound).
constexpr int PHILOX_ROUNDS = 7;
struct Philox4 { uint32_t x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}

// Keep-mask scaling factors for the 4 consecutive elements [4*q, 4*q+4) of a tensor at call site `site`.
// Returns 1/(1-p) for kept elements and 0 for dropped ones.  p == 0 -> all ones without touching rng.
struct DropCtx {
    uint32_t k0, k1, off, site;
    uint32_t thr;       // an element is kept iff its 32 random bits are >= thr = floor(p * 2^32)
    float p, scale;
    bool on;
};
__device__ __forceinline__ DropCtx drop_ctx(const uint64_t* rng, uint32_t site, float p) {
    DropCtx c;
    c.on = p > 0.f;
    c.p = p;
    c.scale = c.on ? 1.f / (1.f - p) : 1.f;
    c.thr = (uint32_t)((double)p * 4294967296.0);
    c.site = site;
    if (c.on) {
        uint64_t seed = rng[0], off = rng[1];
        c.k0 = (uint32_t)seed;
        c.k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32);
        c.off = (uint32_t)off;
    } else {
        c.k0 = c.k1 = c.off = 0;
    }
    return c;
}
__device__ __forceinline__ float4 drop_scale4(const DropCtx& c, uint64_t q) {
    if (!c.on) return make_float4(1.f, 1.f, 1.f, 1.f);
    Philox4 r = philox4x32((uint32_t)q, (uint32_t)(q >> 32), c.site, c.off, c.k0, c.k1);
    float4 o;
    o.x = (r.x >= c.thr) ? c.scale : 0.f;
    o.y = (r.y >= c.thr) ? c.scale : 0.f;
    o.z = (r.z >= c.thr) ? c.scale : 0.f;
    o.w = (r.w >= c.thr) ? c.scale : 0.f;
    return o;
}

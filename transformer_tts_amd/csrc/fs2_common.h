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

// hipFuncAttributeMaxDynamicSharedMemorySize (> 64 KiB of dynamic LDS) is a per-DEVICE attribute of a kernel: a launcher remembers
// per device whether it has been set (one process per GPU is the deployment, but a process that drives several must work too).
struct Fs2PerDevice {
    bool done[16] = {};
    bool need() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 16) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

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
// (FS2_NT_STORES: measurement build -- the row kernels' output stores carry the nontemporal hint, profiles/r04_h_nt_stores.txt)
template <> __device__ __forceinline__ void store4<float>(float* p, float4 v) {
#ifdef FS2_NT_STORES
    typedef float __attribute__((ext_vector_type(4))) f32x4nt;
    __builtin_nontemporal_store(f32x4nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4nt*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
#ifdef FS2_NT_STORES
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(p));
#else
    *reinterpret_cast<bf16x4*>(p) = o;
#endif
}

// ---------------------------------------------------------------- workgroup barrier for LDS hand-offs
// __syncthreads() is a workgroup fence + barrier: once a kernel has global stores or atomics outstanding (a GEMM
// epilogue, the probability rows of the attention kernels) the fence makes hipcc emit s_waitcnt vmcnt(0) in front
// of EVERY later barrier, which also drains the register prefetch ring of global loads at every stage.  The LDS
// producer/consumer hand-offs of a staged pipeline only need the wave's own LDS operations retired:
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------- wave reductions (wave = 64 lanes)
// Four DPP steps (quad xor 1, quad xor 2, row rotate 4 and 8) leave the total of each 16-lane row in all of its
// lanes; the four row totals are then combined through v_readlane.  No LDS crossbar (ds_bpermute, what __shfl_xor
// compiles to) and no dependent ~100-cycle round trips: the row kernels do 2-3 of these reductions per row.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_value(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// fp8 per-tensor scale: 2^k with amax * 2^k < 2^LOG2MAX (k = LOG2MAX - 1 - exponent(amax)); LOG2MAX = 8 for e4m3 (max 448), 15 for
// e5m2 (max 57344); *inv = 2^-k
__device__ __forceinline__ float fs2_pow2_scale(float amax, int log2max, float* inv) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) { *inv = 1.f; return 1.f; }
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu) - 127;
    if (e < -126) e = -126;                       // subnormal amax
    int k = log2max - 1 - e;
    k = k > 126 ? 126 : (k < -126 ? -126 : k);
    *inv = __uint_as_float((unsigned)(127 - k) << 23);
    return __uint_as_float((unsigned)(127 + k) << 23);
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);     // row_ror:4
    v += dpp_mov<0x128>(v);     // row_ror:8
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
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

// Keep-mask scaling factors.  One Philox call serves EIGHT consecutive elements of a tensor at call site `site`
// (counter = element index >> 3); element e uses the 16-bit half (e & 1) of output word (e & 7) >> 1 and is kept iff
// that half is >= thr16 = round(p * 65536).  The factor of a kept element is 65536 / (65536 - thr16) (= 1/(1-p) up to
// the 2^-16 quantisation of p, so the mask stays exactly unbiased); p == 0 -> all ones without touching rng.
struct DropCtx {
    uint32_t k0, k1, off, site;
    uint32_t thr16;
    float p, scale;
    bool on;
};
__device__ __forceinline__ DropCtx drop_ctx(const uint64_t* rng, uint32_t site, float p) {
    DropCtx c;
    c.on = p > 0.f;
    c.p = p;
    c.thr16 = (uint32_t)(p * 65536.f + 0.5f);
    c.scale = c.on ? 65536.f / (65536.f - (float)c.thr16) : 1.f;
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
// factors of elements [8*q8, 8*q8 + 8)
__device__ __forceinline__ void drop_scale8(const DropCtx& c, uint64_t q8, float (&o)[8]) {
    if (!c.on) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 1.f;
        return;
    }
    const Philox4 r = philox4x32((uint32_t)q8, (uint32_t)(q8 >> 32), c.site, c.off, c.k0, c.k1);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = ((w[i] & 0xFFFFu) >= c.thr16) ? c.scale : 0.f;
        o[2 * i + 1] = ((w[i] >> 16) >= c.thr16) ? c.scale : 0.f;
    }
}
// factors of elements [4*q4, 4*q4 + 4): the lower or upper half of the call of q4 >> 1
__device__ __forceinline__ float4 drop_scale4(const DropCtx& c, uint64_t q4) {
    if (!c.on) return make_float4(1.f, 1.f, 1.f, 1.f);
    const uint64_t q8 = q4 >> 1;
    const Philox4 r = philox4x32((uint32_t)q8, (uint32_t)(q8 >> 32), c.site, c.off, c.k0, c.k1);
    const bool hi = (q4 & 1) != 0;
    const uint32_t w0 = hi ? r.z : r.x, w1 = hi ? r.w : r.y;
    float4 o;
    o.x = ((w0 & 0xFFFFu) >= c.thr16) ? c.scale : 0.f;
    o.y = ((w0 >> 16) >= c.thr16) ? c.scale : 0.f;
    o.z = ((w1 & 0xFFFFu) >= c.thr16) ? c.scale : 0.f;
    o.w = ((w1 >> 16) >= c.thr16) ? c.scale : 0.f;
    return o;
}

// MFMA GEMM family for gfx950: linear / implicit-GEMM Conv1d / batched attention products, forward,
// dgrad and wgrad, in bf16 (v_mfma_f32_16x16x32_bf16) or exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Block = 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 4x4 MFMA tiles of 16x16.
// K is consumed in stages of 128 BYTES per row (64 bf16 / 32 f32): both dtypes share one LDS geometry,
// one 16-byte fragment read per lane per 64-byte k-step; only the MFMA issue differs
// (1 x 16x16x32 bf16, or 4 x 16x16x4 f32 whose k-slots are the 4 floats of the same 16 bytes).
//
// LDS images (16 KiB per operand per stage, double buffered -> 64 KiB; epilogue reuses them):
//   row-major operand  [128 rows][128 B], 16-B chunk c of row r stored at c ^ ((r>>1)&7)
//                      -> the 16 rows of a ds_read_b128 lane group fall on 16 distinct 16-B slots;
//   k-major bf16       [64 k][256 B], chunk c of row r at c ^ (((r&3)<<2)|((r>>2)&3)) and read with
//                      ds_read_b64_tr_b16 (hardware transpose; image (b) of the CDNA4 guide, T10);
//   k-major f32        [32 k][512 B], plain, read with ds_read_b32.
// Staging is global -> registers -> LDS with the next stage's loads issued before the MFMAs of the
// current one (one barrier per stage).  Out-of-range rows / conv halo rows are zero-filled by predicate.
#include "common.cuh"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int STAGE_BYTES = 16384;                 // one operand, one stage
constexpr int EPI_LD = 68;                         // floats per epilogue row (64 + 4 pad: conflict-free)
constexpr int EPI_BYTES = 4 * 64 * EPI_LD * 4;     // 4 waves x 64 rows
constexpr int SMEM_BYTES = (4 * STAGE_BYTES > EPI_BYTES) ? 4 * STAGE_BYTES : EPI_BYTES;

template <typename T, bool KM> struct Tile {
    static constexpr int EPC = 16 / (int)sizeof(T);                 // elements per 16-B chunk
    static constexpr int BK = 128 / (int)sizeof(T);                 // k elements per stage
    static constexpr int CPR = KM ? (128 * (int)sizeof(T)) / 16 : 8;  // chunks per LDS row
    static constexpr int ROWB = CPR * 16;                           // bytes per LDS row
    __device__ static __forceinline__ int lds_off(int row, int ch) {
        if constexpr (!KM) return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
        else if constexpr (sizeof(T) == 2) return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
        else return row * 512 + (ch << 4);
    }
};

// bijective XCD-aware remap: blocks that share an XCD (id % 8) get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

// fragment of a 16-wide tile (rows/cols o0..o0+15 of the operand) for k-step ks (64 bytes of k)
template <typename T, bool KM>
__device__ __forceinline__ typename Frag<T>::type read_frag(const unsigned char* lds, int o0, int ks, int lane) {
    typedef Tile<T, KM> TL;
    const int g = lane >> 4, i16 = lane & 15;
    if constexpr (!KM) {
        const int row = o0 + i16;
        return *reinterpret_cast<const typename Frag<T>::type*>(lds + TL::lds_off(row, ks * 4 + g));
    } else if constexpr (sizeof(T) == 2) {
        // k rows ks*32 + 8g + {0..3} and {4..7}; lane 4q+p of the 16-lane group addresses row q, columns 4p..4p+3
        const int q = i16 >> 2, pp = i16 & 3;
        const int ch = (o0 >> 3) + (pp >> 1);
        const int r0 = ks * 32 + 8 * g + q;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + TL::lds_off(r0, ch) + 8 * (pp & 1)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + TL::lds_off(r0 + 4, ch) + 8 * (pp & 1)));
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
        u.s.lo = lo; u.s.hi = hi;
        return u.v;
    } else {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = *reinterpret_cast<const float*>(lds + (ks * 16 + 4 * g + e) * 512 + (o0 + i16) * 4);
        return v;
    }
}

template <typename T>
__device__ __forceinline__ void mma(f32x4& acc, const typename Frag<T>::type& a, const typename Frag<T>::type& b) {
    if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
}

template <typename T, typename TC, bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const FS2Gemm p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Tile<T, AKM> TA;
    typedef Tile<T, BKM> TB;
    constexpr int EPC = TA::EPC, BK = TA::BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
    const int lin = xcd_remap(blockIdx.x, tilesM * tilesN);
    const int m0 = (lin / tilesN) * BM, n0 = (lin % tilesN) * BN;
    int z = blockIdx.y;
    const int split = z % p.split_k; z /= p.split_k;
    const int b2 = z % p.batch2, b1 = z / p.batch2;

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + b1 * p.sA1 + b2 * p.sA2;
    const T* __restrict__ B = reinterpret_cast<const T*>(p.B) + b1 * p.sB1 + b2 * p.sB2;
    const int64_t coff = b1 * p.sC1 + b2 * p.sC2;

    const int Kb = p.Kb > 0 ? p.Kb : p.K;
    const int nkt = (p.K + BK - 1) / BK;                       // k tiles per tap
    const int ntot = (p.conv == 1 ? p.taps : 1) * nkt;         // stages in the whole reduction
    const int per = (ntot + p.split_k - 1) / p.split_k;
    const int it0 = split * per, it1 = min(ntot, it0 + per);
    const int seq = p.seq_len;
    const int shiftB = (p.conv == 2) ? (b2 - p.pad) : 0;

    // ---- per-thread staging coordinates (4 chunks of 16 B per operand per stage)
    int a_row[4], a_ch[4], a_t[4], b_row[4], b_ch[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        a_row[i] = c / TA::CPR; a_ch[i] = c % TA::CPR;
        b_row[i] = c / TB::CPR; b_ch[i] = c % TB::CPR;
        a_t[i] = 0;
        if constexpr (!AKM) { if (p.conv == 1) a_t[i] = (m0 + a_row[i]) % seq; }
    }

    uint4 ra[4], rb[4];
    auto load_stage = [&](int it) {
        const int tap = (p.conv == 1) ? it / nkt : 0;
        const int kb = (it - tap * nkt) * BK;
        const int shiftA = (p.conv == 1) ? tap - p.pad : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (!AKM) {
                const int m = m0 + a_row[i], k = kb + a_ch[i] * EPC;
                bool ok = (m < p.M) && (k < p.K);
                if (p.conv == 1) { const int tt = a_t[i] + shiftA; ok = ok && (tt >= 0) && (tt < seq); }
                if (ok) v = *reinterpret_cast<const uint4*>(A + (int64_t)(m + shiftA) * p.lda + k);
            } else {
                const int k = kb + a_row[i], m = m0 + a_ch[i] * EPC;
                if ((k < p.K) && (m < p.M)) v = *reinterpret_cast<const uint4*>(A + (int64_t)k * p.lda + m);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (!BKM) {
                const int n = n0 + b_row[i], k = kb + b_ch[i] * EPC;
                if ((n < p.N) && (k < p.K)) v = *reinterpret_cast<const uint4*>(B + (int64_t)n * p.ldb + (int64_t)tap * p.K + k);
            } else {
                const int k = kb + b_row[i], n = n0 + b_ch[i] * EPC;
                bool ok = (k < Kb) && (n < p.N);
                if (p.conv == 2) { const int tt = (k % seq) + shiftB; ok = ok && (tt >= 0) && (tt < seq); }
                if (ok) v = *reinterpret_cast<const uint4*>(B + (int64_t)(k + shiftB) * p.ldb + n);
            }
            rb[i] = v;
        }
    };
    auto store_stage = [&](int buf) {
        unsigned char* la = smem + buf * 2 * STAGE_BYTES;
        unsigned char* lb = la + STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(la + TA::lds_off(a_row[i], a_ch[i])) = ra[i];
            *reinterpret_cast<uint4*>(lb + TB::lds_off(b_row[i], b_ch[i])) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (it0 < it1) {
        load_stage(it0);
        store_stage(0);
        __syncthreads();
        for (int it = it0; it < it1; ++it) {
            const int buf = (it - it0) & 1;
            if (it + 1 < it1) load_stage(it + 1);
            const unsigned char* la = smem + buf * 2 * STAGE_BYTES;
            const unsigned char* lb = la + STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typename Frag<T>::type fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = read_frag<T, AKM>(la, wr * 64 + i * 16, ks, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = read_frag<T, BKM>(lb, wc * 64 + j * 16, ks, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma<T>(acc[i][j], fa[i], fb[j]);
            }
            if (it + 1 < it1) store_stage(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: accumulators -> wave-private LDS rows -> coalesced 4-column groups
    __syncthreads();
    float* ew = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
    {
        const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ew[(i * 16 + g * 4 + r) * EPI_LD + j * 16 + i16] = acc[i][j][r];
    }
    __syncthreads();
    TC* __restrict__ C = reinterpret_cast<TC*>(p.C) + coff;
    const int cg = lane & 15;                      // column group: 4 columns
    const int ncol = n0 + wc * 64 + cg * 4;
    const bool col_ok = ncol < p.N;                // N is a multiple of 4 (checked on the host)
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias != nullptr && col_ok && split == 0) bias4 = *reinterpret_cast<const float4*>(p.bias + ncol);
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int rl = pass * 4 + (lane >> 4);
        const int m = m0 + wr * 64 + rl;
        float4 v = *reinterpret_cast<const float4*>(ew + rl * EPI_LD + cg * 4);
        if (!(col_ok && m < p.M)) continue;
        v.x = v.x * p.alpha + bias4.x; v.y = v.y * p.alpha + bias4.y;
        v.z = v.z * p.alpha + bias4.z; v.w = v.w * p.alpha + bias4.w;
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.relu_mask != nullptr) {
            const float4 mk = load4<T>(reinterpret_cast<const T*>(p.relu_mask) + coff + (int64_t)m * p.ldm + ncol);
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
            v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        if (p.residual != nullptr) {
            float4 rr;
            if (p.res_dtype == FS2_F32) rr = load4<float>(reinterpret_cast<const float*>(p.residual) + coff + (int64_t)m * p.ldr + ncol);
            else rr = load4<bf16_t>(reinterpret_cast<const bf16_t*>(p.residual) + coff + (int64_t)m * p.ldr + ncol);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        TC* dst = C + (int64_t)m * p.ldc + ncol;
        if constexpr (sizeof(TC) == 4) {
            if (p.accumulate) {
                atomicAdd(reinterpret_cast<float*>(dst) + 0, v.x); atomicAdd(reinterpret_cast<float*>(dst) + 1, v.y);
                atomicAdd(reinterpret_cast<float*>(dst) + 2, v.z); atomicAdd(reinterpret_cast<float*>(dst) + 3, v.w);
            } else {
                store4<TC>(dst, v);
            }
        } else {
            store4<TC>(dst, v);
            // statistics are taken on the values as stored (rounded to TC)
            v.x = (float)(TC)v.x; v.y = (float)(TC)v.y; v.z = (float)(TC)v.z; v.w = (float)(TC)v.w;
        }
        if (p.colstats != nullptr) {
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
        }
    }
    if (p.colstats != nullptr) {
        // reduce over the 4 row groups (lane>>4), then one atomic per column per wave
        float* f[2] = {&cs.x, &cq.x};
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = f[w][e];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                f[w][e] = s;
            }
        if ((lane >> 4) == 0 && col_ok) {
            atomicAdd(p.colstats + ncol + 0, cs.x); atomicAdd(p.colstats + ncol + 1, cs.y);
            atomicAdd(p.colstats + ncol + 2, cs.z); atomicAdd(p.colstats + ncol + 3, cs.w);
            atomicAdd(p.colstats + p.N + ncol + 0, cq.x); atomicAdd(p.colstats + p.N + ncol + 1, cq.y);
            atomicAdd(p.colstats + p.N + ncol + 2, cq.z); atomicAdd(p.colstats + p.N + ncol + 3, cq.w);
        }
    }
}

template <typename T, typename TC>
int launch(const FS2Gemm& g, dim3 grid, hipStream_t st) {
    if (!g.a_kmajor && !g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, false, false>), grid, dim3(256), SMEM_BYTES, st, g);
    else if (!g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, false, true>), grid, dim3(256), SMEM_BYTES, st, g);
    else if (g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, true, true>), grid, dim3(256), SMEM_BYTES, st, g);
    else { fs2_set_error("fs2_gemm: a_kmajor=1 with b_kmajor=0 is not provided"); return FS2_EINVAL; }
    FS2_CHECK_LAUNCH("fs2_gemm");
    return FS2_OK;
}

}  // namespace

extern "C" int fs2_gemm(const FS2Gemm* gp, void* stream) {
    FS2_REQUIRE(gp != nullptr, "fs2_gemm: null descriptor");
    FS2Gemm g = *gp;
    FS2_REQUIRE(g.dtype == FS2_F32 || g.dtype == FS2_BF16, "fs2_gemm: bad dtype %d", g.dtype);
    FS2_REQUIRE(g.c_dtype == FS2_F32 || g.c_dtype == g.dtype, "fs2_gemm: c_dtype must be f32 or the operand dtype");
    FS2_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "fs2_gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    if (g.split_k < 1) g.split_k = 1;
    if (g.batch1 < 1) g.batch1 = 1;
    if (g.batch2 < 1) g.batch2 = 1;
    const int epc = g.dtype == FS2_BF16 ? 8 : 4;
    FS2_REQUIRE(g.A && g.B && g.C, "fs2_gemm: null operand");
    FS2_REQUIRE(fs2_aligned16(g.A) && fs2_aligned16(g.B) && fs2_aligned16(g.C), "fs2_gemm: operands must be 16-byte aligned");
    FS2_REQUIRE(g.lda % epc == 0 && g.ldb % epc == 0, "fs2_gemm: lda/ldb must be multiples of %d elements", epc);
    FS2_REQUIRE(g.sA1 % epc == 0 && g.sA2 % epc == 0 && g.sB1 % epc == 0 && g.sB2 % epc == 0,
                "fs2_gemm: batch strides of A/B must be multiples of %d elements", epc);
    FS2_REQUIRE(g.N % 4 == 0 || g.ldc >= ((g.N + 3) / 4) * 4, "fs2_gemm: N=%d needs ldc >= N rounded up to 4", g.N);
    FS2_REQUIRE(g.ldc % 4 == 0 && g.sC1 % 4 == 0 && g.sC2 % 4 == 0, "fs2_gemm: ldc and C batch strides must be multiples of 4");
    if (!g.a_kmajor) FS2_REQUIRE(g.K % epc == 0, "fs2_gemm: K=%d must be a multiple of %d for a row-major A", g.K, epc);
    if (!g.a_kmajor && !g.b_kmajor) FS2_REQUIRE(g.K % epc == 0, "fs2_gemm: K must be a multiple of %d", epc);
    if (g.a_kmajor) FS2_REQUIRE(((g.M + epc - 1) / epc) * epc <= g.lda, "fs2_gemm: k-major A needs lda >= M rounded up to %d", epc);
    if (g.b_kmajor) FS2_REQUIRE(((g.N + epc - 1) / epc) * epc <= g.ldb, "fs2_gemm: k-major B needs ldb >= N rounded up to %d", epc);
    FS2_REQUIRE(g.conv >= 0 && g.conv <= 2, "fs2_gemm: bad conv mode");
    if (g.conv == 1) FS2_REQUIRE(!g.a_kmajor && !g.b_kmajor && g.taps >= 1 && g.seq_len > 0, "fs2_gemm: conv=1 needs row-major operands, taps, seq_len");
    if (g.conv == 2) FS2_REQUIRE(g.a_kmajor && g.b_kmajor && g.seq_len > 0 && g.batch2 >= 1, "fs2_gemm: conv=2 needs k-major operands and seq_len");
    if (g.conv == 0) { g.taps = 1; g.pad = 0; if (g.seq_len <= 0) g.seq_len = 1; }
    FS2_REQUIRE(!(g.split_k > 1 && !g.accumulate), "fs2_gemm: split_k > 1 requires accumulate");
    FS2_REQUIRE(!(g.accumulate && g.c_dtype != FS2_F32), "fs2_gemm: accumulate requires fp32 C");
    FS2_REQUIRE(!(g.accumulate && (g.relu || g.relu_mask || g.residual || g.colstats)), "fs2_gemm: accumulate excludes relu/mask/residual/colstats");
    if (g.residual) FS2_REQUIRE(g.ldr % 4 == 0, "fs2_gemm: ldr must be a multiple of 4");
    if (g.relu_mask) FS2_REQUIRE(g.ldm % 4 == 0, "fs2_gemm: ldm must be a multiple of 4");
    if (g.bias) FS2_REQUIRE(fs2_aligned16(g.bias), "fs2_gemm: bias must be 16-byte aligned");
    // columns are stored in groups of 4: a ragged N writes zeros (products of zero-filled B rows) into
    // the pad columns [N, roundup4(N)) and cannot be combined with per-column operands
    if (g.N % 4 != 0)
        FS2_REQUIRE(!g.b_kmajor && !g.bias && !g.residual && !g.relu_mask && !g.colstats && !g.accumulate,
                    "fs2_gemm: N=%d not a multiple of 4 only for plain row-major-B products", g.N);

    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const long zdim = (long)g.batch1 * g.batch2 * g.split_k;
    FS2_REQUIRE(zdim <= 65535, "fs2_gemm: batch1*batch2*split_k = %ld exceeds 65535", zdim);
    dim3 grid((unsigned)tiles, (unsigned)zdim, 1);
    hipStream_t st = (hipStream_t)stream;
    if (g.dtype == FS2_BF16) {
        if (g.c_dtype == FS2_F32) return launch<bf16_t, float>(g, grid, st);
        return launch<bf16_t, bf16_t>(g, grid, st);
    }
    return launch<float, float>(g, grid, st);
}

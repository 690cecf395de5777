// Gather / scatter / scan / reduction / optimizer kernels of the FastSpeech2 path for gfx950:
// embedding, length regulator (wave scan + gather, segmented-sum backward), bucketize + embedding add,
// L1 losses, weight-shadow cast/permute, global grad-norm and fused clip + Adam over flat fp32 arenas.
#include "fs2_common.h"

namespace {

constexpr int TPB = 256;

#define CHECK_DT(name, dt) FS2_REQUIRE((dt) == FS2_F32 || (dt) == FS2_BF16, "%s: bad dtype %d", name, (int)(dt))
#define T_DISPATCH(dtype, T, ...)                                     \
    do {                                                              \
        if ((dtype) == FS2_F32) { typedef float T; __VA_ARGS__; }     \
        else { typedef bf16_t T; __VA_ARGS__; }                       \
    } while (0)

static inline int wave_rows_grid(int64_t rows) {  // 4 waves per block, one row per wave at a time
    int64_t b = (rows + 3) / 4;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
static inline int flat_grid(int64_t n_vec) {
    int64_t b = (n_vec + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

__device__ __forceinline__ float block_sum(float v, float* lds4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds4[wave] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// ------------------------------------------------------------------ embedding
template <typename T>
__global__ __launch_bounds__(TPB) void embedding_fwd_k(const int64_t* __restrict__ ids, const float* __restrict__ table,
        T* __restrict__ out, int64_t n, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const float* src = table + ids[r] * d;
        for (int c = lane * 4; c < d; c += 256) store4<T>(out + r * d + c, load4<float>(src + c));
    }
}
template <typename T>
__global__ __launch_bounds__(TPB) void embedding_bwd_k(const int64_t* __restrict__ ids, const T* __restrict__ dout,
        float* __restrict__ dtable, int64_t n, int d, int64_t padding_idx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const int64_t id = ids[r];
        if (id == padding_idx) continue;
        float* dst = dtable + id * d;
        for (int c = lane * 4; c < d; c += 256) {
            const float4 g = load4<T>(dout + r * d + c);
            atomicAdd(dst + c, g.x); atomicAdd(dst + c + 1, g.y); atomicAdd(dst + c + 2, g.z); atomicAdd(dst + c + 3, g.w);
        }
    }
}

// ------------------------------------------------------------------ length regulator
// one wave per utterance: exclusive prefix sums of max(dur, 0) -> starts[b][0..L]
__global__ __launch_bounds__(64) void lr_scan_k(const int64_t* __restrict__ dur, int32_t* __restrict__ starts, int L) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < L; base += 64) {
        const int i = base + lane;
        int v = 0;
        if (i < L) { const int64_t x = dur[(int64_t)b * L + i]; v = x > 0 ? (int)x : 0; }
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (i < L) starts[(int64_t)b * (L + 1) + i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) starts[(int64_t)b * (L + 1) + L] = carry;
}

template <typename T>
__global__ __launch_bounds__(TPB) void lr_gather_k(const T* __restrict__ x, const int32_t* __restrict__ starts,
        T* __restrict__ out, int B, int L, int Tn, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rows = (int64_t)B * Tn;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const int b = (int)(r / Tn), f = (int)(r - (int64_t)b * Tn);
        const int32_t* st = starts + (int64_t)b * (L + 1);
        T* dst = out + r * d;
        if (f >= st[L]) {   // beyond the utterance: zero padding
            for (int c = lane * 4; c < d; c += 256) store4<T>(dst + c, make_float4(0.f, 0.f, 0.f, 0.f));
            continue;
        }
        // largest i with st[i] <= f  (phonemes of zero duration are skipped automatically)
        int lo = 0, hi = L;   // invariant: st[lo] <= f < st[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (st[mid] <= f) lo = mid; else hi = mid;
        }
        const T* src = x + ((int64_t)b * L + lo) * d;
        for (int c = lane * 4; c < d; c += 256) store4<T>(dst + c, load4<T>(src + c));
    }
}

// scan + gather in one launch (L <= LR_LMAX): block (utterance b, chunk of LR_FRAMES output frames) repeats the utterance's prefix
// sums into LDS with its first wave (two wave scans at L = 128; the first chunk's block also stores them for the backward pass) and
// gathers its frames with the binary search running on the LDS copy.
constexpr int LR_LMAX = 2048, LR_FRAMES = 32;
template <typename T>
__global__ __launch_bounds__(TPB) void lr_scan_gather_k(const T* __restrict__ x, const int64_t* __restrict__ dur, int32_t* __restrict__ starts,
        T* __restrict__ out, int L, int Tn, int d, int nchunk) {
    __shared__ int st[LR_LMAX + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nchunk, ch = blockIdx.x - b * nchunk;
    if (wave == 0) {
        int carry = 0;
        for (int base = 0; base < L; base += 64) {
            const int i = base + lane;
            int v = 0;
            if (i < L) { const int64_t xv = dur[(int64_t)b * L + i]; v = xv > 0 ? (int)xv : 0; }
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(inc, o, 64);
                if (lane >= o) inc += up;
            }
            if (i < L) st[i] = carry + inc - v;
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) st[L] = carry;
    }
    __syncthreads();
    if (ch == 0)
        for (int i = threadIdx.x; i <= L; i += TPB) starts[(int64_t)b * (L + 1) + i] = st[i];
    const int f_end = min(Tn, (ch + 1) * LR_FRAMES);
    for (int f = ch * LR_FRAMES + wave; f < f_end; f += 4) {
        T* dst = out + ((int64_t)b * Tn + f) * d;
        if (f >= st[L]) {   // beyond the utterance: zero padding
            for (int c = lane * 4; c < d; c += 256) store4<T>(dst + c, make_float4(0.f, 0.f, 0.f, 0.f));
            continue;
        }
        int lo = 0, hi = L;   // invariant: st[lo] <= f < st[hi]  (phonemes of zero duration are skipped automatically)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (st[mid] <= f) lo = mid; else hi = mid;
        }
        const T* src = x + ((int64_t)b * L + lo) * d;
        for (int c = lane * 4; c < d; c += 256) store4<T>(dst + c, load4<T>(src + c));
    }
}

template <typename T>
__global__ __launch_bounds__(TPB) void lr_bwd_k(const T* __restrict__ dout, const int32_t* __restrict__ starts,
        T* __restrict__ dx, int B, int L, int Tn, int d, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rows = (int64_t)B * L;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const int b = (int)(r / L), i = (int)(r - (int64_t)b * L);
        const int32_t* st = starts + (int64_t)b * (L + 1);
        const int f0 = st[i];
        const int f1 = min(st[i + 1], Tn);   // frames cropped by max_len get no gradient
        for (int c = lane * 4; c < d; c += 256) {
            float4 acc = accumulate ? load4<T>(dx + r * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int f = f0; f < f1; ++f) {
                const float4 g = load4<T>(dout + ((int64_t)b * Tn + f) * d + c);
                acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
            }
            store4<T>(dx + r * d + c, acc);
        }
    }
}

// ------------------------------------------------------------------ bucketize + embedding add
__device__ __forceinline__ int bucketize(const float* __restrict__ bins, int nb, float v) {
    // number of boundaries strictly below v (torch.bucketize, right=False)
    int lo = 0, hi = nb;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bins[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <typename T>
__global__ __launch_bounds__(TPB) void bucket_embed_add_fwd_k(const T* __restrict__ x, const float* __restrict__ f0,
        const float* __restrict__ en, const float* __restrict__ pbins, const float* __restrict__ ebins, int nb,
        const float* __restrict__ Ep, const float* __restrict__ Ee, T* __restrict__ out, int32_t* __restrict__ idx,
        int64_t M, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // nb <= 256 boundaries: every lane keeps four of each table in registers and a row's bucket = the number of boundaries
    // below its value is four wave ballots -- the binary search cost 2 x 8 dependent loads per row (the same index for sorted
    // boundaries: torch.bucketize, right=False)
    const bool in_regs = nb <= 256;
    const bool has_p = f0 != nullptr, has_e = en != nullptr;      // hp.pitch_pred / hp.energy_pred False: that term does not exist
    float pb[4], eb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = 4 * lane + c;
        pb[c] = (in_regs && has_p && i < nb) ? pbins[i] : __builtin_huge_valf();
        eb[c] = (in_regs && has_e && i < nb) ? ebins[i] : __builtin_huge_valf();
    }
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < M; r += (int64_t)gridDim.x * 4) {
        int ip = 0, ie = 0;
        if (in_regs) {
            const float vp = has_p ? f0[r] : 0.f, ve = has_e ? en[r] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ip += __popcll(__ballot(pb[c] < vp));
                ie += __popcll(__ballot(eb[c] < ve));
            }
        } else {
            if (has_p) ip = bucketize(pbins, nb, f0[r]);
            if (has_e) ie = bucketize(ebins, nb, en[r]);
        }
        if (lane == 0) { idx[r] = has_p ? ip : -1; idx[M + r] = has_e ? ie : -1; }
        const float* ep = has_p ? Ep + (int64_t)ip * d : nullptr;
        const float* ee = has_e ? Ee + (int64_t)ie * d : nullptr;
        for (int c = lane * 4; c < d; c += 256) {
            float4 v = load4<T>(x + r * d + c);
            // same association as the reference: (x + pitch_embedding) + energy_embedding
            if (has_p) { const float4 a = load4<float>(ep + c); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
            if (has_e) { const float4 b = load4<float>(ee + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            store4<T>(out + r * d + c, v);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(TPB) void bucket_embed_bwd_k(const T* __restrict__ dout, const int32_t* __restrict__ idx,
        float* __restrict__ dEp, float* __restrict__ dEe, int64_t M, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < M; r += (int64_t)gridDim.x * 4) {
        float* dp = (dEp != nullptr && idx[r] >= 0) ? dEp + (int64_t)idx[r] * d : nullptr;
        float* de = (dEe != nullptr && idx[M + r] >= 0) ? dEe + (int64_t)idx[M + r] * d : nullptr;
        for (int c = lane * 4; c < d; c += 256) {
            const float4 g = load4<T>(dout + r * d + c);
            if (dp) { atomicAdd(dp + c, g.x); atomicAdd(dp + c + 1, g.y); atomicAdd(dp + c + 2, g.z); atomicAdd(dp + c + 3, g.w); }
            if (de) { atomicAdd(de + c, g.x); atomicAdd(de + c + 1, g.y); atomicAdd(de + c + 2, g.z); atomicAdd(de + c + 3, g.w); }
        }
    }
}

// ------------------------------------------------------------------ L1 loss
template <typename T> __device__ __forceinline__ float l1_target(const void* tgt, int mode, int64_t i) {
    if (mode == 1) return logf((float)reinterpret_cast<const int64_t*>(tgt)[i] + 1.0f);
    return reinterpret_cast<const float*>(tgt)[i];
}
template <typename T>
__global__ __launch_bounds__(TPB) void l1_fwd_k(const T* __restrict__ pred, const void* __restrict__ tgt, int mode,
        int64_t n, float* __restrict__ loss) {
    __shared__ float lds4[4];
    float acc = 0.f;
    int64_t done = 0;
    if constexpr (sizeof(T) == 4) {       // fp32 predictions against fp32 targets: 16-byte loads
        if (mode == 0 && ((((uintptr_t)pred) | ((uintptr_t)tgt)) & 15) == 0) {
            const int64_t n4 = n >> 2;
            const float4* p4 = reinterpret_cast<const float4*>(pred);
            const float4* t4 = reinterpret_cast<const float4*>(tgt);
            for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
                const float4 a = p4[i], b = t4[i];
                acc += (fabsf(a.x - b.x) + fabsf(a.y - b.y)) + (fabsf(a.z - b.z) + fabsf(a.w - b.w));
            }
            done = n4 << 2;
        }
    }
    for (int64_t i = done + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        acc += fabsf(to_f32<T>(pred[i]) - l1_target<T>(tgt, mode, i));
    const float s = block_sum(acc, lds4);
    if (threadIdx.x == 0) atomicAdd(loss, s / (float)n);      // one address: the launcher keeps the grid at <= 256 blocks
}
template <typename T, typename TG>
__global__ __launch_bounds__(TPB) void l1_bwd_k(const T* __restrict__ pred, const void* __restrict__ tgt, int mode,
        int64_t n, const float* __restrict__ gscale, TG* __restrict__ dpred) {
    const float g = gscale[0] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float df = to_f32<T>(pred[i]) - l1_target<T>(tgt, mode, i);
        dpred[i] = from_f32<TG>(df > 0.f ? g : (df < 0.f ? -g : 0.f));
    }
}

// The loss terms of a step in ONE launch each way (the trainer sums five nn.L1Loss() terms: five forward kernels, five
// memsets, four adds and five backward kernels of a few microseconds each otherwise).  blockIdx.y = term.
struct L1Items { FS2L1Item it[8]; };
template <typename T>
__device__ __forceinline__ float l1_partial(const T* pred, const void* tgt, int mode, int64_t n) {
    float acc = 0.f;
    int64_t done = 0;
    if constexpr (sizeof(T) == 4) {
        if (mode == 0 && ((((uintptr_t)pred) | ((uintptr_t)tgt)) & 15) == 0) {
            const int64_t n4 = n >> 2;
            const float4* p4 = reinterpret_cast<const float4*>(pred);
            const float4* t4 = reinterpret_cast<const float4*>(tgt);
            for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
                const float4 a = p4[i], b = t4[i];
                acc += (fabsf(a.x - b.x) + fabsf(a.y - b.y)) + (fabsf(a.z - b.z) + fabsf(a.w - b.w));
            }
            done = n4 << 2;
        }
    }
    for (int64_t i = done + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        acc += fabsf(to_f32<T>(pred[i]) - l1_target<T>(tgt, mode, i));
    return acc;
}
// Two stages, no float atomics: block (x, term) of the first launch leaves its partial sum in workspace[term * L1_BLOCKS + x]; the one
// block of the second launch adds every term's partials in a fixed order and STORES losses[term] = mean and losses[n_items] = the sum
// of the terms in their order (the trainer's total: no zero fill in front, no torch reduction behind, the same bits on every run).
// (One launch with a last-block ticket was measured first: 53-60 us against 16 -- an agent-scope fence writes back the XCD's L2, and
//  1280 blocks issue one each.  The kernel boundary is the cheap fence.)
constexpr int L1_BLOCKS = 256, L1_WS = 8 * L1_BLOCKS;
__global__ __launch_bounds__(TPB) void l1_multi_fwd_k(const L1Items items, float* __restrict__ workspace) {
    __shared__ float lds4[4];
    const FS2L1Item it = items.it[blockIdx.y];
    const float acc = it.pred_dtype == FS2_F32 ? l1_partial<float>(reinterpret_cast<const float*>(it.pred), it.target, it.target_mode, it.n)
                                               : l1_partial<bf16_t>(reinterpret_cast<const bf16_t*>(it.pred), it.target, it.target_mode, it.n);
    const float s = block_sum(acc, lds4);
    if (threadIdx.x == 0) workspace[blockIdx.y * L1_BLOCKS + blockIdx.x] = s;
}
struct L1Counts { int64_t n[8]; };
__global__ __launch_bounds__(TPB) void l1_multi_finish_k(const L1Counts counts, int n_items, const float* __restrict__ workspace, float* __restrict__ losses) {
    __shared__ float lds4[4];
    float total = 0.f;
    for (int t = 0; t < n_items; ++t) {        // (TPB = L1_BLOCKS threads: one partial each, block_sum adds them in a fixed tree)
        const float mean = block_sum(workspace[t * L1_BLOCKS + threadIdx.x], lds4) / (float)counts.n[t];
        if (threadIdx.x == 0) losses[t] = mean;
        total += mean;                          // (the reference's order of terms)
    }
    if (threadIdx.x == 0) losses[n_items] = total;
}
template <typename T, typename TG>
__device__ __forceinline__ void l1_grad(const T* pred, const void* tgt, int mode, int64_t n, float g, TG* dpred) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float df = to_f32<T>(pred[i]) - l1_target<T>(tgt, mode, i);
        dpred[i] = from_f32<TG>(df > 0.f ? g : (df < 0.f ? -g : 0.f));
    }
}
__global__ __launch_bounds__(TPB) void l1_multi_bwd_k(const L1Items items, const float* __restrict__ gscale) {
    const FS2L1Item it = items.it[blockIdx.y];
    const float g = gscale[0] / (float)it.n;
    const bool pf = it.pred_dtype == FS2_F32, gf = it.dpred_dtype == FS2_F32;
    if (pf && gf) l1_grad<float, float>((const float*)it.pred, it.target, it.target_mode, it.n, g, (float*)it.dpred);
    else if (pf) l1_grad<float, bf16_t>((const float*)it.pred, it.target, it.target_mode, it.n, g, (bf16_t*)it.dpred);
    else if (gf) l1_grad<bf16_t, float>((const bf16_t*)it.pred, it.target, it.target_mode, it.n, g, (float*)it.dpred);
    else l1_grad<bf16_t, bf16_t>((const bf16_t*)it.pred, it.target, it.target_mode, it.n, g, (bf16_t*)it.dpred);
}

// ------------------------------------------------------------------ stop-token loss of the autoregressive model
// F.binary_cross_entropy_with_logits(x, y, reduction='mean', pos_weight=pw) (reference train.py:217):
//   l = (1 - y) x + (1 + (pw - 1) y) softplus(-x),  softplus(-x) = log1p(exp(-|x|)) + max(-x, 0)
//   dl/dx = (1 - y) - (1 + (pw - 1) y) sigmoid(-x)
template <typename T>
__global__ __launch_bounds__(TPB) void bce_fwd_k(const T* __restrict__ x, const float* __restrict__ y, int64_t n, float pw,
        float* __restrict__ loss) {
    __shared__ float lds4[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float xv = to_f32<T>(x[i]), yv = y[i];
        const float sp = log1pf(expf(-fabsf(xv))) + fmaxf(-xv, 0.f);
        acc += (1.f - yv) * xv + (1.f + (pw - 1.f) * yv) * sp;
    }
    const float s = block_sum(acc, lds4);
    if (threadIdx.x == 0) atomicAdd(loss, s / (float)n);
}
template <typename T, typename TG>
__global__ __launch_bounds__(TPB) void bce_bwd_k(const T* __restrict__ x, const float* __restrict__ y, int64_t n, float pw,
        const float* __restrict__ gscale, TG* __restrict__ dx) {
    const float g = gscale[0] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float xv = to_f32<T>(x[i]), yv = y[i];
        const float sig_neg = 1.f / (1.f + expf(xv));        // sigmoid(-x)
        dx[i] = from_f32<TG>(g * ((1.f - yv) - (1.f + (pw - 1.f) * yv) * sig_neg));
    }
}

// ------------------------------------------------------------------ plain dropout (decoder pre-net of the autoregressive model)
// out = x * keep_scale(mask), optionally zeroed where gate <= 0 (backward through a ReLU whose output is `gate`).
// Same Philox stream layout as every other dropout of the library: one call per 8 consecutive elements, counter = e >> 3.
template <typename T>
__global__ __launch_bounds__(TPB) void dropout_k(const T* __restrict__ x, const T* __restrict__ gate, T* __restrict__ out,
        int64_t n, float p, const uint64_t* rng, uint32_t site) {
    const DropCtx dc = drop_ctx(rng, site, p);
    const int64_t n8 = (n + 7) >> 3;
    for (int64_t q = (int64_t)blockIdx.x * TPB + threadIdx.x; q < n8; q += (int64_t)gridDim.x * TPB) {
        float sc[8];
        drop_scale8(dc, (uint64_t)q, sc);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int64_t e = q * 8 + c;
            if (e < n) {
                float v = to_f32<T>(x[e]) * sc[c];
                if (gate != nullptr && !(to_f32<T>(gate[e]) > 0.f)) v = 0.f;
                out[e] = from_f32<T>(v);
            }
        }
    }
}

// ------------------------------------------------------------------ casts / weight shadows
template <typename TS, typename TD>
__global__ __launch_bounds__(TPB) void cast_k(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        dst[i] = from_f32<TD>(to_f32<TS>(src[i]));
}
// dst = a + b (fp32 operands, four elements per thread; the post-net's two gradient terms of mel_pred become ONE operand of the
// output Linear's backward products: Models/postnets.py:67,74-75)
template <typename TD>
__global__ __launch_bounds__(TPB) void add_cast_k(const float* __restrict__ a, const float* __restrict__ b, TD* __restrict__ dst, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        dst[4 * i] = from_f32<TD>(x.x + y.x); dst[4 * i + 1] = from_f32<TD>(x.y + y.y);
        dst[4 * i + 2] = from_f32<TD>(x.z + y.z); dst[4 * i + 3] = from_f32<TD>(x.w + y.w);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        dst[i] = from_f32<TD>(a[i] + b[i]);
}
// src (O, I, k) fp32 -> dst; one thread per destination element (destination-contiguous)
template <typename T>
__global__ __launch_bounds__(TPB) void cast_permute_k(const float* __restrict__ src, T* __restrict__ dst, int O, int I,
        int k, int64_t dld, int mode) {
    const int64_t n = (int64_t)O * I * k;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        if (mode == 0) {          // dst[o][j*I + i] = src[o][i][j]
            const int i = (int)(e % I); const int j = (int)((e / I) % k); const int o = (int)(e / ((int64_t)I * k));
            dst[(int64_t)o * dld + (int64_t)j * I + i] = from_f32<T>(src[((int64_t)o * I + i) * k + j]);
        } else {                  // dst[i][j*O + o] = src[o][i][k-1-j]
            const int o = (int)(e % O); const int j = (int)((e / O) % k); const int i = (int)(e / ((int64_t)O * k));
            dst[(int64_t)i * dld + (int64_t)j * O + o] = from_f32<T>(src[((int64_t)o * I + i) * k + (k - 1 - j)]);
        }
    }
}
template <typename T> __device__ __forceinline__ void store2(T* p, float a, float b);
template <> __device__ __forceinline__ void store2<float>(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
template <> __device__ __forceinline__ void store2<bf16_t>(bf16_t* p, float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v; v[0] = (bf16_t)a; v[1] = (bf16_t)b;
    *reinterpret_cast<bf16x2_t*>(p) = v;
}
// tiles of one descriptor with the tap count as a compile-time constant (the index arithmetic divides by k per element:
// with a run-time k those divisions, not the memory system, set the kernel's time)
template <typename T, int KC, int MODE>
__device__ __forceinline__ void cast_tiles(const FS2CastDesc& d, float* tile) {
    const int O = d.O, I = d.I;
    constexpr int k = KC;
    const float* __restrict__ src = d.src;
    // tile = TO_ o x TI_ i x k source elements: the long side is the one the destination runs along (mode 0: 64 consecutive i =
    // 128-byte runs of bf16; mode 1: 64 consecutive o), the source is read in contiguous runs of TI_*k floats per o
    constexpr int TO_ = MODE == 0 ? 16 : 64, TI_ = MODE == 0 ? 64 : 16;
    const int to = (O + TO_ - 1) / TO_, ti = (I + TI_ - 1) / TI_;
    constexpr int run = TI_ * k, ldt = run + 1;            // +1: the transposed reads below hit distinct banks
    for (int t = blockIdx.x; t < to * ti; t += gridDim.x) {
        const int o0 = (t / ti) * TO_, i0 = (t % ti) * TI_;
        __syncthreads();
        if (i0 + TI_ <= I && ((int64_t)I * k) % 4 == 0 && (((uintptr_t)src) & 15) == 0) {       // whole runs: 16-byte loads
            constexpr int run4 = run / 4;
            for (int e = threadIdx.x; e < TO_ * run4; e += TPB) {
                const int oo = e / run4, r = (e % run4) * 4;
                const int o = o0 + oo;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (o < O) v = *reinterpret_cast<const float4*>(src + ((int64_t)o * I + i0) * k + r);
                float* tp = tile + oo * ldt + r;
                tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
            }
        } else {
            for (int e = threadIdx.x; e < TO_ * run; e += TPB) {
                const int oo = e / run, r = e % run;           // r = ii*k + j
                const int o = o0 + oo, i = i0 + r / k;
                tile[oo * ldt + r] = (o < O && i < I) ? src[((int64_t)o * I + i0) * k + r] : 0.f;
            }
        }
        __syncthreads();
        // two consecutive destination elements per thread (one 4-byte store for bf16)
        for (int e = threadIdx.x; e < TO_ * run / 2; e += TPB) {
            if (MODE == 0) {            // dst[o][j*I + i]: 64 consecutive i per (o, j)
                const int ii = (e % 32) * 2, j = (e / 32) % k, oo = e / (32 * k);
                const int o = o0 + oo, i = i0 + ii;
                if (o < O && i < I) {
                    T* dp = reinterpret_cast<T*>(d.dst) + (int64_t)o * d.dld + (int64_t)j * I + i;
                    const float a = tile[oo * ldt + ii * k + j], b = tile[oo * ldt + (ii + 1) * k + j];
                    if (i + 1 < I && (((uintptr_t)dp) & (2 * sizeof(T) - 1)) == 0) store2<T>(dp, a, b);
                    else { dp[0] = from_f32<T>(a); if (i + 1 < I) dp[1] = from_f32<T>(b); }
                }
            } else {                    // dst[i][j*O + o]: 64 consecutive o per (i, j), taps flipped
                const int oo = (e % 32) * 2, j = (e / 32) % k, ii = e / (32 * k);
                const int o = o0 + oo, i = i0 + ii;
                if (o < O && i < I) {
                    T* dp = reinterpret_cast<T*>(d.dst) + (int64_t)i * d.dld + (int64_t)j * O + o;
                    const float a = tile[oo * ldt + ii * k + (k - 1 - j)], b = tile[(oo + 1) * ldt + ii * k + (k - 1 - j)];
                    if (o + 1 < O && (((uintptr_t)dp) & (2 * sizeof(T) - 1)) == 0) store2<T>(dp, a, b);
                    else { dp[0] = from_f32<T>(a); if (o + 1 < O) dp[1] = from_f32<T>(b); }
                }
            }
        }
    }
}

// One block iteration = a tile of 16 o x 64 i x k (forward shadow) or 64 o x 16 i x k (data-gradient shadow) source elements staged in LDS: the source is read in contiguous
// runs of 32*k floats per o, the shadows are written in runs of 32 contiguous elements ([o][j*I + i] or [i][j*O + o]).
template <typename T>
__global__ __launch_bounds__(TPB) void cast_permute_batched_k(const FS2CastDesc* __restrict__ table) {
    __shared__ float tile[64 * (16 * 9 + 1) > 16 * (64 * 9 + 1) ? 64 * (16 * 9 + 1) : 16 * (64 * 9 + 1)];
    const FS2CastDesc d = table[blockIdx.y];
    const int O = d.O, I = d.I, k = d.k;
    const float* __restrict__ src = d.src;
    if (d.mode == 2) {
        const int64_t n = (int64_t)O * I * k;
        for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB)
            reinterpret_cast<float*>(d.dst)[e] = src[e];
        return;
    }
    if (k > 9) {    // generic fallback (no conv of this model has more than 9 taps)
        const int64_t n = (int64_t)O * I * k;
        for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
            if (d.mode == 0) {
                const int i = (int)(e % I); const int j = (int)((e / I) % k); const int o = (int)(e / ((int64_t)I * k));
                reinterpret_cast<T*>(d.dst)[(int64_t)o * d.dld + (int64_t)j * I + i] = from_f32<T>(src[((int64_t)o * I + i) * k + j]);
            } else {
                const int o = (int)(e % O); const int j = (int)((e / O) % k); const int i = (int)(e / ((int64_t)O * k));
                reinterpret_cast<T*>(d.dst)[(int64_t)i * d.dld + (int64_t)j * O + o] = from_f32<T>(src[((int64_t)o * I + i) * k + (k - 1 - j)]);
            }
        }
        return;
    }
    switch (k) {
        case 1: if (d.mode == 0) cast_tiles<T, 1, 0>(d, tile); else cast_tiles<T, 1, 1>(d, tile); break;
        case 2: if (d.mode == 0) cast_tiles<T, 2, 0>(d, tile); else cast_tiles<T, 2, 1>(d, tile); break;
        case 3: if (d.mode == 0) cast_tiles<T, 3, 0>(d, tile); else cast_tiles<T, 3, 1>(d, tile); break;
        case 4: if (d.mode == 0) cast_tiles<T, 4, 0>(d, tile); else cast_tiles<T, 4, 1>(d, tile); break;
        case 5: if (d.mode == 0) cast_tiles<T, 5, 0>(d, tile); else cast_tiles<T, 5, 1>(d, tile); break;
        case 6: if (d.mode == 0) cast_tiles<T, 6, 0>(d, tile); else cast_tiles<T, 6, 1>(d, tile); break;
        case 7: if (d.mode == 0) cast_tiles<T, 7, 0>(d, tile); else cast_tiles<T, 7, 1>(d, tile); break;
        case 8: if (d.mode == 0) cast_tiles<T, 8, 0>(d, tile); else cast_tiles<T, 8, 1>(d, tile); break;
        default: if (d.mode == 0) cast_tiles<T, 9, 0>(d, tile); else cast_tiles<T, 9, 1>(d, tile); break;
    }
}
template <typename T>
__global__ __launch_bounds__(TPB) void onehot_k(const int32_t* __restrict__ idx, T* __restrict__ out, int64_t M, int nb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < M; r += (int64_t)gridDim.x * 4) {
        const int hot = idx[r];
        for (int c = lane * 4; c < nb; c += 256)
            store4<T>(out + r * nb + c, make_float4(c == hot ? 1.f : 0.f, c + 1 == hot ? 1.f : 0.f, c + 2 == hot ? 1.f : 0.f,
                                                     c + 3 == hot ? 1.f : 0.f));
    }
}
__global__ __launch_bounds__(TPB) void permute_add_k(float* __restrict__ scratch, float* __restrict__ grad, int O,
        int I, int k, int rezero) {
    const int64_t n = (int64_t)O * I * k;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int j = (int)(e % k); const int i = (int)((e / k) % I); const int o = (int)(e / ((int64_t)I * k));
        const int64_t si = ((int64_t)o * k + j) * I + i;
        grad[e] += scratch[si];
        if (rezero) scratch[si] = 0.f;     // each scratch element is read exactly once: leave it zeroed for the next user
    }
}

// The same through LDS tiles (k <= 9): for 4 output channels and 64 input channels at a time the k source runs of 64 floats
// ([o][j][i0 .. i0+63], 256 B each) are read contiguously and the 64*k destination floats of each o ([o][i0 .. i0+63][0..k)) are
// written contiguously -- the flat kernel above reads the scratch at a stride of I floats.
__global__ __launch_bounds__(TPB) void permute_add_tiled_k(float* __restrict__ scratch, float* __restrict__ grad, int O, int I, int k,
        int rezero) {
    __shared__ float tile[4][9][65];
    const int nib = (I + 63) / 64, items = ((O + 3) / 4) * nib;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int o0 = (item / nib) * 4, i0 = (item % nib) * 64;
        for (int idx = threadIdx.x; idx < 4 * k * 64; idx += TPB) {
            const int ii = idx & 63, j = (idx >> 6) % k, oo = idx / (64 * k);
            const int o = o0 + oo, i = i0 + ii;
            if (o < O && i < I) {
                const int64_t si = ((int64_t)o * k + j) * I + i;
                tile[oo][j][ii] = scratch[si];
                if (rezero) scratch[si] = 0.f;
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 4 * 64 * k; idx += TPB) {
            const int e = idx % (64 * k), oo = idx / (64 * k);
            const int ii = e / k, j = e - ii * k;
            const int o = o0 + oo, i = i0 + ii;
            if (o < O && i < I) grad[((int64_t)o * I + i) * k + j] += tile[oo][j][ii];
        }
        __syncthreads();
    }
}

// out[n] += sum_m x[m][n]: block = 64 column-groups of 4 x 4 row-slabs.  Segmented output: column c lands at
// out[(c / seg_cols) * seg_stride + c % seg_cols] (the q/v/k bias gradients of one fused projection sit at a constant
// stride in the gradient arena); seg_cols >= N is the plain vector.
template <typename T>
__global__ __launch_bounds__(TPB) void colsum_k(const T* __restrict__ x, int64_t M, int N, int64_t ldx,
        float* __restrict__ out, int seg_cols, int64_t seg_stride) {
    __shared__ __attribute__((aligned(16))) float red[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < N) {
        const int64_t stride = (int64_t)gridDim.y * 4;
        int64_t r = (int64_t)blockIdx.y * 4 + wave;
        for (; r + 3 * stride < M; r += 4 * stride) {       // 4 independent row loads in flight
            const float4 v0 = load4<T>(x + r * ldx + col), v1 = load4<T>(x + (r + stride) * ldx + col);
            const float4 v2 = load4<T>(x + (r + 2 * stride) * ldx + col), v3 = load4<T>(x + (r + 3 * stride) * ldx + col);
            acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
            acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; r < M; r += stride) {
            const float4 v = load4<T>(x + r * ldx + col);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(red + wave * 256 + lane * 4) = acc;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) atomicAdd(out + (int64_t)(c / seg_cols) * seg_stride + c % seg_cols,
                         red[threadIdx.x] + red[256 + threadIdx.x] + red[512 + threadIdx.x] + red[768 + threadIdx.x]);
}

// The same for NARROW matrices (N <= 128: the 80 mel channels): a wave of colsum_k has only N/4 of its 64 lanes on a row, so a wave
// here takes 64 / (N/4) consecutive rows per step (lane -> row lane / (N/4), column group lane % (N/4)): three times the bytes in
// flight at N = 80; the sub-rows of a wave are added up with the waves at the end.
template <typename T>
__global__ __launch_bounds__(TPB) void colsum_narrow_k(const T* __restrict__ x, int64_t M, int N, int64_t ldx,
        float* __restrict__ out, int seg_cols, int64_t seg_stride) {
    __shared__ __attribute__((aligned(16))) float red[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = N >> 2, rpw = 64 / cg;
    const int sub = lane / cg, col = (lane - sub * cg) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub < rpw) {
        const int64_t stride = (int64_t)gridDim.y * 4 * rpw;
        int64_t r = ((int64_t)blockIdx.y * 4 + wave) * rpw + sub;
        for (; r + 7 * stride < M; r += 8 * stride) {       // 8 independent row loads in flight
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load4<T>(x + (r + u * stride) * ldx + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; r < M; r += stride) {
            const float4 v = load4<T>(x + r * ldx + col);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(red + wave * 256 + lane * 4) = acc;
    __syncthreads();
    const int c = threadIdx.x;
    if (c < N) {
        float t = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int sr = 0; sr < rpw; ++sr) t += red[w * 256 + (sr * cg) * 4 + c];
        atomicAdd(out + (int64_t)(c / seg_cols) * seg_stride + c % seg_cols, t);
    }
}

// ------------------------------------------------------------------ optimizer
__global__ __launch_bounds__(TPB) void sqnorm_k(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float lds4[4];
    float acc = 0.f;
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    for (; i + stride < nv; i += 2 * stride) {          // two independent loads in flight per thread
        const float4 v = reinterpret_cast<const float4*>(x)[i], w = reinterpret_cast<const float4*>(x)[i + stride];
        acc += ((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) + ((w.x * w.x + w.y * w.y) + (w.z * w.z + w.w * w.w));
    }
    if (i < nv) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[(nv << 2) + threadIdx.x]; acc += v * v; }
    const float s = block_sum(acc, lds4);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float gmul, float b1, float b2,
                                       float step_size, float bc2s, float eps) {
    g *= gmul;
    m = m + (g - m) * (1.f - b1);            // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + (1.f - b2) * g * g;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / bc2s + eps;
    p = p - step_size * (m / denom);
}
// PERM: some parameter ranges (Conv1d weights, (O,I,k) in the parameter arena) keep their GRADIENT in the layout the weight-gradient
// GEMM produces, [o][j][i]: the optimizer is the only reader that cares, so it gathers instead of a separate permute pass over
// every convolution gradient (0.19 ms per step).  segs: sorted, disjoint {start, end, O, I, k} (int64 x 5), starts multiples of 4.
template <bool PERM>
__global__ __launch_bounds__(TPB) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, int64_t n, const float* __restrict__ hyper, const float* __restrict__ gsq, float b1,
        float b2, float eps, float max_norm, const int64_t* __restrict__ segs, int nseg) {
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], gs = hyper[3];
    float gmul = gs;
    if (max_norm > 0.f && gsq != nullptr) {
        const float total = sqrtf(gsq[0]) * gs;
        gmul *= fminf(1.f, max_norm / (total + 1e-6f));   // clip_grad_norm_
    }
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * TPB) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float4 gg;
        bool gathered = false;
        if constexpr (PERM) {
            const int64_t e0 = i << 2;
            for (int sgi = 0; sgi < nseg; ++sgi) {
                const int64_t start = segs[5 * sgi];
                if (e0 < start) break;
                if (e0 < segs[5 * sgi + 1]) {
                    const int I = (int)segs[5 * sgi + 3], k = (int)segs[5 * sgi + 4];
                    const int64_t e = e0 - start;
                    const int64_t o = e / ((int64_t)I * k);
                    int rem = (int)(e - o * (int64_t)I * k);
                    int ii = rem / k, j = rem - ii * k;
                    const float* gb = g + start + o * (int64_t)I * k;
                    float t4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {           // (o, ii, j) -> [o][j][ii]; a group of four may run into the next o
                        t4[c] = (e + c < segs[5 * sgi + 1] - start) ? gb[(int64_t)j * I + ii] : 0.f;
                        if (++j == k) { j = 0; if (++ii == I) { ii = 0; gb += (int64_t)I * k; } }
                    }
                    gg = make_float4(t4[0], t4[1], t4[2], t4[3]);
                    gathered = true;
                    break;
                }
            }
        }
        if (!gathered) gg = reinterpret_cast<const float4*>(g)[i];
        adam1(pp.x, gg.x, mm.x, vv.x, gmul, b1, b2, step_size, bc2s, eps);
        adam1(pp.y, gg.y, mm.y, vv.y, gmul, b1, b2, step_size, bc2s, eps);
        adam1(pp.z, gg.z, mm.z, vv.z, gmul, b1, b2, step_size, bc2s, eps);
        adam1(pp.w, gg.w, mm.w, vv.w, gmul, b1, b2, step_size, bc2s, eps);
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        adam1(p[i], g[i], m[i], v[i], gmul, b1, b2, step_size, bc2s, eps);
    }
}

__global__ void rng_advance_k(uint64_t* rng) { rng[1] += 1; }

}  // namespace

// ==================================================================== host launchers
extern "C" int fs2_embedding_fwd(const int64_t* ids, const float* table, void* out, int out_dtype, int64_t n, int d,
                                 void* stream) {
    CHECK_DT("fs2_embedding_fwd", out_dtype);
    FS2_REQUIRE(d > 0 && d % 4 == 0, "fs2_embedding_fwd: d=%d must be a multiple of 4", d);
    if (n <= 0) return FS2_OK;
    T_DISPATCH(out_dtype, T, { hipLaunchKernelGGL((embedding_fwd_k<T>), dim3(wave_rows_grid(n)), dim3(TPB), 0, (hipStream_t)stream, ids, table, (T*)out, n, d); });
    FS2_CHECK_LAUNCH("fs2_embedding_fwd");
    return FS2_OK;
}
extern "C" int fs2_embedding_bwd(const int64_t* ids, const void* dout, int dout_dtype, float* dtable, int64_t n, int d,
                                 int64_t padding_idx, void* stream) {
    CHECK_DT("fs2_embedding_bwd", dout_dtype);
    FS2_REQUIRE(d > 0 && d % 4 == 0, "fs2_embedding_bwd: d=%d must be a multiple of 4", d);
    if (n <= 0) return FS2_OK;
    T_DISPATCH(dout_dtype, T, { hipLaunchKernelGGL((embedding_bwd_k<T>), dim3(wave_rows_grid(n)), dim3(TPB), 0, (hipStream_t)stream, ids, (const T*)dout, dtable, n, d, padding_idx); });
    FS2_CHECK_LAUNCH("fs2_embedding_bwd");
    return FS2_OK;
}

extern "C" int fs2_length_regulate_fwd(const void* x, int dtype, const int64_t* dur, void* out, int32_t* starts, int B,
                                       int L, int T, int d, void* stream) {
    CHECK_DT("fs2_length_regulate_fwd", dtype);
    FS2_REQUIRE(B > 0 && L > 0 && T > 0 && d > 0 && d % 4 == 0, "fs2_length_regulate_fwd: bad shape B=%d L=%d T=%d d=%d", B, L, T, d);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = (T + LR_FRAMES - 1) / LR_FRAMES;
    if (L <= LR_LMAX && (int64_t)B * nchunk <= 0x7FFFFFFF) {       // one launch
        T_DISPATCH(dtype, TT, { hipLaunchKernelGGL((lr_scan_gather_k<TT>), dim3((unsigned)(B * nchunk)), dim3(TPB), 0, st, (const TT*)x, dur, starts, (TT*)out, L, T, d, nchunk); });
        FS2_CHECK_LAUNCH("fs2_length_regulate_fwd");
        return FS2_OK;
    }
    hipLaunchKernelGGL(lr_scan_k, dim3(B), dim3(64), 0, st, dur, starts, L);
    FS2_CHECK_LAUNCH("fs2_length_regulate_fwd(scan)");
    T_DISPATCH(dtype, TT, { hipLaunchKernelGGL((lr_gather_k<TT>), dim3(wave_rows_grid((int64_t)B * T)), dim3(TPB), 0, st, (const TT*)x, starts, (TT*)out, B, L, T, d); });
    FS2_CHECK_LAUNCH("fs2_length_regulate_fwd(gather)");
    return FS2_OK;
}
extern "C" int fs2_length_regulate_bwd(const void* dout, int dtype, const int32_t* starts, void* dx, int B, int L, int T,
                                       int d, int accumulate, void* stream) {
    CHECK_DT("fs2_length_regulate_bwd", dtype);
    FS2_REQUIRE(B > 0 && L > 0 && T > 0 && d > 0 && d % 4 == 0, "fs2_length_regulate_bwd: bad shape");
    T_DISPATCH(dtype, TT, { hipLaunchKernelGGL((lr_bwd_k<TT>), dim3(wave_rows_grid((int64_t)B * L)), dim3(TPB), 0, (hipStream_t)stream, (const TT*)dout, starts, (TT*)dx, B, L, T, d, accumulate); });
    FS2_CHECK_LAUNCH("fs2_length_regulate_bwd");
    return FS2_OK;
}

extern "C" int fs2_bucket_embed_add_fwd(const void* x, int dtype, const float* f0, const float* energy, const float* pbins,
                                        const float* ebins, int nbins, const float* Ep, const float* Ee, void* out,
                                        int32_t* idx, int64_t M, int d, void* stream) {
    CHECK_DT("fs2_bucket_embed_add_fwd", dtype);
    FS2_REQUIRE(d > 0 && d % 4 == 0 && nbins > 0, "fs2_bucket_embed_add_fwd: bad d/nbins");
    if (M <= 0) return FS2_OK;
    T_DISPATCH(dtype, T, { hipLaunchKernelGGL((bucket_embed_add_fwd_k<T>), dim3(wave_rows_grid(M)), dim3(TPB), 0, (hipStream_t)stream, (const T*)x, f0, energy, pbins, ebins, nbins, Ep, Ee, (T*)out, idx, M, d); });
    FS2_CHECK_LAUNCH("fs2_bucket_embed_add_fwd");
    return FS2_OK;
}
extern "C" int fs2_bucket_embed_bwd(const void* dout, int dtype, const int32_t* idx, float* dEp, float* dEe, int64_t M,
                                    int d, void* stream) {
    CHECK_DT("fs2_bucket_embed_bwd", dtype);
    FS2_REQUIRE(d > 0 && d % 4 == 0, "fs2_bucket_embed_bwd: bad d");
    if (M <= 0) return FS2_OK;
    T_DISPATCH(dtype, T, { hipLaunchKernelGGL((bucket_embed_bwd_k<T>), dim3(wave_rows_grid(M)), dim3(TPB), 0, (hipStream_t)stream, (const T*)dout, idx, dEp, dEe, M, d); });
    FS2_CHECK_LAUNCH("fs2_bucket_embed_bwd");
    return FS2_OK;
}

extern "C" int fs2_l1_fwd(const void* pred, int pred_dtype, const void* target, int target_mode, int64_t n, float* loss,
                          void* stream) {
    CHECK_DT("fs2_l1_fwd", pred_dtype);
    FS2_REQUIRE(n > 0 && (target_mode == 0 || target_mode == 1), "fs2_l1_fwd: bad n/target_mode");
    const int blocks = flat_grid(n >> 2) < 256 ? (flat_grid(n >> 2) < 1 ? 1 : flat_grid(n >> 2)) : 256;      // every block ends with one atomic on the same address
    T_DISPATCH(pred_dtype, T, { hipLaunchKernelGGL((l1_fwd_k<T>), dim3(blocks), dim3(TPB), 0, (hipStream_t)stream, (const T*)pred, target, target_mode, n, loss); });
    FS2_CHECK_LAUNCH("fs2_l1_fwd");
    return FS2_OK;
}
extern "C" int fs2_l1_bwd(const void* pred, int pred_dtype, const void* target, int target_mode, int64_t n,
                          const float* gscale, void* dpred, int dpred_dtype, void* stream) {
    CHECK_DT("fs2_l1_bwd", pred_dtype); CHECK_DT("fs2_l1_bwd", dpred_dtype);
    FS2_REQUIRE(n > 0 && (target_mode == 0 || target_mode == 1), "fs2_l1_bwd: bad n/target_mode");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(flat_grid(n)), block(TPB);
    if (pred_dtype == FS2_F32 && dpred_dtype == FS2_F32) hipLaunchKernelGGL((l1_bwd_k<float, float>), grid, block, 0, st, (const float*)pred, target, target_mode, n, gscale, (float*)dpred);
    else if (pred_dtype == FS2_F32) hipLaunchKernelGGL((l1_bwd_k<float, bf16_t>), grid, block, 0, st, (const float*)pred, target, target_mode, n, gscale, (bf16_t*)dpred);
    else if (dpred_dtype == FS2_F32) hipLaunchKernelGGL((l1_bwd_k<bf16_t, float>), grid, block, 0, st, (const bf16_t*)pred, target, target_mode, n, gscale, (float*)dpred);
    else hipLaunchKernelGGL((l1_bwd_k<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)pred, target, target_mode, n, gscale, (bf16_t*)dpred);
    FS2_CHECK_LAUNCH("fs2_l1_bwd");
    return FS2_OK;
}

static int l1_items_check(const char* who, const FS2L1Item* items, int n_items, bool bwd, L1Items* out) {
    FS2_REQUIRE(items != nullptr && n_items > 0 && n_items <= 8, "%s: 1..8 items", who);
    for (int i = 0; i < n_items; ++i) {
        const FS2L1Item& it = items[i];
        FS2_REQUIRE(it.pred && it.target && it.n > 0 && (it.target_mode == 0 || it.target_mode == 1), "%s: bad item %d", who, i);
        FS2_REQUIRE(it.pred_dtype == FS2_F32 || it.pred_dtype == FS2_BF16, "%s: item %d: pred dtype", who, i);
        if (bwd) FS2_REQUIRE(it.dpred && (it.dpred_dtype == FS2_F32 || it.dpred_dtype == FS2_BF16), "%s: item %d: dpred", who, i);
        out->it[i] = it;
    }
    return FS2_OK;
}
extern "C" int64_t fs2_l1_multi_workspace_floats(void) { return L1_WS; }
extern "C" int fs2_l1_multi_fwd(const FS2L1Item* items, int n_items, float* losses, float* workspace, void* stream) {
    L1Items a = {};
    const int rc = l1_items_check("fs2_l1_multi_fwd", items, n_items, false, &a);
    if (rc != FS2_OK) return rc;
    FS2_REQUIRE(losses != nullptr && workspace != nullptr, "fs2_l1_multi_fwd: null losses / workspace");
    static_assert(TPB == L1_BLOCKS, "the finishing block reads one partial per thread");
    L1Counts c = {};
    for (int i = 0; i < n_items; ++i) c.n[i] = a.it[i].n;
    hipLaunchKernelGGL(l1_multi_fwd_k, dim3(L1_BLOCKS, (unsigned)n_items), dim3(TPB), 0, (hipStream_t)stream, a, workspace);
    FS2_CHECK_LAUNCH("fs2_l1_multi_fwd");
    hipLaunchKernelGGL(l1_multi_finish_k, dim3(1), dim3(TPB), 0, (hipStream_t)stream, c, n_items, workspace, losses);
    FS2_CHECK_LAUNCH("fs2_l1_multi_fwd (finish)");
    return FS2_OK;
}
extern "C" int fs2_l1_multi_bwd(const FS2L1Item* items, int n_items, const float* gscale, void* stream) {
    L1Items a = {};
    const int rc = l1_items_check("fs2_l1_multi_bwd", items, n_items, true, &a);
    if (rc != FS2_OK) return rc;
    FS2_REQUIRE(gscale != nullptr, "fs2_l1_multi_bwd: null gscale");
    hipLaunchKernelGGL(l1_multi_bwd_k, dim3(512, (unsigned)n_items), dim3(TPB), 0, (hipStream_t)stream, a, gscale);
    FS2_CHECK_LAUNCH("fs2_l1_multi_bwd");
    return FS2_OK;
}

extern "C" int fs2_bce_logits_fwd(const void* x, int x_dtype, const float* y, int64_t n, float pos_weight, float* loss, void* stream) {
    CHECK_DT("fs2_bce_logits_fwd", x_dtype);
    FS2_REQUIRE(n > 0 && x && y && loss, "fs2_bce_logits_fwd: bad arguments");
    T_DISPATCH(x_dtype, T, { hipLaunchKernelGGL((bce_fwd_k<T>), dim3(flat_grid(n)), dim3(TPB), 0, (hipStream_t)stream, (const T*)x, y, n, pos_weight, loss); });
    FS2_CHECK_LAUNCH("fs2_bce_logits_fwd");
    return FS2_OK;
}
extern "C" int fs2_bce_logits_bwd(const void* x, int x_dtype, const float* y, int64_t n, float pos_weight, const float* gscale,
                                  void* dx, int dx_dtype, void* stream) {
    CHECK_DT("fs2_bce_logits_bwd", x_dtype); CHECK_DT("fs2_bce_logits_bwd", dx_dtype);
    FS2_REQUIRE(n > 0 && x && y && gscale && dx, "fs2_bce_logits_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(flat_grid(n)), block(TPB);
    if (x_dtype == FS2_F32 && dx_dtype == FS2_F32) hipLaunchKernelGGL((bce_bwd_k<float, float>), grid, block, 0, st, (const float*)x, y, n, pos_weight, gscale, (float*)dx);
    else if (x_dtype == FS2_F32) hipLaunchKernelGGL((bce_bwd_k<float, bf16_t>), grid, block, 0, st, (const float*)x, y, n, pos_weight, gscale, (bf16_t*)dx);
    else if (dx_dtype == FS2_F32) hipLaunchKernelGGL((bce_bwd_k<bf16_t, float>), grid, block, 0, st, (const bf16_t*)x, y, n, pos_weight, gscale, (float*)dx);
    else hipLaunchKernelGGL((bce_bwd_k<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, y, n, pos_weight, gscale, (bf16_t*)dx);
    FS2_CHECK_LAUNCH("fs2_bce_logits_bwd");
    return FS2_OK;
}
extern "C" int fs2_dropout(const void* x, const void* relu_gate, void* out, int dtype, int64_t n, float p, const uint64_t* rng,
                           uint32_t site, void* stream) {
    CHECK_DT("fs2_dropout", dtype);
    FS2_REQUIRE(n > 0 && x && out, "fs2_dropout: bad arguments");
    FS2_REQUIRE(p == 0.f || rng != nullptr, "fs2_dropout: dropout needs rng");
    FS2_REQUIRE(p >= 0.f && p < 1.f, "fs2_dropout: p must be in [0, 1)");
    T_DISPATCH(dtype, T, { hipLaunchKernelGGL((dropout_k<T>), dim3(flat_grid((n + 7) >> 3)), dim3(TPB), 0, (hipStream_t)stream, (const T*)x, (const T*)relu_gate, (T*)out, n, p, rng, site); });
    FS2_CHECK_LAUNCH("fs2_dropout");
    return FS2_OK;
}

extern "C" int fs2_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
    CHECK_DT("fs2_cast", src_dtype); CHECK_DT("fs2_cast", dst_dtype);
    if (n <= 0) return FS2_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(flat_grid(n)), block(TPB);
    if (src_dtype == FS2_F32 && dst_dtype == FS2_BF16) hipLaunchKernelGGL((cast_k<float, bf16_t>), grid, block, 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == FS2_BF16 && dst_dtype == FS2_F32) hipLaunchKernelGGL((cast_k<bf16_t, float>), grid, block, 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == FS2_F32) hipLaunchKernelGGL((cast_k<float, float>), grid, block, 0, st, (const float*)src, (float*)dst, n);
    else hipLaunchKernelGGL((cast_k<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    FS2_CHECK_LAUNCH("fs2_cast");
    return FS2_OK;
}
extern "C" int fs2_add_cast(const float* a, const float* b, void* dst, int dst_dtype, int64_t n, void* stream) {
    CHECK_DT("fs2_add_cast", dst_dtype);
    if (n <= 0) return FS2_OK;
    FS2_REQUIRE(a && b && dst && fs2_aligned16(a) && fs2_aligned16(b), "fs2_add_cast: null or unaligned operand");
    dim3 grid(flat_grid((n + 3) / 4)), block(TPB);
    if (dst_dtype == FS2_BF16) hipLaunchKernelGGL((add_cast_k<bf16_t>), grid, block, 0, (hipStream_t)stream, a, b, (bf16_t*)dst, n);
    else hipLaunchKernelGGL((add_cast_k<float>), grid, block, 0, (hipStream_t)stream, a, b, (float*)dst, n);
    FS2_CHECK_LAUNCH("fs2_add_cast");
    return FS2_OK;
}
extern "C" int fs2_cast_permute(const float* src, void* dst, int O, int I, int k, int64_t dld, int mode, int dtype,
                                void* stream) {
    CHECK_DT("fs2_cast_permute", dtype);
    FS2_REQUIRE(O > 0 && I > 0 && k > 0 && (mode == 0 || mode == 1), "fs2_cast_permute: bad shape/mode");
    FS2_REQUIRE(dld >= (int64_t)k * (mode == 0 ? I : O), "fs2_cast_permute: dld too small");
    const int64_t n = (int64_t)O * I * k;
    T_DISPATCH(dtype, T, { hipLaunchKernelGGL((cast_permute_k<T>), dim3(flat_grid(n)), dim3(TPB), 0, (hipStream_t)stream, src, (T*)dst, O, I, k, dld, mode); });
    FS2_CHECK_LAUNCH("fs2_cast_permute");
    return FS2_OK;
}
extern "C" int fs2_cast_permute_batched(const FS2CastDesc* table, int n, int dtype, void* stream) {
    CHECK_DT("fs2_cast_permute_batched", dtype);
    FS2_REQUIRE(table != nullptr && n > 0 && n <= 65535, "fs2_cast_permute_batched: bad table");
    dim3 grid(96, (unsigned)n);
    T_DISPATCH(dtype, T, { hipLaunchKernelGGL((cast_permute_batched_k<T>), grid, dim3(TPB), 0, (hipStream_t)stream, table); });
    FS2_CHECK_LAUNCH("fs2_cast_permute_batched");
    return FS2_OK;
}
extern "C" int fs2_onehot(const int32_t* idx, void* out, int dtype, int64_t M, int nb, void* stream) {
    CHECK_DT("fs2_onehot", dtype);
    FS2_REQUIRE(nb > 0 && nb % 4 == 0, "fs2_onehot: nb must be a multiple of 4");
    if (M <= 0) return FS2_OK;
    T_DISPATCH(dtype, T, { hipLaunchKernelGGL((onehot_k<T>), dim3(wave_rows_grid(M)), dim3(TPB), 0, (hipStream_t)stream, idx, (T*)out, M, nb); });
    FS2_CHECK_LAUNCH("fs2_onehot");
    return FS2_OK;
}
// ---------------------------------------------------------------- split-K finishing pass
// out[m][n] = act(scratch[m][n] + bias[n]) (+ residual[m][n]) in TO, and scratch := 0 again (self-cleaning), for products
// whose output has too few tiles to fill the chip: they run split-K into an fp32 scratch (fs2_gemm accumulate) and the
// bias / ReLU / residual / cast that the fused epilogue would have done happens here, one 16-byte group per thread.
template <typename TO>
__global__ __launch_bounds__(TPB) void splitk_finish_k(float* __restrict__ scratch, int64_t groups, int N4,
        const float* __restrict__ bias, const void* __restrict__ residual, int res_dtype, int64_t ldr, int relu,
        TO* __restrict__ out, int64_t ldc) {
    for (int64_t gi = (int64_t)blockIdx.x * TPB + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * TPB) {
        const int64_t m = gi / N4;
        const int n = (int)(gi - m * N4) * 4;
        float4* sp = reinterpret_cast<float4*>(scratch + m * (int64_t)(N4 * 4) + n);
        float4 v = *sp;
        *sp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias != nullptr) {
            const float4 b = *reinterpret_cast<const float4*>(bias + n);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (residual != nullptr) {
            float4 r;
            if (res_dtype == FS2_F32) r = load4<float>(reinterpret_cast<const float*>(residual) + m * ldr + n);
            else r = load4<bf16_t>(reinterpret_cast<const bf16_t*>(residual) + m * ldr + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        store4<TO>(out + m * ldc + n, v);
    }
}

extern "C" int fs2_splitk_finish(float* scratch, int64_t M, int N, const float* bias, const void* residual, int res_dtype,
                                 int64_t ldr, int relu, void* out, int out_dtype, int64_t ldc, void* stream) {
    CHECK_DT("fs2_splitk_finish", out_dtype);
    FS2_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0 && ldc >= N && (residual == nullptr || (ldr % 4 == 0 && ldr >= N)),
                "fs2_splitk_finish: N, ldc, ldr must be multiples of 4");
    FS2_REQUIRE(fs2_aligned16(scratch) && fs2_aligned16(out) && (bias == nullptr || fs2_aligned16(bias)), "fs2_splitk_finish: 16-byte alignment required");
    const int64_t groups = M * (N / 4);
    T_DISPATCH(out_dtype, TO, { hipLaunchKernelGGL((splitk_finish_k<TO>), dim3(flat_grid(groups)), dim3(TPB), 0, (hipStream_t)stream,
                                                   scratch, groups, N / 4, bias, residual, res_dtype, ldr, relu, (TO*)out, ldc); });
    FS2_CHECK_LAUNCH("fs2_splitk_finish");
    return FS2_OK;
}

// Sliced split-K (fs2_gemm accumulate = 2): out[m][n] = act(sum_s slices[s][m][n] + bias[n]) (+ residual[m][n]).  Plain loads of
// nsplit fp32 slices (slice stride `sstride` elements, row stride ld): no atomics were involved, nothing to clean.
template <typename TO>
__global__ __launch_bounds__(TPB) void splitk_reduce_k(const float* __restrict__ slices, int nsplit, int64_t sstride, int64_t ld, int64_t groups,
        int N4, const float* __restrict__ bias, const void* __restrict__ residual, int res_dtype, int64_t ldr, int relu,
        TO* __restrict__ out, int64_t ldc) {
    for (int64_t gi = (int64_t)blockIdx.x * TPB + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * TPB) {
        const int64_t m = gi / N4;
        const int n = (int)(gi - m * N4) * 4;
        const float* sp = slices + m * ld + n;
        float4 v = *reinterpret_cast<const float4*>(sp);
        for (int s = 1; s < nsplit; ++s) {
            const float4 w = *reinterpret_cast<const float4*>(sp + s * sstride);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (bias != nullptr) {
            const float4 b = *reinterpret_cast<const float4*>(bias + n);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (residual != nullptr) {
            float4 r;
            if (res_dtype == FS2_F32) r = load4<float>(reinterpret_cast<const float*>(residual) + m * ldr + n);
            else r = load4<bf16_t>(reinterpret_cast<const bf16_t*>(residual) + m * ldr + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        store4<TO>(out + m * ldc + n, v);
    }
}

extern "C" int fs2_splitk_reduce(const float* slices, int nsplit, int64_t slice_stride, int64_t ld, int64_t M, int N, const float* bias,
                                 const void* residual, int res_dtype, int64_t ldr, int relu, void* out, int out_dtype, int64_t ldc,
                                 void* stream) {
    CHECK_DT("fs2_splitk_reduce", out_dtype);
    FS2_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0 && ldc >= N && ld % 4 == 0 && ld >= N && (residual == nullptr || (ldr % 4 == 0 && ldr >= N)),
                "fs2_splitk_reduce: N, ld, ldc, ldr must be multiples of 4");
    FS2_REQUIRE(nsplit >= 1 && nsplit <= 64 && slice_stride % 4 == 0 && (nsplit == 1 || slice_stride >= M * ld), "fs2_splitk_reduce: bad slice layout");
    FS2_REQUIRE(fs2_aligned16(slices) && fs2_aligned16(out) && (bias == nullptr || fs2_aligned16(bias)), "fs2_splitk_reduce: 16-byte alignment required");
    const int64_t groups = M * (N / 4);
    T_DISPATCH(out_dtype, TO, { hipLaunchKernelGGL((splitk_reduce_k<TO>), dim3(flat_grid(groups)), dim3(TPB), 0, (hipStream_t)stream,
                                                   slices, nsplit, slice_stride, ld, groups, N / 4, bias, residual, res_dtype, ldr, relu, (TO*)out, ldc); });
    FS2_CHECK_LAUNCH("fs2_splitk_reduce");
    return FS2_OK;
}

extern "C" int fs2_permute_add(float* scratch, float* grad, int O, int I, int k, int rezero, void* stream) {
    FS2_REQUIRE(O > 0 && I > 0 && k > 0, "fs2_permute_add: bad shape");
    const int64_t n = (int64_t)O * I * k;
    if (k <= 9 && k > 1) {
        int64_t items = (int64_t)((O + 3) / 4) * ((I + 63) / 64);
        if (items > 2048) items = 2048;
        hipLaunchKernelGGL(permute_add_tiled_k, dim3((unsigned)items), dim3(TPB), 0, (hipStream_t)stream, scratch, grad, O, I, k, rezero);
    } else {
        hipLaunchKernelGGL(permute_add_k, dim3(flat_grid(n)), dim3(TPB), 0, (hipStream_t)stream, scratch, grad, O, I, k, rezero);
    }
    FS2_CHECK_LAUNCH("fs2_permute_add");
    return FS2_OK;
}
static int colsum_launch(const char* name, const void* x, int dtype, int64_t M, int N, int64_t ldx, float* out, int seg_cols,
                         int64_t seg_stride, void* stream) {
    if (M <= 0) return FS2_OK;
    int64_t slabs = (M + 3) / 4;
    if (slabs > 512) slabs = 512;
    if (N <= 128) {         // narrow: several rows per wave
        const int rpw = 64 / (N / 4);
        // (every block ends with N atomics on the same N addresses: 512 blocks serialised ~15 of this kernel's 19 us at 44,400 x 80)
        slabs = (M + 32 * rpw - 1) / (32 * rpw);
        if (slabs > 128) slabs = 128;
        dim3 gridn(1, (unsigned)slabs);
        T_DISPATCH(dtype, T, { hipLaunchKernelGGL((colsum_narrow_k<T>), gridn, dim3(TPB), 0, (hipStream_t)stream, (const T*)x, M, N, ldx, out, seg_cols, seg_stride); });
        FS2_CHECK_LAUNCH(name);
        return FS2_OK;
    }
    dim3 grid((N + 255) / 256, (unsigned)slabs);
    T_DISPATCH(dtype, T, { hipLaunchKernelGGL((colsum_k<T>), grid, dim3(TPB), 0, (hipStream_t)stream, (const T*)x, M, N, ldx, out, seg_cols, seg_stride); });
    FS2_CHECK_LAUNCH(name);
    return FS2_OK;
}
extern "C" int fs2_colsum(const void* x, int dtype, int64_t M, int N, int64_t ldx, float* out, void* stream) {
    CHECK_DT("fs2_colsum", dtype);
    FS2_REQUIRE(N > 0 && N % 4 == 0 && ldx % 4 == 0 && ldx >= N, "fs2_colsum: N and ldx must be multiples of 4");
    return colsum_launch("fs2_colsum", x, dtype, M, N, ldx, out, N, 0, stream);
}
extern "C" int fs2_colsum_segmented(const void* x, int dtype, int64_t M, int N, int64_t ldx, float* out, int seg_cols,
                                    int64_t seg_stride, void* stream) {
    CHECK_DT("fs2_colsum_segmented", dtype);
    FS2_REQUIRE(N > 0 && N % 4 == 0 && ldx % 4 == 0 && ldx >= N, "fs2_colsum_segmented: N and ldx must be multiples of 4");
    FS2_REQUIRE(seg_cols > 0 && N % seg_cols == 0 && seg_stride >= seg_cols, "fs2_colsum_segmented: need N %% seg_cols == 0 and seg_stride >= seg_cols");
    return colsum_launch("fs2_colsum_segmented", x, dtype, M, N, ldx, out, seg_cols, seg_stride, stream);
}

extern "C" int fs2_sqnorm(const float* x, int64_t n, float* out, void* stream) {
    FS2_REQUIRE(n > 0 && fs2_aligned16(x), "fs2_sqnorm: n > 0 and 16-byte aligned x required");
    const int blocks = flat_grid(n >> 2) < 768 ? flat_grid(n >> 2) : 768;      // every block ends with one atomic on `out`: contended atomics serialise
    hipLaunchKernelGGL(sqnorm_k, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream, x, n, out);
    FS2_CHECK_LAUNCH("fs2_sqnorm");
    return FS2_OK;
}
extern "C" int fs2_adam_step_perm(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, const float* gsq,
                                  float beta1, float beta2, float eps, float max_norm, const int64_t* perm_segments, int n_segments,
                                  void* stream) {
    FS2_REQUIRE(n > 0 && fs2_aligned16(p) && fs2_aligned16(g) && fs2_aligned16(m) && fs2_aligned16(v), "fs2_adam_step: arenas must be 16-byte aligned");
    FS2_REQUIRE(n_segments >= 0 && (n_segments == 0 || perm_segments != nullptr), "fs2_adam_step_perm: bad segment table");
    if (n_segments > 0)
        hipLaunchKernelGGL((adam_k<true>), dim3(flat_grid(n >> 2)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, hyper, gsq, beta1, beta2, eps,
                           max_norm, perm_segments, n_segments);
    else
        hipLaunchKernelGGL((adam_k<false>), dim3(flat_grid(n >> 2)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, hyper, gsq, beta1, beta2, eps,
                           max_norm, (const int64_t*)nullptr, 0);
    FS2_CHECK_LAUNCH("fs2_adam_step");
    return FS2_OK;
}
extern "C" int fs2_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, const float* gsq,
                             float beta1, float beta2, float eps, float max_norm, void* stream) {
    return fs2_adam_step_perm(p, g, m, v, n, hyper, gsq, beta1, beta2, eps, max_norm, nullptr, 0, stream);
}
// n16 16-byte words of zeros (the per-step clear of the gradient arena and of the small accumulators that live behind it)
__global__ __launch_bounds__(256) void zero_k(float4* __restrict__ p, int64_t n16) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = z;
}
extern "C" int fs2_zero(void* ptr, int64_t nbytes, void* stream) {
    FS2_REQUIRE(ptr != nullptr && nbytes >= 0 && fs2_aligned16(ptr) && nbytes % 16 == 0, "fs2_zero: 16-byte aligned pointer and size required");
    if (nbytes == 0) return FS2_OK;
    const int64_t n16 = nbytes / 16;
    const int64_t want = (n16 + 255) / 256;
    hipLaunchKernelGGL(zero_k, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, (float4*)ptr, n16);
    FS2_CHECK_LAUNCH("fs2_zero");
    return FS2_OK;
}
// up to 8 device-to-device copies in one launch (the seven input tensors of a step into the static buffers of its captured graph:
// seven copy launches of the runtime otherwise); blockIdx.y = copy, 16-byte chunks + byte tail
struct CopyItems { const unsigned char* src[8]; unsigned char* dst[8]; int64_t n[8]; };
__global__ __launch_bounds__(256) void copy_batched_k(const CopyItems it) {
    const unsigned char* src = it.src[blockIdx.y];
    unsigned char* dst = it.dst[blockIdx.y];
    const int64_t n = it.n[blockIdx.y];
    int64_t done = 0;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const int64_t n16 = n >> 4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        done = n16 << 4;
    }
    for (int64_t i = done + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int fs2_copy_batched(const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream) {
    FS2_REQUIRE(src && dst && nbytes && n > 0 && n <= 8, "fs2_copy_batched: 1..8 copies");
    CopyItems it = {};
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        FS2_REQUIRE(nbytes[i] >= 0 && (nbytes[i] == 0 || (src[i] && dst[i])), "fs2_copy_batched: bad copy %d", i);
        it.src[i] = static_cast<const unsigned char*>(src[i]); it.dst[i] = static_cast<unsigned char*>(dst[i]); it.n[i] = nbytes[i];
        most = nbytes[i] > most ? nbytes[i] : most;
    }
    if (most == 0) return FS2_OK;
    int64_t blocks = (most / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(copy_batched_k, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, it);
    FS2_CHECK_LAUNCH("fs2_copy_batched");
    return FS2_OK;
}
extern "C" int fs2_rng_advance(uint64_t* rng, void* stream) {
    hipLaunchKernelGGL(rng_advance_k, dim3(1), dim3(1), 0, (hipStream_t)stream, rng);
    FS2_CHECK_LAUNCH("fs2_rng_advance");
    return FS2_OK;
}

/*
 * fs2_hip.h -- C ABI of libfs2_hip.so: the MI355X (gfx950) kernels of the FastSpeech2 training path.
 *
 * The reference (syoamakase/Transformer_TTS) is pure Python/PyTorch and has NO plugin / operator /
 * FFI interface: its hot path calls stock PyTorch ops.  This header is the
 * boundary the build defines in their place; every entry point cites the reference call site whose
 * arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller;
 *  - no allocation, no ownership transfer, no host synchronisation, no host reads of device data:
 *    every call only enqueues kernels on `stream` (a hipStream_t) and is graph-capturable;
 *  - stateless and re-entrant; return 0, or a negative FS2_E* code with text in fs2_last_error();
 *  - activations are channels-last [rows = (batch, time)][channels], row-major;
 *  - dtype codes: FS2_F32 = 0 (exact-fp32 mode: f32-input MFMA), FS2_BF16 = 1 (bf16 MFMA, fp32 accumulate);
 *  - dropout: Philox4x32-7 keyed by the device-resident {seed, offset} pair `rng` and a per call
 *    site id `site`; the backward entry point regenerates the mask from the same triple.
 */
#ifndef FS2_HIP_H
#define FS2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS2_F32 0
#define FS2_BF16 1
#define FS2_FP8 2      /* fs2_gemm operands only: A and B OCP fp8 e4m3 (one byte per element) */
#define FS2_BF8_FP8 3  /* fs2_gemm operands only: A OCP fp8 e5m2 (a gradient), B e4m3 (weights) */

#define FS2_OK 0
#define FS2_EINVAL (-1)   /* bad shape / alignment / argument */
#define FS2_ELAUNCH (-2)  /* hipLaunchKernel failed */

const char* fs2_last_error(void);
int fs2_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM family (MFMA).  C[M,N] = alpha * sum_k A(m,k) * B(n,k)  (+bias[n]) (ReLU) (*relu_mask>0) (+residual)
 *
 * Replaces: nn.Linear (Models/modules.py:32-41,49-51,68; Models/encoder.py:57; Models/postnets.py:43,67),
 * nn.Conv1d as implicit GEMM (Models/modules.py:76-84; Models/varianceadaptor.py:203-209,216-221;
 * Models/postnets.py:28-39,71-75), torch.matmul in attention() (Models/modules.py:8,20) and the autograd
 * backward of all of them (dgrad = same kernel on a transposed/flipped weight shadow, wgrad = k-major form).
 *
 *  a_kmajor = 0: A stored [M][K] (lda = row stride);  a_kmajor = 1: A stored [K][M].  Same for B / N.
 *  conv = 1 (A row-major only): rows are (b,t) with t = m % seq_len; the reduction runs over `taps`
 *      x K with A row m+tap-pad (zero outside [0,seq_len)) and B column tap*K + k  (B is [N][taps*K]).
 *  conv = 2 (A and B k-major, wgrad): reduction rows are (b,t); B row r is read at r + batch2_index - pad
 *      (zero outside the sequence); batch2 enumerates the taps (use sB2 = 0, sC2 = K_in).
 *  Row strides and the contiguous-dim extents must be multiples of 16 bytes (8 bf16 / 4 f32).
 *  Kb = valid reduction extent of B when it differs from K (0 = K): rows >= Kb of a k-major B read as 0.
 *  accumulate = 1: C is fp32 and results are atomically added (split_k > 1 requires accumulate != 0).
 *  accumulate = 2 (row-major bf16 operands, un-batched, no fused epilogue operand; fp32 C): SLICED split-K -- C is a workspace of
 *      split_k slices [split][M][ldc] (slice stride sC1 elements); slice s receives the partial sums of the s-th part of the
 *      reduction with plain stores (nothing is read, nothing has to be zeroed); fs2_splitk_reduce adds the slices up and applies
 *      bias / ReLU / residual / cast.  split_k is lowered so that no part is empty: read it back with fs2_gemm_last_splits().
 *  colstats != NULL: per-column sum and sum of squares of the stored values are atomically added to
 *      colstats[0..N) and colstats[N..2N)  (BatchNorm batch statistics, Models/postnets.py:58-59).
 */
typedef struct FS2Gemm {
    const void* A;
    const void* B;
    void* C;
    const float* bias;
    const void* residual;
    const void* relu_mask;
    float* colstats;
    int64_t lda, ldb, ldc, ldr, ldm;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2; /* batch strides in elements (C strides also apply to residual/mask) */
    int32_t M, N, K, Kb;
    int32_t a_kmajor, b_kmajor;
    int32_t dtype;      /* of A and B */
    int32_t c_dtype;    /* of C */
    int32_t res_dtype;  /* of residual */
    int32_t relu;
    int32_t accumulate;
    int32_t split_k;
    int32_t batch1, batch2;
    int32_t conv, taps, pad, seq_len;
    float alpha;
    int32_t colstats_mode; /* 0: sum and sum of squares; 1: column sums only (bias gradient) into colstats[0..N) */
    int32_t tile_order;    /* 0: chosen by fs2_gemm; 1: output tiles walked n-fastest; 2: m-fastest with the column panels
                              spread so that the workgroups sharing a row slab of A run at the same time on one XCD (one
                              L2): for tall products with 2..8 column tiles, where a 128-row slab of A would otherwise be
                              fetched from HBM once per column tile */
    /* fp8 operands (dtype FS2_FP8 / FS2_BF8_FP8; row-major tall products on the 16-wave kernel: K, lda, ldb multiples of 16):
     * DEVICE scalars multiplied into alpha -- the de-quantisation factors fs2_quantize_fp8 wrote for A and B (NULL = 1). */
    const float* scale_a;
    const float* scale_b;
    /* fp8 copy of the OUTPUT written by the epilogue (the 16-wave ring kernel with fp8 operands and a bf16 C; otherwise must be NULL):
     * q8[m][n] (ld = ldc, one byte per element) = fp8(C[m][n] as stored * scale), scale = the power of two fs2_quantize_fp8 would derive
     * from q8_prev[0] -- a SPECULATION: the amax this tensor had one training step ago.  The epilogue also reduces max |C| into
     * q8_state[0] (zeroed by the caller).  fs2_quantize_fp8_repair then writes q8_state[1] and re-quantises from C only if the true amax
     * asks for another scale: the codes equal fs2_amax + fs2_quantize_fp8 of C, always, without a pass over C in the common case. */
    void* q8;
    float* q8_state;
    const float* q8_prev;
    int32_t q8_bf8;      /* 0: e4m3, 1: e5m2 */
    int32_t q8_reserved;
} FS2Gemm;

int fs2_gemm(const FS2Gemm* g, void* stream);
/* Rows of the block tile the last fs2_gemm call of this thread was launched with: 64 / 128 (4-wave kernel) or 192 / 256
 * (16-wave LDS-DMA kernel for tall row-major bf16 products, 256 columns wide).  Measurement aid (bench.py groups its
 * per-launch timings by it); FS2_GEMM_BIG=0 in the environment keeps every product on the 4-wave kernel. */
int fs2_gemm_last_tile(void);
/* Number of slices the last sliced split-K fs2_gemm call (accumulate = 2) of this thread wrote (<= FS2Gemm.split_k). */
int fs2_gemm_last_splits(void);

/* Finishing pass of a split-K product whose fused epilogue could not run (few output tiles, long K: the product
 * accumulates fp32 partial sums into `scratch` [M][N], zero on entry, with FS2Gemm.accumulate = 1, split_k > 1):
 *   out[m][n] = act(scratch[m][n] + bias[n]) + residual[m][n]   in out_dtype;   scratch is zero again on return.  */
int fs2_splitk_finish(float* scratch, int64_t M, int N, const float* bias, const void* residual, int res_dtype,
                      int64_t ldr, int relu, void* out, int out_dtype, int64_t ldc, void* stream);

/* Finishing pass of a SLICED split-K product (FS2Gemm.accumulate = 2):
 *   out[m][n] = act(sum_{s < nsplit} slices[s * slice_stride + m * ld + n] + bias[n]) + residual[m][n]   in out_dtype.  */
int fs2_splitk_reduce(const float* slices, int nsplit, int64_t slice_stride, int64_t ld, int64_t M, int N, const float* bias,
                      const void* residual, int res_dtype, int64_t ldr, int relu, void* out, int out_dtype, int64_t ldc,
                      void* stream);

/* Weight gradients without float atomics.  fs2_wgrad_sliced runs the product of an fs2_gemm descriptor with a_kmajor = b_kmajor = 1,
 * accumulate = 1 (dW += dY^T X, Conv1d taps as conv = 2) but stores the 128 x 128 partial tiles of its k-split with plain stores into
 * the workspace `ws` (64 KiB per workgroup, ~6 TB/s chip-wide against ~1.3 TB/s of float atomics) and describes them in `part`;
 * fs2_wgrad_reduce(parts, n) later adds the partial tiles of n such products into their gradients, one launch for all of them (the
 * trainer reduces once per announced parameter range).  Returns the number of floats of `ws` used, 0 when the product does not run
 * in that form (the caller then calls fs2_gemm), negative on error.  part->splits > 0: uniform k-split (slice = item number);
 * part->splits = -U < 0: balanced stream of U stages per workgroup (slice = workgroup + tile; part->reserved = stages of the whole
 * reduction, bit 30 set for conv = 2): fs2_wgrad_reduce reads both forms.                                                      */
typedef struct FS2WgradPart {
    const float* ws;
    float* dst;
    int64_t ldc, sC1, sC2;
    int32_t M, N, tilesM, tilesN, splits, n2, nbatch, block_begin;
    float alpha;
    int32_t reserved;
    const float* scale_a;      /* fp8 operands: device de-quantisation factors multiplied into alpha (NULL = 1) */
    const float* scale_b;
} FS2WgradPart;
int64_t fs2_wgrad_sliced(const FS2Gemm* g, float* ws, int64_t ws_floats, FS2WgradPart* part, void* stream);
/* fp8 operands (dtype FS2_BF8_FP8: A = dY in e5m2, B = X in e4m3, one byte per element, lda / ldb multiples of 16, FS2Gemm.scale_a / scale_b)
 * are taken by fs2_wgrad_sliced / fs2_wgrad_grouped only.  fs2_wgrad_plan tells beforehand how a product would run: 0 the 16-wave kernel
 * does not take it, 1 uniform k-split (partial tiles), 2 balanced stream (partial tiles while the workspace has room) or more than 256
 * output tiles.  Where a form-2 product does not get partial tiles: bf16 goes through fs2_gemm (fs2_wgrad_sliced returns 0); fp8:
 * fs2_wgrad_sliced launches it with the float-atomic flush, returns 1 and sets part->splits = 0 (complete, nothing to reduce). */
int fs2_wgrad_plan(const FS2Gemm* g);
/* n <= 4 such products in ONE launch (the weight gradients of one layer's backward, launched when the last of them is known): one
 * launch ramp / first-stage latency / tail for the group, k-splits sized for the group's total work.  parts[i] describes product i.
 * Products the group does not take are marked parts[i].splits == 0 (the caller launches those on their own).  Returns the floats of
 * `ws` used, or 0 when nothing was launched. */
int64_t fs2_wgrad_grouped(const FS2Gemm* descs, int n, float* ws, int64_t ws_floats, FS2WgradPart* parts, void* stream);
int fs2_wgrad_reduce(const FS2WgradPart* parts, int n, void* stream);

/* Weight shadows (fp32 master (O, I, k) as in the reference state_dict -> kernel layout, dtype `dtype`):
 *  mode 0 (forward):  dst[o*dld + j*I + i]       = src[o][i][j]
 *  mode 1 (dgrad):    dst[i*dld + j*O + o]       = src[o][i][k-1-j]
 * and the inverse for weight gradients: grad[o][i][j] += scratch[o*(k*I) + j*I + i].          */
int fs2_cast_permute(const float* src, void* dst, int O, int I, int k, int64_t dld, int mode, int dtype, void* stream);
int fs2_permute_add(float* scratch, float* grad, int O, int I, int k, int rezero_scratch, void* stream);
int fs2_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* dst (dst_dtype) = a + b, both fp32, n elements: the two gradient terms that reach the post-net's mel_pred (reference
 * Models/postnets.py:67,74-75: mel_pred feeds the loss directly and the convolution stack) as one operand. */
int fs2_add_cast(const float* a, const float* b, void* dst, int dst_dtype, int64_t n, void* stream);
/* n <= 8 device-to-device copies of nbytes[i] bytes each in ONE launch (host arrays of device pointers): the input tensors of a step
 * into the static buffers of its captured hipGraph (train_fastspeech2.GraphedTrainStep), one launch instead of one per tensor. */
int fs2_copy_batched(const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream);
/* All weight shadows of a model in ONE launch: `table` is a DEVICE array of n descriptors. */
typedef struct FS2CastDesc {
    const float* src;
    void* dst;
    int64_t dld;
    int32_t O, I, k, mode; /* as fs2_cast_permute; mode 2 = plain fp32 copy of O*I*k elements (fused bias vectors) */
} FS2CastDesc;
int fs2_cast_permute_batched(const FS2CastDesc* table, int n, int dtype, void* stream);
/* out[m][idx[m]] = 1, every other entry of the row 0  (one-hot operand of the embedding-gradient GEMM) */
int fs2_onehot(const int32_t* idx, void* out, int dtype, int64_t M, int nb, void* stream);
/* out[n] += sum_m x[m][n]   (bias gradients); x is [M][N] with row stride ldx */
int fs2_colsum(const void* x, int dtype, int64_t M, int N, int64_t ldx, float* out, void* stream);
/* as fs2_colsum, but column c is added to out[(c / seg_cols) * seg_stride + c % seg_cols]: the column blocks of one fused
 * tensor (the q / v / k bias gradients of a fused projection) go to vectors that sit seg_stride floats apart. */
int fs2_colsum_segmented(const void* x, int dtype, int64_t M, int N, int64_t ldx, float* out, int seg_cols,
                         int64_t seg_stride, void* stream);

/* Per-tensor fp8 quantisation with current scaling (BASELINE.json configs[4]: fp8 MFMA GEMMs).
 *   fs2_amax:          state[0] = max(state[0], max |src|)   (state[0] zeroed by the caller beforehand)
 *   fs2_quantize_fp8:  scale = 2^k, the largest power of two with state[0] * scale < 2^8 (e4m3, bf8 = 0) or 2^15 (e5m2,
 *                      bf8 = 1); dst[i] = round-to-nearest-even fp8 of src[i] * scale (saturating); state[1] = 1 / scale, the
 *                      factor fs2_gemm takes as FS2Gemm.scale_a / scale_b.  n is rounded up to 16 internally; dst must hold it. */
int fs2_amax(const void* src, int src_dtype, int64_t n, float* state, void* stream);
int fs2_quantize_fp8(const void* src, int src_dtype, void* dst, int bf8, int64_t n, float* state, void* stream);
/* The same for the row kernels that produce a GEMM operand: the NEXT fs2_add_ln_fwd (y), fs2_add_ln_bwd (da), fs2_ffn_tail_fwd (y) or
 * fs2_ffn_tail_bwd (g) launch of this host thread with a bf16 output also writes its fp8 copy q8[row * d + col] (scale speculated
 * from prev[0], amax into state[0]; rows must be a multiple of 4 columns long, which these kernels require anyway). */
int fs2_q8_next(void* q8, float* state, const float* prev, int bf8);
/* Second half of FS2Gemm.q8 / fs2_q8_next: state[1] = 1/scale(state[0]); if scale(state[0]) != scale(prev[0]) quantise src again into dst. */
int fs2_quantize_fp8_repair(const void* src, int src_dtype, void* dst, int bf8, int64_t n, float* state, const float* prev, void* stream);
/* The same for many bf16 tensors in two launches (the weight shadows of a model, once per optimizer step).  `table` lives in DEVICE
 * memory; entry i covers blocks [block_begin, block_begin + nblocks) of the grid, nblocks = ceil(n / 32768) (at least 1); the caller
 * zeroes every state[0] beforehand; dst holds n rounded up to 16 bytes. */
typedef struct FS2QuantDesc {
    const void* src;
    void* dst;
    float* state;
    int64_t n;
    int32_t block_begin, nblocks;
} FS2QuantDesc;
int fs2_quantize_fp8_batched(const FS2QuantDesc* table, int n, int total_blocks, int bf8, void* stream);

/* nn.Embedding gather / scatter-add (Models/encoder.py:55,84; Models/varianceadaptor.py:57,62). */
int fs2_embedding_fwd(const int64_t* ids, const float* table, void* out, int out_dtype, int64_t n, int d, void* stream);
int fs2_embedding_bwd(const int64_t* ids, const void* dout, int dout_dtype, float* dtable, int64_t n, int d,
                      int64_t padding_idx, void* stream);

/* PositionalEncoder (Models/modules.py:107-111): out = dropout(a + alpha * pe[t]);  backward gives da and dalpha. */
int fs2_pe_add_fwd(const void* a, int a_dtype, const float* pe, const float* alpha, float* out, int B, int t, int d,
                   float p, const uint64_t* rng, uint32_t site, void* stream);
int fs2_pe_add_bwd(const float* dout, const float* pe, void* da, int da_dtype, float* dalpha, int B, int t, int d,
                   float p, const uint64_t* rng, uint32_t site, float* dcolsum, void* stream);

/* The head of an FFT stack in one row pass each way: x = dropout(a + alpha * pe[t]) (fp32, the residual stream) and y = LayerNorm(x)
 * (Models/modules.py:107-111, then norm_1 of the first layer, Models/layers.py:31); with ids != NULL row r of `a` is table row ids[r]
 * (nn.Embedding, Models/encoder.py:55,84; fp32 table).  Backward: LayerNorm backward of dy (+ ds, the residual stream's gradient, may
 * be NULL), then the positional encoder's: da, dalpha += sum(da * pe), dcolsum (may be NULL) += column sums of da, dgamma / dbeta. */
int fs2_pe_add_ln_fwd(const void* a, int a_dtype, const int64_t* ids, const float* pe, const float* alpha, const float* gamma,
                      const float* beta, float* x, void* y, int y_dtype, float* mean, float* rstd, int B, int t, int d, float eps,
                      float p, const uint64_t* rng, uint32_t site, void* stream);
int fs2_ln_pe_add_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                      const float* ds, const float* pe, void* da, int da_dtype, float* dgamma, float* dbeta, float* dalpha,
                      float* dcolsum, int B, int t, int d, float p, const uint64_t* rng, uint32_t site, void* stream);

/* dcolsum (may be NULL) on the backward entry points below: dcolsum[c] += sum over rows of the gradient tensor the
 * call writes (dx / da / g) -- the bias gradient of the layer that produced the forward input, fused here so the
 * tensor is not read a second time.                                                                             */
/* nn.LayerNorm over the last dim (eps 1e-5), optionally followed by dropout (Models/varianceadaptor.py:219,222)
 * and, in backward, by the ReLU mask of its input (x > 0).  dx_accumulate: dx += instead of dx =.        */
int fs2_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                      float* mean, float* rstd, int64_t M, int d, float eps, float p, const uint64_t* rng,
                      uint32_t site, void* stream);
int fs2_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                      const float* rstd, void* dx, int dx_dtype, float* dgamma, float* dbeta, int64_t M, int d,
                      float p, const uint64_t* rng, uint32_t site, int relu_mask, int dx_accumulate, float* dcolsum,
                      void* stream);

/* Residual + dropout + LayerNorm (Models/layers.py:31-35,40 with the next norm fused):
 *   s = r + dropout_p(a);  y = LN(s)      r,s fp32 [M][d];  a,y dtype `dtype`.
 * backward: ds_total = ds_down (may be NULL) + LNbwd(dy);  dr = ds_total;  da = dropout'(ds_total).  */
int fs2_add_ln_fwd(const float* r, const void* a, int dtype, float* s, const float* gamma, const float* beta, void* y,
                   float* mean, float* rstd, int64_t M, int d, float eps, float p, const uint64_t* rng, uint32_t site,
                   void* stream);
int fs2_add_ln_bwd(const float* ds_down, const void* dy, int dtype, const float* s, const float* gamma,
                   const float* mean, const float* rstd, float* dr, void* da, float* dgamma, float* dbeta, int64_t M,
                   int d, float p, const uint64_t* rng, uint32_t site, float* dcolsum, void* stream);

/* FeedForward tail (Models/modules.py:85-87): y = LN(dropout_p(f2 + h)) ;  backward returns g = d(f2) = d(h). */
int fs2_ffn_ln_fwd(const void* f2, const void* h, int dtype, const float* gamma, const float* beta, void* y,
                   float* mean, float* rstd, int64_t M, int d, float eps, float p, const uint64_t* rng, uint32_t site,
                   void* stream);
int fs2_ffn_ln_bwd(const void* dy, const void* f2, const void* h, int dtype, const float* gamma, const float* mean,
                   const float* rstd, void* g, float* dgamma, float* dbeta, int64_t M, int d, float p,
                   const uint64_t* rng, uint32_t site, float* dcolsum, void* stream);

/* The FeedForward tail TOGETHER with the residual add and the LayerNorm that follow it in EncoderLayer.forward
 * (Models/modules.py:85-87, Models/layers.py:40,31, Models/encoder.py:112): one row pass instead of fs2_ffn_ln_* + fs2_add_ln_*.
 *   forward:  yff = LN1(dropout(f2 + h), site1);  s = r + dropout(yff, site2);  y = LN2(s)     (statistics of both returned)
 *   backward: dr = ds_down + LN2bwd(dy);  g = d(f2) = d(h) = dropout1'(LN1bwd(dropout2'(dr)));  the four affine gradients and
 *             dcolsum (+= column sums of g: the bias gradient of the second convolution) accumulate.
 * The intermediate is rounded to `dtype` where the two-kernel form stores it: same results up to the last place.           */
int fs2_ffn_tail_fwd(const void* f2, const void* h, int dtype, const float* r, const float* gamma1, const float* beta1,
                     const float* gamma2, const float* beta2, float* s, void* y, float* mean1, float* rstd1, float* mean2,
                     float* rstd2, int64_t M, int d, float eps, float p, const uint64_t* rng, uint32_t site1, uint32_t site2,
                     void* stream);
int fs2_ffn_tail_bwd(const float* ds_down, const void* dy, int dtype, const float* s, const float* gamma2, const float* mean2,
                     const float* rstd2, const void* f2, const void* h, const float* gamma1, const float* mean1,
                     const float* rstd1, float* dr, void* g, float* dgamma2, float* dbeta2, float* dgamma1, float* dbeta1,
                     float* dcolsum, int64_t M, int d, float p, const uint64_t* rng, uint32_t site1, uint32_t site2,
                     void* stream);

/* attention() softmax (Models/modules.py:9-19): in place on S [rows = B*H*t][ld = tp >= t]:
 *   P = softmax(mask_keys(S, -1e4));  P_drop = dropout_p(P) (always on).  S already holds QK^T/sqrt(d_k).
 * key_mask [B][t] bytes (0 = padded key).  Columns [t, tp) are written as 0.  p_drop may alias p_out when p == 0.
 * backward: dS = P * (dP' - sum_j dP'_j P_j), dP' = dropout'(dP), written in place over dP.           */
int fs2_softmax_fwd(void* s_inout_p, void* p_drop, int dtype, const uint8_t* key_mask, int B, int H, int t, int tp,
                    int64_t batch_stride, float p, const uint64_t* rng, uint32_t site, void* stream);
int fs2_softmax_bwd(void* dp_inout_ds, int64_t dp_batch_stride, const void* p_saved, int64_t p_batch_stride, int dtype,
                    int B, int H, int t, int tp, float p, const uint64_t* rng, uint32_t site, void* stream);

/* Rectangular / causal form of the two entry points above, for the autoregressive Transformer-TTS decoder
 * (Models/layers.py:108-118 with the masks of train.py:26-58): tq query rows per head against tk keys (row stride
 * tkp >= tk, a multiple of 8); key_mask is (B, tk); causal != 0 (needs tq == tk) additionally masks key j > query i. */
int fs2_softmax_rect_fwd(void* s_inout_p, void* p_drop, int dtype, const uint8_t* key_mask, int B, int H, int tq, int tk, int tkp,
                         int64_t batch_stride, int causal, float p, const uint64_t* rng, uint32_t site, void* stream);
int fs2_softmax_rect_bwd(void* dp_inout_ds, int64_t dp_batch_stride, const void* p_saved, int64_t p_batch_stride, int dtype,
                         int B, int H, int tq, int tk, int tkp, float p, const uint64_t* rng, uint32_t site, void* stream);

/* attention() scores + softmax + dropout in one kernel (Models/modules.py:8-19), bf16 only: a workgroup keeps the
 * 64 x tp score strip of its query rows in LDS, so QK^T/sqrt(d_k) never reaches HBM.
 *   q, k: rows of one head = dk contiguous bf16 at  base + b*batch_stride + i*row_stride + h*head_stride  (elements)
 *   p_out / p_drop: (B,[..],H,t,tp) as fs2_softmax_fwd writes them (same values, same Philox counters, pad columns 0),
 *   so fs2_softmax_bwd and the PV / backward GEMMs are unchanged.  alpha = 1/sqrt(d_k).
 *
 * fs2_attn_probs_lds_bytes(t, dk): dynamic LDS the kernel needs, or -1 when (t, dk) is not supported (dk not in
 * {32,64,128} or the strip does not fit 160 KiB: t > ~1016) -- then use fs2_gemm + fs2_softmax_fwd.
 *
 * fs2_attn_ds_bwd is the backward counterpart (dP = dO V^T kept in LDS, then the fs2_softmax_bwd arithmetic against the
 * saved probabilities): ds_out = P * (dP' - sum_j dP'_j P_j), dP' = dropout'(dO V^T); pad columns 0.  d_out rows are
 * dk contiguous bf16 at  base + b*do_batch_stride + i*do_row_stride + h*head_stride, v likewise with its strides.
 *
 * Optional second product from the same LDS strip (dk == 128 only; pass o_out / dq_out = NULL to skip):
 *   forward:  o_out[b][i][h][:]  = dropout(P)[i][:] V         (Models/modules.py:20; v has q's strides)
 *   backward: dq_out[b][i][h][:] = dq_alpha * dS[i][:] K      (k has v's strides)
 * output rows are dk contiguous bf16 at  base + b*batch_stride + i*row_stride + h*head_stride.                    */
int fs2_attn_probs_lds_bytes(int t, int dk);
int fs2_attn_probs_fwd(const void* q, const void* k, int64_t row_stride, int64_t batch_stride, int head_stride, int dk,
                       const uint8_t* key_mask, void* p_out, void* p_drop, int64_t p_batch_stride, int B, int H, int t,
                       int tp, float alpha, float p, const uint64_t* rng, uint32_t site, const void* v, void* o_out,
                       int64_t o_row_stride, int64_t o_batch_stride, void* stream);
int fs2_attn_ds_bwd(const void* d_out, int64_t do_row_stride, int64_t do_batch_stride, const void* v, int64_t v_row_stride,
                    int64_t v_batch_stride, int head_stride, int dk, const void* p_saved, int64_t p_batch_stride,
                    void* ds_out, int64_t ds_batch_stride, int B, int H, int t, int tp, float p, const uint64_t* rng,
                    uint32_t site, const void* k, void* dq_out, int64_t dq_row_stride, int64_t dq_batch_stride,
                    float dq_alpha, void* stream);

/* attention() of Models/modules.py:7-21 WITHOUT the probability tensors in HBM (hp.return_attn = False), bf16, d_k in {64, 96, 128},
 * up to 16384 keys; self-attention, and the two attentions of the autoregressive decoder (Models/layers.py:108-118 with the masks of
 * train.py:26-58): `causal` masks key j > query i like a padded key, tq != tk with separate query / key-value strides is the
 * encoder-decoder attention.
 *   forward:  o[b][i][h][:] = dropout(softmax(mask(alpha q k^T)))[i][:] v and, per query row, stats[b][h][i] = {maximum of the masked
 *             scaled scores, sum of exp(score - maximum)} (fp32 pairs).  With p > 0 it draws the dropout mask from the Philox counters of
 *             fs2_attn_probs_fwd / fs2_softmax_rect_fwd (element offset b*p_batch_stride + (h*tq + i)*tkp + key of the virtual
 *             (B,[..],H,tq,tkp) tensor: every path draws the same mask) and stashes it, one bit per probability, in keep_bits
 *             (fs2_flash_attn_keep_words_rect(B, H, tq, tk) 16-bit words; may be NULL when p == 0).
 *   backward: recomputes the probabilities from q, k and stats, reads the keep-bits and writes dq = alpha dS k, dk = alpha dS^T q,
 *             dv = dropout(P)^T d_out; aux is a (B,H,tq,4) fp32 workspace it fills itself; dbias_q/k/v (optional, H*dk floats each)
 *             receive += the column sums of dq / dk / dv (the bias gradients of the projections, Models/modules.py:49-51).
 * Key tiles whose keys are all masked are skipped (their probabilities are exp(-1e4 - max) = 0 in fp32 whenever the row has an
 * unmasked key; a row without one is computed in full and gives the reference's uniform distribution).
 *   q rows of one head = dk contiguous bf16 at  q + b*q_batch_stride + i*q_row_stride + h*head_stride  (elements); k, v rows at
 *   base + b*kv_batch_stride + j*kv_row_stride + h*head_stride; o / d_out / dq rows (queries) and dk / dv rows (keys) likewise with
 *   their own row and batch strides.                                                                                              */
typedef struct FS2FlashAttn {
    const void *q, *k, *v;
    int64_t q_row_stride, q_batch_stride, kv_row_stride, kv_batch_stride;
    int32_t head_stride, dk;
    const uint8_t* key_mask;  /* (B, tk) bytes, 0 = padded key */
    const int32_t* key_info;  /* optional (B, 3) from fs2_flash_attn_mask_info, or NULL */
    void* o;
    int64_t o_row_stride, o_batch_stride;
    float* stats;             /* (B, H, tq, 2) */
    uint16_t* keep_bits;
    int32_t pregenerated;     /* forward: read keep_bits (written by fs2_flash_attn_keep_bits) instead of drawing them */
    int32_t causal;
    int64_t p_batch_stride;
    int32_t B, H, tq, tk, tkp, reserved0;
    float alpha, p;
    const uint64_t* rng;
    uint32_t site, reserved1;
    /* backward only */
    const void* d_out;
    int64_t do_row_stride, do_batch_stride;
    float* aux;               /* (B, H, tq, 4) */
    void *dq, *dk_out, *dv_out;
    int64_t dq_row_stride, dq_batch_stride, dkv_row_stride, dkv_batch_stride;
    float *dbias_q, *dbias_k, *dbias_v;
} FS2FlashAttn;
int fs2_flash_attention_fwd(const FS2FlashAttn* d, void* stream);
int fs2_flash_attention_bwd(const FS2FlashAttn* d, void* stream);
/* The attention maps the reference returns (Models/modules.py:19-21: the probabilities AFTER dropout; Models/encoder.py:97,105 stacks them per
 * layer) written after the fact from q, k, stats and keep_bits of a finished fs2_flash_attention_fwd call with the same descriptor (set
 * `pregenerated`: no rng is read): probs[b * probs_batch_stride + (h * tq + i) * tkp + j] (bf16), keys in [tk, tkp) and every key the forward
 * skipped as exactly-zero written as 0.  The layer's arithmetic (forward and backward) never reads them. */
int fs2_flash_attention_probs(const FS2FlashAttn* d, void* probs, int64_t probs_batch_stride, void* stream);
int64_t fs2_flash_attn_keep_words_rect(int B, int H, int tq, int tk);
/* The round-2 entry points below are the self-attention, d_k = 128 case of the two calls above (q, k, v rows of one fused projection
 * tensor: common strides; tq = tk = t).                                                                                           */
int64_t fs2_flash_attn_keep_words(int B, int H, int t);
/* info[B][3]: {number of leading unmasked keys, last unmasked key + 1} of every batch row b, and in info[r][2] the batch row with
 * the r-th longest unmasked prefix (the kernels start the longest rows first).  Optional: pass it as key_info to the calls below
 * (one scan per stack instead of one per workgroup), or NULL: every workgroup scans its mask row itself, rows in batch order. */
int fs2_flash_attn_mask_info(const uint8_t* key_mask, int B, int t, int32_t* info, void* stream);
/* The reference's create_masks for the FastSpeech2 task (train_fastspeech2.py:55-82: mask = pos != pad on the (B, t) int64 positions the
 * collate function made; row stride ld >= t elements) and the row bounds / ranking above in ONE launch: mask[b][j] = pos[b][j] != pad (bytes 0 / 1), info as
 * fs2_flash_attn_mask_info writes it.  B <= 1024.  ticket: one 32-bit word, zero-filled ONCE by the caller (the launch leaves it zero
 * again), used by one stream at a time. */
int fs2_pad_mask_info(const int64_t* pos, int64_t ld, int64_t pad, int B, int t, uint8_t* mask, int32_t* info, uint32_t* ticket, void* stream);
int fs2_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, int64_t batch_stride, int head_stride,
                       const uint8_t* key_mask, const int32_t* key_info, void* o_out, int64_t o_row_stride, int64_t o_batch_stride,
                       float* stats, uint16_t* keep_bits, int pregenerated, int64_t p_batch_stride, int B, int H, int t, int tp, float alpha,
                       float p, const uint64_t* rng, uint32_t site, void* stream);
/* the same keep-bits ahead of time (then pass pregenerated = 1 to fs2_flash_attn_fwd, which reads instead of drawing them): a host
 * that knows the shapes of the next layers generates their masks on a side stream while the matrix pipes are busy elsewhere */
int fs2_flash_attn_keep_bits(uint16_t* keep_bits, int64_t p_batch_stride, int B, int H, int t, int tp, float p, const uint64_t* rng,
                             uint32_t site, void* stream);
int fs2_flash_attn_bwd(const void* q, const void* k, const void* v, int64_t row_stride, int64_t batch_stride, int head_stride,
                       const uint8_t* key_mask, const int32_t* key_info, const void* o_saved, int64_t o_row_stride,
                       int64_t o_batch_stride, const void* d_out, int64_t do_row_stride, int64_t do_batch_stride, const float* stats,
                       const uint16_t* keep_bits, float* aux, void* dq, void* dk, void* dv, int64_t g_row_stride,
                       int64_t g_batch_stride, float* dbias_q, float* dbias_k, float* dbias_v, int B, int H, int t, float alpha,
                       float p, void* stream);

/* LengthRegulator (Models/varianceadaptor.py:141-184,233-249): out[b][f] = x[b][i] for the phoneme i whose
 * duration interval contains frame f, 0 beyond sum(dur) or max_len.  starts is a [B][L+1] int32 workspace
 * (exclusive prefix sums) produced by forward and consumed by backward (segmented sum).               */
int fs2_length_regulate_fwd(const void* x, int dtype, const int64_t* dur, void* out, int32_t* starts, int B, int L,
                            int T, int d, void* stream);
int fs2_length_regulate_bwd(const void* dout, int dtype, const int32_t* starts, void* dx, int B, int L, int T, int d,
                            int accumulate, void* stream);

/* torch.bucketize + nn.Embedding adds (Models/varianceadaptor.py:100,116,123-126):
 *   out = x + Ep[#bins_p < f0] + Ee[#bins_e < energy];  idx (int32 [2][M]) saved for the scatter-add backward.
 *   f0 == NULL (hp.pitch_pred False, :93,122) or energy == NULL (hp.energy_pred False, :112,124): that term is left out, its idx row
 *   holds -1 and its bins / table pointers are not read; fs2_bucket_embed_bwd skips a NULL dEp / dEe. */
int fs2_bucket_embed_add_fwd(const void* x, int dtype, const float* f0, const float* energy, const float* pbins,
                             const float* ebins, int nbins, const float* Ep, const float* Ee, void* out, int32_t* idx,
                             int64_t M, int d, void* stream);
int fs2_bucket_embed_bwd(const void* dout, int dtype, const int32_t* idx, float* dEp, float* dEe, int64_t M, int d,
                         void* stream);

/* VariancePredictor head (Models/varianceadaptor.py:223-229): out[m] = mask[m] ? x[m].w + b : 0. */
int fs2_linear1_fwd(const void* x, int dtype, const float* w, const float* b, const uint8_t* mask, float* out,
                    int64_t M, int d, void* stream);
int fs2_linear1_bwd(const float* dout, const void* x, int dtype, const float* w, const uint8_t* mask, void* dx,
                    float* dw, float* db, int64_t M, int d, void* stream);

/* The tail of a VariancePredictor in one row pass each way (Models/varianceadaptor.py:226-231):
 *   forward:  out[m] = mask[m] ? dot(dropout(LN(x[m])), w) + b : 0        (the normalised rows are not stored)
 *   backward: = fs2_linear1_bwd followed by fs2_layernorm_bwd (relu_mask: x was relu(z), returns dz): dx, and += into dgamma,
 *             dbeta, dw, db, dcolsum (column sums of dx: the bias gradient of the convolution that produced x).                */
int fs2_ln_linear1_fwd(const void* x, int dtype, const float* gamma, const float* beta, const float* w, const float* b,
                       const uint8_t* mask, float* out, float* mean, float* rstd, int64_t M, int d, float eps, float p,
                       const uint64_t* rng, uint32_t site, void* stream);
int fs2_ln_linear1_bwd(const float* dout, const void* x, int dtype, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, const float* w, const uint8_t* mask, void* dx, float* dgamma, float* dbeta, float* dw,
                       float* db, float* dcolsum, int64_t M, int d, float p, const uint64_t* rng, uint32_t site, int relu_mask,
                       void* stream);

/* BatchNorm1d(batch statistics) + tanh + dropout (Models/postnets.py:71-73).
 *  colstats: x [M][C] -> sums[0..C) += sum, sums[C..2C) += sum of squares (also available fused in fs2_gemm)
 *  finalize: mean/rstd from the (possibly all-reduced) sums and count; running stats updated with
 *            momentum (unbiased variance), num_batches_tracked += 1.  count_dev != NULL: the row count is read
 *            from device memory (all-reduced together with the sums under data parallelism), `count` ignored
 *  fwd: y = dropout(tanh((x-mean)*rstd*gamma+beta))
 *  bwd_reduce: red[0..C) += sum dz, red[C..2C) += sum dz*xhat   (dz = dropout'(dy) * (1 - tanh^2))
 *  bwd_apply: dx = gamma*rstd*(dz - red0/count - xhat*red1/count);  dgamma += red1, dbeta += red0        */
int fs2_colstats(const void* x, int dtype, int64_t M, int C, float* sums, void* stream);
int fs2_bn_finalize(const float* sums, float count, const float* count_dev, float eps, float momentum, float* mean,
                    float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int C,
                    void* stream);
int fs2_bn_tanh_fwd(const void* x, int dtype, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, void* y, int64_t M, int C, float p, const uint64_t* rng, uint32_t site,
                    void* stream);
/* fs2_bn_finalize + fs2_bn_tanh_fwd in one launch (training forward of a post-net layer, reference Models/postnets.py:58-59,71-73):
 * mean / rstd are derived from `sums` = [sum | sum of squares] (+ count, or count_dev on the device) by every wave, stored for the
 * backward pass and folded into the running statistics by the first wave of the grid. */
int fs2_bn_stats_tanh_fwd(const void* x, int dtype, const float* sums, float count, const float* count_dev, float eps, float momentum,
                          const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, int64_t M, int C, float p, const uint64_t* rng,
                          uint32_t site, void* stream);
int fs2_bn_tanh_bwd_reduce(const void* dy, const void* x, int dtype, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, float* red, int64_t M, int C, float p,
                           const uint64_t* rng, uint32_t site, void* stream);
int fs2_bn_tanh_bwd_apply(const void* dy, const void* x, int dtype, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, const float* red, float count,
                          const float* count_dev, void* dx, float* dgamma, float* dbeta, int64_t M, int C, float p,
                          const uint64_t* rng, uint32_t site, float* dcolsum, void* stream);

/* nn.L1Loss (train_fastspeech2.py:212-259): loss[0] += sum|pred - tgt| / n.  target_mode 1: tgt = log(int64 tgt + 1).
 * backward: dpred = sign(pred - tgt) * (*gscale) / n  (gscale: device scalar, upstream gradient).          */
int fs2_l1_fwd(const void* pred, int pred_dtype, const void* target, int target_mode, int64_t n, float* loss,
               void* stream);
int fs2_l1_bwd(const void* pred, int pred_dtype, const void* target, int target_mode, int64_t n, const float* gscale,
               void* dpred, int dpred_dtype, void* stream);
/* up to 8 L1 terms in one launch each way (the trainer's five nn.L1Loss() terms, train_fastspeech2.py:212-259):
 * forward: losses[i] = mean |pred_i - target_i| for i < n_items and losses[n_items] = their sum in the order of the terms (n_items + 1
 * floats, STORED: no zero fill needed; two launches -- partial sums, then one block that adds them in a fixed order -- without float
 * atomics: the same bits on every run).  workspace: fs2_l1_multi_workspace_floats() floats of scratch (no initialisation needed), used
 * by one stream at a time;
 * backward: dpred_i = gscale[0] * sign(pred_i - target_i) / n_i  (gscale = d(loss)/d(sum of the terms), on the device). */
typedef struct {
    const void* pred;       /* n elements, pred_dtype (FS2_F32 / FS2_BF16) */
    const void* target;     /* target_mode 0: fp32; 1: int64, the target is log(target + 1) */
    void* dpred;            /* backward only: n elements, dpred_dtype */
    int64_t n;
    int32_t pred_dtype, target_mode, dpred_dtype, reserved;
} FS2L1Item;
int64_t fs2_l1_multi_workspace_floats(void);
int fs2_l1_multi_fwd(const FS2L1Item* items, int n_items, float* losses, float* workspace, void* stream);
int fs2_l1_multi_bwd(const FS2L1Item* items, int n_items, const float* gscale, void* stream);

/* Stop-token loss of the autoregressive model: F.binary_cross_entropy_with_logits(x, y, reduction='mean',
 * pos_weight) (train.py:217).  fwd adds the mean to *loss; bwd writes dx = gscale[0] / n * dl/dx. */
int fs2_bce_logits_fwd(const void* x, int x_dtype, const float* y, int64_t n, float pos_weight, float* loss, void* stream);
int fs2_bce_logits_bwd(const void* x, int x_dtype, const float* y, int64_t n, float pos_weight, const float* gscale, void* dx,
                       int dx_dtype, void* stream);
/* nn.Dropout on a flat tensor (decoder pre-net, Models/prenets.py:30-37): out = x * keep(mask) / (1 - p); with relu_gate
 * (may be NULL) the result is also zeroed where relu_gate <= 0 -- the backward through Dropout(ReLU(.)) in one pass. */
int fs2_dropout(const void* x, const void* relu_gate, void* out, int dtype, int64_t n, float p, const uint64_t* rng,
                uint32_t site, void* stream);

/* clip_grad_norm_(1.0) + Adam (train_fastspeech2.py:304-315,416), flat arenas of n fp32 elements.
 *  sqnorm: out[0] += sum x^2.   adam: coef = min(1, max_norm / (sqrt(*gsq * gscale^2) + 1e-6));
 *  hyper (device float[4]) = {lr, 1-beta1^t, 1-beta2^t, grad_scale (e.g. 1/world)}.                     */
int fs2_sqnorm(const float* x, int64_t n, float* out, void* stream);
int fs2_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, const float* gsq,
                  float beta1, float beta2, float eps, float max_norm, void* stream);
/* the same with some parameter ranges whose GRADIENT is stored as [o][j][i] (what the Conv1d weight-gradient GEMM writes) while the
 * parameter and its moments are (O,I,k): perm_segments = n_segments x {start, end, O, I, k} (int64, sorted, disjoint, starts
 * multiples of 4).  Saves the separate permute pass (fs2_permute_add) over every convolution gradient.                         */
int fs2_adam_step_perm(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, const float* gsq,
                       float beta1, float beta2, float eps, float max_norm, const int64_t* perm_segments, int n_segments,
                       void* stream);

/* Diagnostics (tools/attn_phases.py): shader-clock stamps of the LDS-strip attention kernels are written to `buf` (device memory,
 * 4 x 64-bit words per workgroup) while it is set; NULL switches the stamps off.  Not used by the product. */
void fs2_debug_attn_timer(unsigned long long* buf);

/* nbytes (a multiple of 16) of zeros at ptr (16-byte aligned): optimizer.zero_grad() of the reference loops (train_fastspeech2.py:154,
 * train.py:205) on the flat gradient arena, together with the small per-step accumulators kept behind it. */
int fs2_zero(void* ptr, int64_t nbytes, void* stream);

/* rng[1] += 1 (one step of the dropout stream). */
int fs2_rng_advance(uint64_t* rng, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FS2_HIP_H */

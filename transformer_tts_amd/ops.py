"""Tensor-level bindings of the C ABI in include/fs2_hip.h (ctypes -> libfs2_hip.so).

PyTorch is used only as the owner of device memory and of the current HIP stream: every function
takes CUDA(=HIP) tensors, passes raw pointers / sizes / strides to the extern "C" launchers and
returns without synchronising.  There is NO CPU fallback: if the library is missing or a tensor is
not on the GPU the call raises.
"""
import ctypes
import os

import torch

from . import bounds as _bounds
from . import build as _build

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}

_lib = None


class FS2CastDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dld", ctypes.c_int64), ("O", ctypes.c_int32),
                ("I", ctypes.c_int32), ("k", ctypes.c_int32), ("mode", ctypes.c_int32)]


class FS2L1Item(ctypes.Structure):
    _fields_ = [("pred", ctypes.c_void_p), ("target", ctypes.c_void_p), ("dpred", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("pred_dtype", ctypes.c_int32), ("target_mode", ctypes.c_int32), ("dpred_dtype", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class FS2FlashAttn(ctypes.Structure):
    _fields_ = [("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("q_row_stride", ctypes.c_int64), ("q_batch_stride", ctypes.c_int64), ("kv_row_stride", ctypes.c_int64),
                ("kv_batch_stride", ctypes.c_int64), ("head_stride", ctypes.c_int32), ("dk", ctypes.c_int32),
                ("key_mask", ctypes.c_void_p), ("key_info", ctypes.c_void_p), ("o", ctypes.c_void_p),
                ("o_row_stride", ctypes.c_int64), ("o_batch_stride", ctypes.c_int64), ("stats", ctypes.c_void_p),
                ("keep_bits", ctypes.c_void_p), ("pregenerated", ctypes.c_int32), ("causal", ctypes.c_int32),
                ("p_batch_stride", ctypes.c_int64), ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("tq", ctypes.c_int32),
                ("tk", ctypes.c_int32), ("tkp", ctypes.c_int32), ("reserved0", ctypes.c_int32), ("alpha", ctypes.c_float),
                ("p", ctypes.c_float), ("rng", ctypes.c_void_p), ("site", ctypes.c_uint32), ("reserved1", ctypes.c_uint32),
                ("d_out", ctypes.c_void_p), ("do_row_stride", ctypes.c_int64), ("do_batch_stride", ctypes.c_int64),
                ("aux", ctypes.c_void_p), ("dq", ctypes.c_void_p), ("dk_out", ctypes.c_void_p), ("dv_out", ctypes.c_void_p),
                ("dq_row_stride", ctypes.c_int64), ("dq_batch_stride", ctypes.c_int64), ("dkv_row_stride", ctypes.c_int64),
                ("dkv_batch_stride", ctypes.c_int64), ("dbias_q", ctypes.c_void_p), ("dbias_k", ctypes.c_void_p),
                ("dbias_v", ctypes.c_void_p)]


class FS2Gemm(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p), ("relu_mask", ctypes.c_void_p), ("colstats", ctypes.c_void_p),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64), ("ldr", ctypes.c_int64),
        ("ldm", ctypes.c_int64),
        ("sA1", ctypes.c_int64), ("sA2", ctypes.c_int64), ("sB1", ctypes.c_int64), ("sB2", ctypes.c_int64),
        ("sC1", ctypes.c_int64), ("sC2", ctypes.c_int64),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("Kb", ctypes.c_int32),
        ("a_kmajor", ctypes.c_int32), ("b_kmajor", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("c_dtype", ctypes.c_int32), ("res_dtype", ctypes.c_int32), ("relu", ctypes.c_int32),
        ("accumulate", ctypes.c_int32), ("split_k", ctypes.c_int32), ("batch1", ctypes.c_int32),
        ("batch2", ctypes.c_int32), ("conv", ctypes.c_int32), ("taps", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("seq_len", ctypes.c_int32), ("alpha", ctypes.c_float), ("colstats_mode", ctypes.c_int32),
        ("tile_order", ctypes.c_int32),
        ("scale_a", ctypes.c_void_p), ("scale_b", ctypes.c_void_p),
        ("q8", ctypes.c_void_p), ("q8_state", ctypes.c_void_p), ("q8_prev", ctypes.c_void_p), ("q8_bf8", ctypes.c_int32),
        ("q8_reserved", ctypes.c_int32),
    ]


# name -> argtypes (restype is always int unless noted); mirrors include/fs2_hip.h one to one
_P, _I, _L, _F, _U32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32
SIGNATURES = {
    "fs2_gemm": [ctypes.POINTER(FS2Gemm), _P],
    "fs2_gemm_last_tile": [],
    "fs2_gemm_last_splits": [],
    "fs2_amax": [_P, _I, _L, _P, _P],
    "fs2_quantize_fp8": [_P, _I, _P, _I, _L, _P, _P],
    "fs2_cast_permute": [_P, _P, _I, _I, _I, _L, _I, _I, _P],
    "fs2_permute_add": [_P, _P, _I, _I, _I, _I, _P],
    "fs2_cast": [_P, _I, _P, _I, _L, _P],
    "fs2_add_cast": [_P, _P, _P, _I, _L, _P],
    "fs2_copy_batched": [_P, _P, _P, _I, _P],
    "fs2_cast_permute_batched": [_P, _I, _I, _P],
    "fs2_onehot": [_P, _P, _I, _L, _I, _P],
    "fs2_colsum_segmented": [_P, _I, _L, _I, _L, _P, _I, _L, _P],
    "fs2_colsum": [_P, _I, _L, _I, _L, _P, _P],
    "fs2_embedding_fwd": [_P, _P, _P, _I, _L, _I, _P],
    "fs2_embedding_bwd": [_P, _P, _I, _P, _L, _I, _L, _P],
    "fs2_pe_add_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _F, _P, _U32, _P],
    "fs2_pe_add_bwd": [_P, _P, _P, _I, _P, _I, _I, _I, _F, _P, _U32, _P, _P],
    "fs2_pe_add_ln_fwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _F, _F, _P, _U32, _P],
    "fs2_ln_pe_add_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _F, _P, _U32, _P],
    "fs2_layernorm_fwd": [_P, _I, _P, _P, _P, _I, _P, _P, _L, _I, _F, _F, _P, _U32, _P],
    "fs2_layernorm_bwd": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _L, _I, _F, _P, _U32, _I, _I, _P, _P],
    "fs2_add_ln_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _P, _U32, _P],
    "fs2_add_ln_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _P, _P],
    "fs2_ffn_ln_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _F, _F, _P, _U32, _P],
    "fs2_ffn_ln_bwd": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _P, _P],
    "fs2_splitk_finish": [_P, _L, _I, _P, _P, _I, _L, _I, _P, _I, _L, _P],
    "fs2_splitk_reduce": [_P, _I, _L, _L, _L, _I, _P, _P, _I, _L, _I, _P, _I, _L, _P],
    "fs2_q8_next": [_P, _P, _P, _I],
    "fs2_quantize_fp8_repair": [_P, _I, _P, _I, _L, _P, _P, _P],
    "fs2_quantize_fp8_batched": [_P, _I, _I, _I, _P],
    "fs2_wgrad_sliced": [ctypes.POINTER(FS2Gemm), _P, _L, _P, _P],      # returns int64
    "fs2_wgrad_plan": [ctypes.POINTER(FS2Gemm)],
    "fs2_wgrad_grouped": [_P, _I, _P, _L, _P, _P],      # returns int64
    "fs2_wgrad_reduce": [_P, _I, _P],
    "fs2_softmax_fwd": [_P, _P, _I, _P, _I, _I, _I, _I, _L, _F, _P, _U32, _P],
    "fs2_attn_probs_lds_bytes": [_I, _I],
    "fs2_attn_probs_fwd": [_P, _P, _L, _L, _I, _I, _P, _P, _P, _L, _I, _I, _I, _I, _F, _F, _P, _U32, _P, _P, _L, _L, _P],
    "fs2_flash_attn_keep_words": [_I, _I, _I],          # returns int64
    "fs2_flash_attn_keep_words_rect": [_I, _I, _I, _I],  # returns int64
    "fs2_flash_attention_fwd": [ctypes.POINTER(FS2FlashAttn), _P],
    "fs2_flash_attention_bwd": [ctypes.POINTER(FS2FlashAttn), _P],
    "fs2_flash_attention_probs": [ctypes.POINTER(FS2FlashAttn), _P, _L, _P],
    "fs2_flash_attn_mask_info": [_P, _I, _I, _P, _P],
    "fs2_pad_mask_info": [_P, _L, _L, _I, _I, _P, _P, _P, _P],
    "fs2_flash_attn_fwd": [_P, _P, _P, _L, _L, _I, _P, _P, _P, _L, _L, _P, _P, _I, _L, _I, _I, _I, _I, _F, _F, _P, _U32, _P],
    "fs2_flash_attn_keep_bits": [_P, _L, _I, _I, _I, _I, _F, _P, _U32, _P],
    "fs2_flash_attn_bwd": [_P, _P, _P, _L, _L, _I, _P, _P, _P, _L, _L, _P, _L, _L, _P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _I, _I, _I, _F,
                           _F, _P],
    "fs2_attn_ds_bwd": [_P, _L, _L, _P, _L, _L, _I, _I, _P, _L, _P, _L, _I, _I, _I, _I, _F, _P, _U32, _P, _P, _L, _L, _F, _P],
    "fs2_softmax_bwd": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _P, _U32, _P],
    "fs2_length_regulate_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "fs2_length_regulate_bwd": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "fs2_bucket_embed_add_fwd": [_P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _P],
    "fs2_bucket_embed_bwd": [_P, _I, _P, _P, _P, _L, _I, _P],
    "fs2_linear1_fwd": [_P, _I, _P, _P, _P, _P, _L, _I, _P],
    "fs2_linear1_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _P],
    "fs2_colstats": [_P, _I, _L, _I, _P, _P],
    "fs2_bn_finalize": [_P, _F, _P, _F, _F, _P, _P, _P, _P, _P, _I, _P],
    "fs2_bn_tanh_fwd": [_P, _I, _P, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _P],
    "fs2_bn_stats_tanh_fwd": [_P, _I, _P, _F, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _P],
    "fs2_bn_tanh_bwd_reduce": [_P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _P],
    "fs2_bn_tanh_bwd_apply": [_P, _P, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _P, _P],
    "fs2_softmax_rect_fwd": [_P, _P, _I, _P, _I, _I, _I, _I, _I, _L, _I, _F, _P, _U32, _P],
    "fs2_softmax_rect_bwd": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _F, _P, _U32, _P],
    "fs2_bce_logits_fwd": [_P, _I, _P, _L, _F, _P, _P],
    "fs2_bce_logits_bwd": [_P, _I, _P, _L, _F, _P, _P, _I, _P],
    "fs2_dropout": [_P, _P, _P, _I, _L, _F, _P, _U32, _P],
    "fs2_l1_fwd": [_P, _I, _P, _I, _L, _P, _P],
    "fs2_l1_bwd": [_P, _I, _P, _I, _L, _P, _P, _I, _P],
    "fs2_sqnorm": [_P, _L, _P, _P],
    "fs2_ln_linear1_fwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _P, _U32, _P],
    "fs2_ln_linear1_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _I, _P],
    "fs2_ffn_tail_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _P, _U32, _U32, _P],
    "fs2_ffn_tail_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _U32, _U32, _P],
    "fs2_l1_multi_workspace_floats": [],                 # returns int64
    "fs2_l1_multi_fwd": [_P, _I, _P, _P, _P],
    "fs2_l1_multi_bwd": [_P, _I, _P, _P],
    "fs2_adam_step": [_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _P],
    "fs2_adam_step_perm": [_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _P, _I, _P],
    "fs2_rng_advance": [_P, _P],
    "fs2_zero": [_P, _L, _P],
    "fs2_debug_attn_timer": [_P],      # diagnostics (returns void)
}


def library_path():
    return _build.LIB


def lib():
    """Load libfs2_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m transformer_tts_amd.build` "
                               "(hipcc --offload-arch=gfx950).  transformer_tts_amd has no CPU fallback.")
        l = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        l.fs2_debug_attn_timer.restype = None
        l.fs2_last_error.restype = ctypes.c_char_p
        l.fs2_flash_attn_keep_words.restype = ctypes.c_int64
        l.fs2_flash_attn_keep_words_rect.restype = ctypes.c_int64
        l.fs2_l1_multi_workspace_floats.restype = ctypes.c_int64
        l.fs2_wgrad_sliced.restype = ctypes.c_int64
        l.fs2_wgrad_grouped.restype = ctypes.c_int64
        l.fs2_abi_version.restype = ctypes.c_int
        _lib = l
    return _lib


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib().fs2_last_error().decode()}")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("transformer_tts_amd ops need GPU tensors (no CPU fallback)")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    # (the raw handle of the current stream without building a torch.cuda.Stream object: ~1.5 us less per launch on the eager path)
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _dt(t):
    return _DT[t.dtype]


def _c(t):
    assert t.is_contiguous(), "contiguous tensor required"
    return t


class Rng:
    """Device-resident {seed, offset} pair of the dropout streams; `advance()` once per step."""

    def __init__(self, seed, device):
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)

    def advance(self):
        _check(lib().fs2_rng_advance(_p(self.state), _stream()), "fs2_rng_advance")


def _rng_ptr(rng, p):
    if p > 0.0 and rng is None:
        raise RuntimeError("dropout p > 0 needs an ops.Rng")
    return _p(rng.state) if rng is not None else None


# ------------------------------------------------------------------------------------------------ GEMM family
def _ld(t):
    """row stride of a matrix view whose last dim is contiguous"""
    assert t.stride(-1) == 1 or t.shape[-1] == 1, f"last dim must be contiguous, strides {t.stride()}"
    return t.stride(-2)


def _gemm_call(g):
    if _bounds.ENABLED:
        _bounds.check_gemm(g)
    _check(lib().fs2_gemm(ctypes.byref(g), _stream()), "fs2_gemm")


def _epilogue(g, out, bias, relu, residual, relu_mask, colstats, alpha, colsum=None):
    g.C, g.ldc, g.c_dtype = _p(out), _ld(out), _dt(out)
    g.bias = _p(bias)
    g.relu = int(relu)
    g.alpha = float(alpha)
    if residual is not None:
        g.residual, g.ldr, g.res_dtype = _p(residual), _ld(residual), _dt(residual)
    if relu_mask is not None:
        g.relu_mask, g.ldm = _p(relu_mask), _ld(relu_mask)
    if colstats is not None:
        g.colstats = _p(colstats)
    elif colsum is not None:        # column sums only (bias gradient of the layer that produced this GEMM's input)
        g.colstats, g.colstats_mode = _p(colsum), 1


_splitk_scratch = {}


def _splitk_plan(M, N, k_total, g, relu_mask, colstats, colsum, alpha):
    """split-K factor for a forward / data-gradient product with too few output tiles to fill 256 CUs and a long
    reduction (config 2: the 6144 x 256 x (9 x 1024) encoder convolutions: 48 tiles of 128 x 256), or 1.  The product
    then writes fp32 partial sums into one workspace slice per split (FS2Gemm.accumulate = 2: plain stores, no atomics)
    and fs2_splitk_reduce adds the slices and applies bias / ReLU / residual / cast.  FS2_SPLITK_FWD=0 switches it off
    (A/B measurements)."""
    if g.dtype != BF16 or relu_mask is not None or colstats is not None or colsum is not None or alpha != 1.0:
        return 1
    if os.environ.get("FS2_SPLITK_FWD", "1") == "0" or N % 8 != 0 or N > 2048:
        return 1
    tiles = ((M + 127) // 128) * ((N + 255) // 256)
    stages = (k_total + 63) // 64
    if tiles > 96 or stages < 48:
        return 1
    if os.environ.get("FS2_SPLITK_N"):             # (A/B measurements)
        return int(os.environ["FS2_SPLITK_N"])
    # about one work item per CU (the kernel picks its row-slab height for the split count), >= 12 stages per split
    return int(max(1, min(256 // tiles, stages // 12, 16)))


def _splitk_run(g, M, N, split, out, bias, relu, residual):
    key = (out.device, M, N, split)
    slices = _splitk_scratch.get(key)
    if slices is None:
        slices = _splitk_scratch[key] = torch.empty((split, M, N), dtype=torch.float32, device=out.device)
    g.split_k, g.accumulate = split, 2
    g.sC1 = M * N
    _epilogue(g, slices[0], None, False, None, None, None, 1.0)
    _gemm_call(g)
    nsplit = lib().fs2_gemm_last_splits()
    _check(lib().fs2_splitk_reduce(_p(slices), nsplit, M * N, N, M, N, _p(bias), _p(residual),
                                   _dt(residual) if residual is not None else 0, _ld(residual) if residual is not None else 0,
                                   int(relu), _p(out), _dt(out), _ld(out), _stream()), "fs2_splitk_reduce")
    return out


# ---- fp8 operand mode (BASELINE.json configs[4]): per-tensor current scaling, e4m3 for activations / weights, e5m2 for gradients
FP8, BF8_FP8 = 2, 3
FP8_MODE = {"on": False, "backward": False}      # set by Runtime (hp.fp8) and by the autograd Functions' backward


class fp8_backward:
    """context of a hand-written backward: row-major products inside it quantise their A operand (a gradient) to e5m2"""

    def __enter__(self):
        self.prev = FP8_MODE["backward"]
        FP8_MODE["backward"] = True

    def __exit__(self, *exc):
        FP8_MODE["backward"] = self.prev


_FP8_STATES = {"buf": None, "prev": None, "idx": 0}


def fp8_begin_step(device, slots=4096):
    """one zero-fill for all {amax, 1/scale} pairs of a training step (instead of one tiny fill per quantised tensor).  The states of
    the step before are kept in a second buffer (copied, so that both buffers keep their addresses inside a captured graph): the k-th
    quantisation of a step is the same call site as the k-th of the previous one, and its old amax is the speculation of the
    producers that write the fp8 copy of their output themselves (FS2Gemm.q8)."""
    buf = _FP8_STATES["buf"]
    if buf is None or buf.device != device:
        _FP8_STATES["buf"] = torch.zeros((slots, 2), dtype=torch.float32, device=device)
        _FP8_STATES["prev"] = torch.zeros((slots, 2), dtype=torch.float32, device=device)
    else:
        _FP8_STATES["prev"].copy_(buf)
        buf.zero_()
    _FP8_STATES["idx"] = 0


def _fp8_state(device, with_prev=False):
    buf, prev, i = _FP8_STATES["buf"], _FP8_STATES["prev"], _FP8_STATES["idx"]
    if buf is None or buf.device != device or i >= buf.shape[0]:
        st = torch.zeros(2, dtype=torch.float32, device=device)
        return (st, None) if with_prev else st
    _FP8_STATES["idx"] = i + 1
    return (buf[i], prev[i]) if with_prev else buf[i]


def quantize_fp8(x, bf8=False):
    """x (bf16 / fp32, contiguous rows) -> (uint8 tensor of the same shape holding OCP e4m3 (or e5m2) codes, state) with
    state = device float[2] {amax, 1/scale}; scale is the largest power of two that keeps amax * scale below 2^8 (2^15)."""
    x = _c(x)
    n = x.numel()
    state = _fp8_state(x.device)
    q = torch.empty(((n + 15) // 16 * 16,), dtype=torch.uint8, device=x.device)
    _check(lib().fs2_amax(_p(x), _dt(x), n, _p(state), _stream()), "fs2_amax")
    _check(lib().fs2_quantize_fp8(_p(x), _dt(x), _p(q), int(bf8), n, _p(state), _stream()), "fs2_quantize_fp8")
    return q[:n].view(x.shape), state


class FS2QuantDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("state", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("block_begin", ctypes.c_int32), ("nblocks", ctypes.c_int32)]


_FP8_W = {}            # data_ptr of a bf16 weight shadow -> (fp8 codes, state {amax, 1/scale}, numel): filled by fp8_quantize_shadows
_FP8_W_TABLES = {}


def fp8_quantize_shadows(shadows):
    """e4m3 codes + scale of every weight shadow in `shadows` (contiguous bf16 tensors) in two launches and one memset: what
    _fp8_operands would otherwise do with two launches per weight and product.  The Runtime calls it after every batched shadow
    refresh; a shadow rewritten through ops.cast_permute drops out of the cache."""
    shadows = [t for t in shadows if t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() >= 16]
    if not shadows:
        return
    key = tuple(t.data_ptr() for t in shadows)
    tab = _FP8_W_TABLES.get(key)
    if tab is None:
        dev = shadows[0].device
        states = torch.zeros((len(shadows), 2), dtype=torch.float32, device=dev)
        codes = [torch.empty(((t.numel() + 15) // 16 * 16,), dtype=torch.uint8, device=dev) for t in shadows]
        arr = (FS2QuantDesc * len(shadows))()
        blocks = 0
        for i, (d, t, q) in enumerate(zip(arr, shadows, codes)):
            d.src, d.dst, d.state, d.n = t.data_ptr(), q.data_ptr(), states[i].data_ptr(), t.numel()
            d.block_begin, d.nblocks = blocks, max(1, -(-t.numel() // 32768))
            blocks += d.nblocks
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(dev)
        tab = _FP8_W_TABLES[key] = (table, len(shadows), blocks, states, codes, list(shadows))
    table, n, blocks, states, codes, keep = tab
    states.zero_()
    _check(lib().fs2_quantize_fp8_batched(_p(table), n, blocks, 0, _stream()), "fs2_quantize_fp8_batched")
    for i, t in enumerate(keep):
        _FP8_W[t.data_ptr()] = (codes[i][:t.numel()].view(t.shape), states[i], t.numel())


def _fp8_eligible(g, M, N, K_total_row, x, w, residual=None, relu_mask=None, stats=False):
    """products the fp8 mode takes: bf16 operands, contiguous rows with K a multiple of 16, N a multiple of 8, enough rows for
    the 16-wave kernel to make sense, and an epilogue combination that kernel is compiled for (gemm_ring.hip epi_compiled)"""
    if not (FP8_MODE["on"] and g.dtype == BF16 and M >= 1024 and N % 8 == 0 and g.K % 16 == 0 and x.stride(-1) == 1 and
            w.stride(-1) == 1 and x.stride(-2) == x.shape[-1] and w.stride(-2) == w.shape[-1]):
        return False
    if stats and (N > 2048 or residual is not None):
        return False
    if residual is not None and relu_mask is not None and residual.dtype != torch.float32:
        return False
    # the refusals of fs2_gemm_ring_try itself (the product must stay on bf16 operands then, not fail after its
    # operands were quantised): 32-bit byte offsets over M + 512 rows of C / mask / residual / A and N + 512 rows of B, conv padding
    rows = M + 512
    if rows * N * 4 >= 0x7FFFFFF0 or rows * K_total_row >= 0x7FFFFFF0 or (N + 512) * max(1, g.taps if g.conv == 1 else 1) * K_total_row >= 0x7FFFFFF0:
        return False
    if g.conv == 1 and not (0 <= g.pad <= g.taps):
        return False
    return True


FP8_FUSED_OUT = os.environ.get("FS2_FP8_Q8", "1") != "0"     # producers write the fp8 copy of their output (FS2Gemm.q8)
FP8_WGRAD = os.environ.get("FS2_FP8_WGRAD", "1") != "0"      # weight gradients multiply the fp8 copies of dY (e5m2) and X (e4m3)


def _wgrad_fp8(g, dy, x):
    """the descriptor of a weight-gradient product with its operands swapped for their fp8 copies -- when the fp8 mode is on, both
    tensors carry one (dY in e5m2 from the data-gradient product or its producer, X in e4m3 from the forward) and the product runs in
    the uniform k-split form (fs2_wgrad_plan); else None"""
    if not (FP8_MODE["on"] and FP8_WGRAD) or dy is None or x is None:
        return None
    a, b = getattr(dy, "_fs2_q8", None), getattr(x, "_fs2_q8", None)
    if a is None or b is None or not a[2] or b[2] or a[0].numel() != dy.numel() or b[0].numel() != x.numel():
        return None
    g2 = FS2Gemm.from_buffer_copy(g)
    g2.A, g2.B, g2.dtype = _p(a[0]), _p(b[0]), BF8_FP8
    g2.scale_a, g2.scale_b = _p(a[1][1:]), _p(b[1][1:])
    if lib().fs2_wgrad_plan(ctypes.byref(g2)) == 0:       # (1: uniform k-split; 2: balanced stream / > 256 output tiles)
        return None
    return g2, (a, b)


def _fp8_q8_request(g, out2, relu_mask, colstats, colsum, residual):
    """ask the product described by g (fp8 operands already set) to write the fp8 copy of its bf16 output `out2` as well; returns the
    handle the consumer picks up (out._fs2_q8) or None when the epilogue / shape has no such form"""
    M, N = out2.shape
    if not FP8_FUSED_OUT or out2.dtype != torch.bfloat16 or N % 16 != 0 or out2.stride(0) != N or out2.stride(1) != 1 or residual is not None:
        return None
    if colstats is not None or (relu_mask is None) != (colsum is None):       # plain (bias / ReLU) or mask + column sums
        return None
    state, prev = _fp8_state(out2.device, with_prev=True)
    if prev is None:
        return None
    q = torch.empty((M * N + 15) // 16 * 16, dtype=torch.uint8, device=out2.device)
    bf8 = bool(FP8_MODE["backward"])
    g.q8, g.q8_state, g.q8_prev, g.q8_bf8 = _p(q), _p(state), _p(prev), int(bf8)
    return (q, state, prev, bf8, M * N)


def _fp8_q8_finish(handle, out2):
    q, state, prev, bf8, n = handle
    _check(lib().fs2_quantize_fp8_repair(_p(out2), BF16, _p(q), int(bf8), n, _p(state), _p(prev), _stream()), "fs2_quantize_fp8_repair")
    return (q[:n].view(out2.shape), state, bf8)


class _RowQ8:
    """fp8 copy of the bf16 row output `y` of one of the four LayerNorm-family kernels that produce GEMM operands, written by the kernel
    itself: `with _RowQ8(y) as r: <launch>` arms fs2_q8_next before the launch and runs the repair launch behind it"""

    def __init__(self, y):
        self.y, self.handle = y, None
        if FP8_MODE["on"] and FP8_FUSED_OUT and y.is_cuda and y.dtype == torch.bfloat16 and y.is_contiguous() and y.numel() >= 1024 * 64:
            state, prev = _fp8_state(y.device, with_prev=True)
            if prev is not None:
                n = y.numel()
                q = torch.empty((n + 15) // 16 * 16, dtype=torch.uint8, device=y.device)
                self.handle = (q, state, prev, bool(FP8_MODE["backward"]), n)

    def __enter__(self):
        if self.handle is not None:
            q, state, prev, bf8, n = self.handle
            _check(lib().fs2_q8_next(_p(q), _p(state), _p(prev), int(bf8)), "fs2_q8_next")
        return self

    def __exit__(self, *exc):
        if self.handle is not None and exc[0] is None:
            self.y._fs2_q8 = _fp8_q8_finish(self.handle, self.y)


def view2d(x, M, d):
    """x.view(M, d) that keeps the fp8 copy a producer attached to x (a plain .view() makes a new tensor object)"""
    y = x.view(M, d)
    pre = getattr(x, "_fs2_q8", None)
    if pre is not None:
        y._fs2_q8 = (pre[0].view(M, d), pre[1], pre[2])
    return y


def _q8_of(x):
    """the fp8 copy kept with x, or with the tensor x is a whole-tensor alias of (an autograd Function that returns its input hands out
    `x.view_as(x)`: a new tensor object over the same elements)"""
    pre = getattr(x, "_fs2_q8", None)
    if pre is None:
        base = x._base
        if base is not None and base.numel() == x.numel() and base.is_contiguous() and x.is_contiguous() \
                and base.storage_offset() == x.storage_offset():
            pre = getattr(base, "_fs2_q8", None)
            if pre is not None:
                pre = (pre[0].view(x.shape), pre[1], pre[2])
    return pre


def _fp8_operands(g, x2, w, pre=None):
    """quantise both operands of a row-major product and point the descriptor at them (pre: the fp8 copy the producer of x2 wrote)"""
    if pre is not None and pre[2] == bool(FP8_MODE["backward"]) and pre[0].numel() == x2.numel():
        xq, sx = pre[0].view(x2.shape), pre[1]
    else:
        xq, sx = quantize_fp8(x2, bf8=FP8_MODE["backward"])
        if FP8_WGRAD:       # the weight gradient of this layer multiplies the same tensor: keep the copy with it (and with its base)
            x2._fs2_q8 = (xq, sx, bool(FP8_MODE["backward"]))
            base = x2._base
            if base is not None and base.numel() == x2.numel() and base.is_contiguous():
                base._fs2_q8 = (xq.view(base.shape), sx, bool(FP8_MODE["backward"]))
    hit = _FP8_W.get(w.data_ptr())
    if hit is not None and hit[2] == w.numel() and hit[0].shape == w.shape:
        wq, sw = hit[0], hit[1]
    else:
        wq, sw = quantize_fp8(w, bf8=False)
    g.A, g.B, g.lda, g.ldb = _p(xq), _p(wq), xq.stride(-2), wq.stride(-2)
    g.dtype = BF8_FP8 if FP8_MODE["backward"] else FP8
    g.scale_a, g.scale_b = _p(sx[1:]), _p(sw[1:])
    return (xq, sx, wq, sw)          # keep alive until the launch is enqueued (stream-ordered allocator reuse is safe after)


def linear(x, w, bias=None, relu=False, residual=None, relu_mask=None, colstats=None, out=None, out_dtype=None,
           alpha=1.0, colsum=None, q8_out=False):
    """out[M,N] = alpha * x[M,K] @ w[N,K]^T (+bias)(ReLU)(*mask>0)(+residual).  x, w: same dtype (f32 | bf16).
    q8_out (fp8 operand mode only): the output feeds another row-major product -- let the epilogue write its fp8 copy (out._fs2_q8)."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or x.dtype, device=x.device)
    g = FS2Gemm()
    g.A, g.B, g.lda, g.ldb = _p(x), _p(w), _ld(x), _ld(w)
    g.M, g.N, g.K, g.dtype = M, N, K, _dt(x)
    g.split_k = g.batch1 = g.batch2 = 1
    split = _splitk_plan(M, N, K, g, relu_mask, colstats, colsum, alpha)
    if split > 1:
        return _splitk_run(g, M, N, split, out, bias, relu, residual)
    keep = _fp8_operands(g, x, w, _q8_of(x)) \
        if _fp8_eligible(g, M, N, K, x, w, residual, relu_mask, colstats is not None or colsum is not None) else None
    _epilogue(g, out, bias, relu, residual, relu_mask, colstats, alpha, colsum)
    handle = _fp8_q8_request(g, out, relu_mask, colstats, colsum, residual) if (q8_out and keep is not None and alpha == 1.0) else None
    _gemm_call(g)
    if handle is not None:
        out._fs2_q8 = _fp8_q8_finish(handle, out)
    del keep
    return out


def conv(x, w, taps, pad, bias=None, relu=False, residual=None, relu_mask=None, colstats=None, out=None,
         out_dtype=None, colsum=None, q8_out=False):
    """Conv1d over time as implicit GEMM: x (B,t,C) channels-last, w (N, taps*C) [n][j*C + c];
    out[b,t,n] = sum_{j,c} x[b, t+j-pad, c] * w[n, j*C+c]  (zero outside the sequence)."""
    B, t, C = x.shape
    N = w.shape[0]
    assert w.shape[1] == taps * C
    if out is None:
        out = torch.empty((B, t, N), dtype=out_dtype or x.dtype, device=x.device)
    x2, o2 = x.view(B * t, C), out.view(B * t, N)
    g = FS2Gemm()
    g.A, g.B, g.lda, g.ldb = _p(x2), _p(w), _ld(x2), _ld(w)
    g.M, g.N, g.K, g.dtype = B * t, N, C, _dt(x)
    g.split_k = g.batch1 = g.batch2 = 1
    g.conv, g.taps, g.pad, g.seq_len = 1, taps, pad, t
    r2 = residual.view(B * t, N) if residual is not None else None
    m2 = relu_mask.view(B * t, N) if relu_mask is not None else None
    split = _splitk_plan(B * t, N, taps * C, g, relu_mask, colstats, colsum, 1.0)
    if split > 1:
        _splitk_run(g, B * t, N, split, o2, bias, relu, r2)
        return out
    keep = _fp8_operands(g, x2, w, _q8_of(x)) \
        if _fp8_eligible(g, B * t, N, C, x2, w, residual, relu_mask, colstats is not None or colsum is not None) else None
    _epilogue(g, o2, bias, relu, r2, m2, colstats, 1.0, colsum)
    handle = _fp8_q8_request(g, o2, m2, colstats, colsum, r2) if (q8_out and keep is not None) else None
    _gemm_call(g)
    if handle is not None:
        out._fs2_q8 = _fp8_q8_finish(handle, o2)
    del keep
    return out


class FS2WgradPart(ctypes.Structure):
    _fields_ = [("ws", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("ldc", ctypes.c_int64), ("sC1", ctypes.c_int64), ("sC2", ctypes.c_int64),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("tilesM", ctypes.c_int32), ("tilesN", ctypes.c_int32),
                ("splits", ctypes.c_int32), ("n2", ctypes.c_int32), ("nbatch", ctypes.c_int32), ("block_begin", ctypes.c_int32),
                ("alpha", ctypes.c_float), ("reserved", ctypes.c_int32), ("scale_a", ctypes.c_void_p), ("scale_b", ctypes.c_void_p)]


class _WgradSlices:
    """Partial tiles of the long-reduction weight gradients (fs2_wgrad_sliced / fs2_wgrad_grouped) and the products waiting for their
    reduce.  One workspace per device, reused from offset 0 after every reduce (the stream is in order), so the same ~64 MiB stay hot.
    Deferred products (the models' backward, defer=True) are not even launched at once: up to four of them -- the weight gradients of
    one layer -- go into ONE launch when wgrad_flush() is called (or a fifth arrives); their operands are kept alive until then."""
    FLOATS = 128 << 20         # 512 MiB: the partial tiles of a whole backward pass (one reduce at its end) -- ~16-25 MiB per product

    def __init__(self):
        self.ws = {}
        self.off = 0
        self.parts = []
        self.keep = []
        self.spans = []
        self.pending = []      # (descriptor copy, gradient tensor, tensors to keep alive)
        self.device = None
        self.enabled = os.environ.get("FS2_WGRAD_SLICED", "1") != "0"
        self.group = os.environ.get("FS2_WGRAD_GROUP", "1") != "0"
        self.on_group = None   # measurement hook: callable(descriptors, launch) that must call launch() once (bench.py)
        self.on_one = None     # measurement hook for a product of a group launched on its own: callable(descriptor, launch) -> launch()

    def _ws(self, device):
        ws = self.ws.get(device)
        if ws is None:
            ws = self.ws[device] = torch.empty(self.FLOATS, dtype=torch.float32, device=device)
        return ws

    def run(self, g, out, defer, extra_bytes=0, keep=()):
        if not self.enabled or not out.is_cuda:
            return False
        # two products adding into the same gradient (the mel Linear sees two gradient terms) must not share a reduce launch
        lo = out.data_ptr()
        hi = lo + 4 * (sum((n - 1) * st for n, st in zip(out.shape, out.stride())) + 1) + extra_bytes
        if (self.parts or self.pending) and (self.device != out.device or any(lo < b and a < hi for a, b in self.spans)):
            self.flush()
        self.device = out.device
        if defer and self.group:
            self.pending.append((FS2Gemm.from_buffer_copy(g), out, keep))      # (the fp8 copies of the operands are looked up at launch)
            self.spans.append((lo, hi))
            if len(self.pending) >= 4:
                self._launch_pending()
            if len(self.parts) >= 36:
                self.flush()
            return True
        f8 = _wgrad_fp8(g, *keep) if len(keep) == 2 else None
        if not self._one(f8[0] if f8 else g, out):
            return False
        self.spans.append((lo, hi))
        if not defer or len(self.parts) >= 36:
            self.flush()
        return True

    def _one(self, g, out):
        """one product with its partial tiles in the workspace; False: not in that form (the caller uses fs2_gemm)"""
        ws = self._ws(out.device)
        part = FS2WgradPart()
        if _bounds.ENABLED:
            _bounds.check_gemm(g, "fs2_wgrad_sliced")
        def launch():
            return lib().fs2_wgrad_sliced(ctypes.byref(g), ws.data_ptr() + 4 * self.off, self.FLOATS - self.off, ctypes.byref(part), _stream())
        for attempt in range(2):
            used = self.on_one(g, launch) if self.on_one is not None else launch()
            if used < 0:
                _check(int(used), "fs2_wgrad_sliced")
            if used > 0 or not self.parts or attempt == 1:
                break
            self._reduce()         # perhaps the workspace was full: retry from offset 0
        if used <= 0:
            return False
        if part.splits != 0:            # (0: an fp8 product flushed with float atomics, complete; < 0: balanced stream, partial tiles)
            if _bounds.ENABLED:
                _bounds.check_part(part, ws.data_ptr() + 4 * self.off, 4 * int(used), "fs2_wgrad_sliced part")
            self.off += int(used)
            self.parts.append(part)
            self.keep.append(out)
        return True

    def _launch_pending(self):
        pend, self.pending = self.pending, []
        if not pend:
            return
        ws = self._ws(pend[0][1].device)
        if self.FLOATS - self.off < (20 << 20):
            self._reduce()
        n = len(pend)
        f8 = [(_wgrad_fp8(p[0], *p[2]) if len(p[2]) == 2 else None) for p in pend]      # fp8 copies that exist by now (kept alive via pend)
        pend = [((f[0] if f else p[0]), p[1], p[2] + ((f[1],) if f else ())) for p, f in zip(pend, f8)]
        descs = (FS2Gemm * n)(*[p[0] for p in pend])
        parts = (FS2WgradPart * n)()
        res = {}
        if _bounds.ENABLED:
            for d in descs:
                _bounds.check_gemm(d, "fs2_wgrad_grouped")

        def launch():
            res["used"] = lib().fs2_wgrad_grouped(descs, n, ws.data_ptr() + 4 * self.off, self.FLOATS - self.off, parts, _stream())
            return [res["used"] > 0 and parts[i].splits > 0 for i in range(n)]      # the products the launch took
        if self.on_group is not None:
            self.on_group(descs, launch)
        else:
            launch()
        used = res["used"]
        if used < 0:
            _check(int(used), "fs2_wgrad_grouped")
        taken = [used > 0 and parts[i].splits > 0 for i in range(n)]
        if used > 0:
            if _bounds.ENABLED:
                for i in range(n):
                    if taken[i]:
                        _bounds.check_part(parts[i], ws.data_ptr() + 4 * self.off, 4 * int(used), "fs2_wgrad_grouped part")
            self.off += int(used)
            self.parts.extend(parts[i] for i in range(n) if taken[i])
            self.keep.extend(p[1] for i, p in enumerate(pend) if taken[i])
        for i, (g, out, _) in enumerate(pend):  # not in the group: on its own (partial tiles, or fs2_gemm's own flush)
            if not taken[i] and not self._one(g, out):
                _gemm_call(g)

    def _reduce(self):
        if self.parts:
            arr = (FS2WgradPart * len(self.parts))(*self.parts)
            n = len(self.parts)
            if _bounds.ENABLED:
                ws = self._ws(self.device)
                for q in self.parts:
                    _bounds.check_part(q, ws.data_ptr(), 4 * self.FLOATS)
            self.parts, self.keep, self.off = [], [], 0
            _check(lib().fs2_wgrad_reduce(arr, n, _stream()), "fs2_wgrad_reduce")

    def flush(self):
        self._launch_pending()
        self._reduce()
        self.spans = []

    def reset(self):
        """forget queued products and unreduced partial tiles WITHOUT launching anything: what a step that died between queueing and
        flushing (an exception inside a backward) left behind must not be added into the next step's gradients"""
        self.pending, self.parts, self.keep, self.spans, self.off = [], [], [], [], 0


_WG = _WgradSlices()


def wgrad_flush():
    """add the partial tiles of every deferred weight-gradient product into its gradient (one launch); the models call it once per
    announced parameter range and at the end of every backward"""
    _WG.flush()


def wgrad_launch():
    """launch the queued weight-gradient products now (their operands may be freed afterwards) but leave the partial tiles in the
    workspace: the reduce into the gradients waits for wgrad_flush()"""
    _WG._launch_pending()


def wgrad_reset():
    """drop whatever a failed step left queued (called at the start of every step, before zero_grad)"""
    _WG.reset()


def _wgrad_call(g, out, defer, extra_bytes=0, keep=()):
    if not _WG.run(g, out, defer, extra_bytes, keep):
        _gemm_call(g)


def _pick_split(M_out, N_out, K_red, batch):
    """split-K factor of a weight-gradient GEMM: ~384 blocks in all (1-2 per CU; measured optimum 256-384 on the
    config-2 shapes, tools/gemm_bench.py wsplit), at least 6 K-stages of 64 per split, at most 64 splits"""
    tiles = ((M_out + 127) // 128) * ((N_out + 127) // 128) * batch
    ktiles = max(1, (K_red + 63) // 64)
    want = max(1, int(round(384 / max(tiles, 1))))
    return int(max(1, min(want, max(1, ktiles // 6), 64)))


def wgrad(dy, x, out, split=None, defer=False):
    """out[N,K] (fp32) += dy[M,N]^T @ x[M,K]  (k-major operands; long reductions: partial tiles + reduce, else split-K with fp32 atomics).
    defer: the sum into `out` may wait until wgrad_flush()"""
    M, N = dy.shape
    K = x.shape[1]
    assert out.dtype == torch.float32 and out.shape == (N, K)
    g = FS2Gemm()
    g.A, g.B, g.lda, g.ldb = _p(dy), _p(x), _ld(dy), _ld(x)
    g.a_kmajor = g.b_kmajor = 1
    g.M, g.N, g.K, g.dtype = N, K, M, _dt(x)
    g.batch1 = g.batch2 = 1
    g.split_k = split or _pick_split(N, K, M, 1)
    g.accumulate = 1
    _epilogue(g, out, None, False, None, None, None, 1.0)
    _wgrad_call(g, out, defer and split is None, keep=(dy, x))
    return out


def wgrad_batched(dy, x, outs, defer=False):
    """outs[j][N,K] (fp32) += dy[:, j*N:(j+1)*N]^T @ x   for the column blocks of one dy (M, len(outs)*N): one launch
    with the blocks as the batch when the outputs sit at a constant address stride, else one launch per block."""
    nb = len(outs)
    M, N = dy.shape[0], dy.shape[1] // nb
    K = x.shape[1]
    ptrs = [o.data_ptr() for o in outs]
    step = ptrs[1] - ptrs[0] if nb > 1 else 0
    regular = nb > 1 and step > 0 and step % 4 == 0 and all(ptrs[j + 1] - ptrs[j] == step for j in range(nb - 1)) \
        and all(o.dtype == torch.float32 and o.shape == (N, K) and o.is_contiguous() for o in outs)
    if not regular:
        for j, o in enumerate(outs):
            wgrad(dy[:, j * N:(j + 1) * N], x, o, defer=defer)
        return outs
    g = FS2Gemm()
    g.A, g.B, g.lda, g.ldb = _p(dy), _p(x), _ld(dy), _ld(x)
    g.a_kmajor = g.b_kmajor = 1
    g.M, g.N, g.K, g.dtype = N, K, M, _dt(x)
    g.batch1, g.batch2 = nb, 1
    g.sA1, g.sB1, g.sC1 = N, 0, step // 4
    g.split_k = _pick_split(N, K, M, nb)
    g.accumulate = 1
    _epilogue(g, outs[0], None, False, None, None, None, 1.0)
    _wgrad_call(g, outs[0], defer, step * (nb - 1), keep=(dy, x))
    return outs


def conv_wgrad(dy, x, taps, pad, out, defer=False):
    """out[N, taps*C] (fp32) += sum_{b,t} dy[b,t,n] * x[b, t+j-pad, c]   (kernel layout [n][j*C+c])."""
    B, t, N = dy.shape
    C = x.shape[2]
    assert out.dtype == torch.float32 and out.shape == (N, taps * C)
    dy2, x2 = dy.view(B * t, N), x.view(B * t, C)
    g = FS2Gemm()
    g.A, g.B, g.lda, g.ldb = _p(dy2), _p(x2), _ld(dy2), _ld(x2)
    g.a_kmajor = g.b_kmajor = 1
    g.M, g.N, g.K, g.dtype = N, C, B * t, _dt(x)
    g.batch1, g.batch2 = 1, taps
    g.sC2 = C
    g.conv, g.taps, g.pad, g.seq_len = 2, taps, pad, t
    g.split_k = _pick_split(N, C, B * t, taps)
    g.accumulate = 1
    _epilogue(g, out, None, False, None, None, None, 1.0)
    _wgrad_call(g, out, defer, keep=(dy, x))
    return out


def bmm(a, b, out, trans_a=False, trans_b=True, alpha=1.0):
    """Batched product over two leading batch dims of 4-D strided views (last dim contiguous):
       trans_b=True  (NT): out[..,M,N] = a[..,M,K] @ b[..,N,K]^T
       trans_b=False (NN): out[..,M,N] = a[..,M,Ka>=Kb] @ b[..,Kb,N]      (a's extra columns must be 0)
       trans_a=True  (TN): out[..,M,N] = a[..,K,Ma>=M]^T @ b[..,K,N]      (requires trans_b=False)"""
    assert a.dim() == 4 and b.dim() == 4 and out.dim() == 4
    g = FS2Gemm()
    g.A, g.B = _p(a), _p(b)
    g.lda, g.ldb = _ld(a), _ld(b)
    g.sA1, g.sA2, g.sB1, g.sB2, g.sC1, g.sC2 = a.stride(0), a.stride(1), b.stride(0), b.stride(1), out.stride(0), out.stride(1)
    g.batch1, g.batch2 = out.shape[0], out.shape[1]
    g.M, g.N = out.shape[2], out.shape[3]
    g.dtype = _dt(a)
    g.split_k = 1
    if trans_a:
        assert not trans_b
        g.a_kmajor = g.b_kmajor = 1
        g.K = a.shape[2]
        assert b.shape[2] == g.K
    elif trans_b:
        g.K = a.shape[3]
        assert b.shape[3] == g.K
    else:
        g.b_kmajor = 1
        g.K, g.Kb = a.shape[3], b.shape[2]
    _epilogue(g, out, None, False, None, None, None, alpha)
    _gemm_call(g)
    return out


# ------------------------------------------------------------------------------------------------ shadows / casts
def cast_permute(src, dst, mode):
    """src fp32 (O,I,k) or (O,I); dst 2-D view (rows, >= k*I | k*O) with contiguous last dim."""
    O, I = src.shape[0], src.shape[1]
    k = src.shape[2] if src.dim() == 3 else 1
    _check(lib().fs2_cast_permute(_p(_c(src)), _p(dst), O, I, k, _ld(dst), mode, _dt(dst), _stream()), "fs2_cast_permute")
    if _FP8_W:
        base = dst._base if dst._base is not None else dst      # (a view into a fused shadow invalidates the whole shadow)
        _FP8_W.pop(base.data_ptr(), None)
        _FP8_W.pop(dst.data_ptr(), None)
    return dst


def permute_add(scratch, grad, rezero=False):
    """grad (O,I,k) fp32 += scratch (O, k*I) laid out [o][j*I+i]; rezero: leave the scratch zero-filled"""
    O, I = grad.shape[0], grad.shape[1]
    k = grad.shape[2] if grad.dim() == 3 else 1
    _check(lib().fs2_permute_add(_p(_c(scratch)), _p(_c(grad)), O, I, k, int(rezero), _stream()), "fs2_permute_add")


def cast(src, dtype, out=None):
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    _check(lib().fs2_cast(_p(_c(src)), _dt(src), _p(_c(out)), _dt(out), src.numel(), _stream()), "fs2_cast")
    return out


def add_cast(a, b, dtype):
    """(a + b).to(dtype) for two fp32 tensors of one shape, one pass"""
    assert a.dtype == b.dtype == torch.float32 and a.shape == b.shape
    out = torch.empty(a.shape, dtype=dtype, device=a.device)
    _check(lib().fs2_add_cast(_p(_c(a)), _p(_c(b)), _p(out), _dt(out), a.numel(), _stream()), "fs2_add_cast")
    return out


def copy_batched(dsts, srcs):
    """dsts[i].copy_(srcs[i]) for up to 8 pairs of contiguous device tensors of equal dtype and size per launch"""
    for k in range(0, len(dsts), 8):
        d, s = dsts[k:k + 8], srcs[k:k + 8]
        n = len(d)
        for a, b in zip(d, s):
            assert a.dtype == b.dtype and a.numel() == b.numel() and a.is_contiguous() and b.is_contiguous() and a.device == b.device
        sp = (ctypes.c_void_p * n)(*[_p(t) for t in s])
        dp = (ctypes.c_void_p * n)(*[_p(t) for t in d])
        nb = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in d])
        _check(lib().fs2_copy_batched(ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p), ctypes.cast(nb, ctypes.c_void_p), n,
                                      _stream()), "fs2_copy_batched")


def zero(t):
    """t (contiguous, 16-byte aligned, a multiple of 16 bytes) = 0"""
    _check(lib().fs2_zero(_p(_c(t)), t.numel() * t.element_size(), _stream()), "fs2_zero")
    return t


def cast_permute_batched(table_dev, n, dtype):
    """table_dev: uint8 device tensor holding n FS2CastDesc records"""
    _check(lib().fs2_cast_permute_batched(_p(table_dev), n, _DT[dtype], _stream()), "fs2_cast_permute_batched")


def make_cast_table(entries, device):
    """entries: list of (src fp32 tensor, dst 2-D view or 1-D fp32 tensor, mode) -> device table for cast_permute_batched"""
    arr = (FS2CastDesc * len(entries))()
    for d, (src, dst, mode) in zip(arr, entries):
        d.src, d.dst, d.mode = src.data_ptr(), dst.data_ptr(), mode
        if mode == 2:
            d.O, d.I, d.k, d.dld = src.numel(), 1, 1, 0
        else:
            d.O, d.I = src.shape[0], src.shape[1]
            d.k = src.shape[2] if src.dim() == 3 else 1
            d.dld = _ld(dst)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
    return raw.to(device)


def onehot(idx, nb, dtype):
    """idx int32 (M,) -> (M, nb) one-hot matrix in `dtype`"""
    M = idx.numel()
    out = torch.empty((M, nb), dtype=dtype, device=idx.device)
    _check(lib().fs2_onehot(_p(_c(idx)), _p(out), _dt(out), M, nb, _stream()), "fs2_onehot")
    return out


def colsum(x, out):
    """out[N] (fp32) += sum over rows of x[M,N] (row stride may exceed N)."""
    M, N = x.shape
    _check(lib().fs2_colsum(_p(x), _dt(x), M, N, _ld(x), _p(out), _stream()), "fs2_colsum")
    return out


def colsum_blocks(x, outs):
    """outs[j][n] (fp32) += sum over rows of x[:, j*n_j + n] for the equal column blocks of x: ONE launch when the
    outputs sit at a constant address stride (parameter arena), else one launch per block."""
    nb = len(outs)
    M, N = x.shape
    d = N // nb
    ptrs = [o.data_ptr() for o in outs]
    step = ptrs[1] - ptrs[0] if nb > 1 else 0
    if nb > 1 and step >= 4 * d and step % 4 == 0 and all(ptrs[j + 1] - ptrs[j] == step for j in range(nb - 1)) \
            and all(o.dtype == torch.float32 and o.numel() == d and o.is_contiguous() for o in outs):
        _check(lib().fs2_colsum_segmented(_p(x), _dt(x), M, N, _ld(x), _p(outs[0]), d, step // 4, _stream()),
               "fs2_colsum_segmented")
    else:
        for j, o in enumerate(outs):
            colsum(x[:, j * d:(j + 1) * d], o)
    return outs


# ------------------------------------------------------------------------------------------------ embedding / PE
def embedding_fwd(ids, table, out_dtype):
    out = torch.empty(tuple(ids.shape) + (table.shape[1],), dtype=out_dtype, device=table.device)
    _check(lib().fs2_embedding_fwd(_p(_c(ids)), _p(_c(table)), _p(out), _dt(out), ids.numel(), table.shape[1], _stream()),
           "fs2_embedding_fwd")
    return out


def embedding_bwd(ids, dout, dtable, padding_idx=-1):
    _check(lib().fs2_embedding_bwd(_p(_c(ids)), _p(_c(dout)), _dt(dout), _p(_c(dtable)), ids.numel(), dtable.shape[1],
                                   padding_idx, _stream()), "fs2_embedding_bwd")


def pe_add_fwd(a, pe, alpha, p, rng, site):
    B, t, d = a.shape
    out = torch.empty((B, t, d), dtype=torch.float32, device=a.device)
    _check(lib().fs2_pe_add_fwd(_p(_c(a)), _dt(a), _p(pe), _p(alpha), _p(out), B, t, d, p, _rng_ptr(rng, p), site,
                                _stream()), "fs2_pe_add_fwd")
    return out


def pe_add_bwd(dout, pe, da_dtype, dalpha, p, rng, site, need_da=True, dcolsum=None):
    B, t, d = dout.shape
    da = torch.empty((B, t, d), dtype=da_dtype, device=dout.device) if need_da else None
    _check(lib().fs2_pe_add_bwd(_p(_c(dout)), _p(pe), _p(da), _DT[da_dtype], _p(dalpha), B, t, d, p,
                                _rng_ptr(rng, p), site, _p(dcolsum), _stream()), "fs2_pe_add_bwd")
    return da


# ------------------------------------------------------------------------------------------------ LayerNorm family
def pe_add_ln_fwd(a, pe, alpha, gamma, beta, out_dtype, p, rng, site, ids=None, eps=1e-5):
    """x = dropout(a + alpha pe[t]) (fp32), y = LayerNorm(x) (out_dtype) in one pass -> (x, y, mean, rstd); ids (B, t) int64: the rows
    of `a` (then the fp32 embedding table) are gathered by id"""
    if ids is not None:
        B, t = ids.shape
        d = a.shape[1]
        assert a.dtype == torch.float32 and ids.dtype == torch.int64
        ids = _c(ids)
    else:
        B, t, d = a.shape
    x = torch.empty((B, t, d), dtype=torch.float32, device=a.device)
    y = torch.empty((B, t, d), dtype=out_dtype, device=a.device)
    mean = torch.empty(B * t, dtype=torch.float32, device=a.device)
    rstd = torch.empty(B * t, dtype=torch.float32, device=a.device)
    _check(lib().fs2_pe_add_ln_fwd(_p(_c(a)), _dt(a), _p(ids), _p(pe), _p(alpha), _p(gamma), _p(beta), _p(x), _p(y), _dt(y), _p(mean), _p(rstd),
                                   B, t, d, eps, p, _rng_ptr(rng, p), site, _stream()), "fs2_pe_add_ln_fwd")
    return x, y, mean, rstd


def ln_pe_add_bwd(dy, x, gamma, mean, rstd, ds, pe, da_dtype, dgamma, dbeta, dalpha, p, rng, site, dcolsum=None):
    """backward of pe_add_ln_fwd: da = gradient of `a` (da_dtype); dgamma / dbeta / dalpha / dcolsum are added to; ds (fp32, may be None):
    the gradient that reaches x through the residual stream"""
    B, t, d = x.shape
    da = torch.empty((B, t, d), dtype=da_dtype, device=x.device)
    _check(lib().fs2_ln_pe_add_bwd(_p(_c(dy)), _dt(dy), _p(_c(x)), _p(gamma), _p(mean), _p(rstd), _p(None if ds is None else _c(ds)), _p(pe), _p(da),
                                   _dt(da), _p(dgamma), _p(dbeta), _p(dalpha), _p(dcolsum), B, t, d, p, _rng_ptr(rng, p), site, _stream()),
           "fs2_ln_pe_add_bwd")
    return da


def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-5, p=0.0, rng=None, site=0):
    d = x.shape[-1]
    M = x.numel() // d
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    _check(lib().fs2_layernorm_fwd(_p(_c(x)), _dt(x), _p(gamma), _p(beta), _p(y), _dt(y), _p(mean), _p(rstd), M, d, eps,
                                   p, _rng_ptr(rng, p), site, _stream()), "fs2_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, p=0.0, rng=None, site=0, relu_mask=False, dx=None,
                  dcolsum=None):
    """dx (dtype of x) = LNbwd(dropout'(dy)) (* (x > 0) when relu_mask); with dx given: dx += ..."""
    d = x.shape[-1]
    M = x.numel() // d
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    _check(lib().fs2_layernorm_bwd(_p(_c(dy)), _dt(dy), _p(_c(x)), _dt(x), _p(gamma), _p(mean), _p(rstd), _p(_c(dx)),
                                   _dt(dx), _p(dgamma), _p(dbeta), M, d, p, _rng_ptr(rng, p), site, int(relu_mask),
                                   int(acc), _p(dcolsum), _stream()), "fs2_layernorm_bwd")
    return dx


def add_ln_fwd(r, a, gamma, beta, eps=1e-5, p=0.0, rng=None, site=0):
    """s = r + dropout(a) (fp32);  y = LN(s) (dtype of a).  returns s, y, mean, rstd"""
    d = r.shape[-1]
    M = r.numel() // d
    s = torch.empty_like(r)
    y = torch.empty_like(a)
    mean = torch.empty(M, dtype=torch.float32, device=r.device)
    rstd = torch.empty(M, dtype=torch.float32, device=r.device)
    with _RowQ8(y):
        _check(lib().fs2_add_ln_fwd(_p(_c(r)), _p(_c(a)), _dt(a), _p(s), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, d,
                                    eps, p, _rng_ptr(rng, p), site, _stream()), "fs2_add_ln_fwd")
    return s, y, mean, rstd


def add_ln_bwd(ds_down, dy, s, gamma, mean, rstd, dgamma, dbeta, p=0.0, rng=None, site=0, dcolsum=None):
    """returns dr (fp32, = ds_down + LNbwd(dy)) and da (dtype of dy, = dropout'(dr))"""
    d = s.shape[-1]
    M = s.numel() // d
    dr = torch.empty_like(s)
    da = torch.empty_like(dy)
    with _RowQ8(da):
        _check(lib().fs2_add_ln_bwd(_p(ds_down), _p(_c(dy)), _dt(dy), _p(_c(s)), _p(gamma), _p(mean), _p(rstd), _p(dr),
                                    _p(da), _p(dgamma), _p(dbeta), M, d, p, _rng_ptr(rng, p), site, _p(dcolsum), _stream()),
               "fs2_add_ln_bwd")
    return dr, da


def ffn_ln_fwd(f2, h, gamma, beta, eps=1e-5, p=0.0, rng=None, site=0):
    d = h.shape[-1]
    M = h.numel() // d
    y = torch.empty_like(h)
    mean = torch.empty(M, dtype=torch.float32, device=h.device)
    rstd = torch.empty(M, dtype=torch.float32, device=h.device)
    _check(lib().fs2_ffn_ln_fwd(_p(_c(f2)), _p(_c(h)), _dt(h), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, d, eps,
                                p, _rng_ptr(rng, p), site, _stream()), "fs2_ffn_ln_fwd")
    return y, mean, rstd


def ffn_ln_bwd(dy, f2, h, gamma, mean, rstd, dgamma, dbeta, p=0.0, rng=None, site=0, dcolsum=None):
    d = h.shape[-1]
    M = h.numel() // d
    g = torch.empty_like(h)
    _check(lib().fs2_ffn_ln_bwd(_p(_c(dy)), _p(_c(f2)), _p(_c(h)), _dt(h), _p(gamma), _p(mean), _p(rstd), _p(g),
                                _p(dgamma), _p(dbeta), M, d, p, _rng_ptr(rng, p), site, _p(dcolsum), _stream()),
           "fs2_ffn_ln_bwd")
    return g


def ffn_tail_fwd(f2, h, r, gamma1, beta1, gamma2, beta2, eps=1e-5, p=0.0, rng=None, site1=0, site2=0):
    """ffn_ln_fwd followed by add_ln_fwd in one row pass: yff = LN1(dropout(f2 + h, site1)); s = r + dropout(yff, site2);
    y = LN2(s).  returns s (fp32), y, mean1, rstd1, mean2, rstd2"""
    d = h.shape[-1]
    M = h.numel() // d
    s = torch.empty_like(r)
    y = torch.empty_like(h)
    st = [torch.empty(M, dtype=torch.float32, device=h.device) for _ in range(4)]
    with _RowQ8(y):
        _check(lib().fs2_ffn_tail_fwd(_p(_c(f2)), _p(_c(h)), _dt(h), _p(_c(r)), _p(gamma1), _p(beta1), _p(gamma2), _p(beta2), _p(s), _p(y),
                                      _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), M, d, eps, p, _rng_ptr(rng, p), site1, site2, _stream()),
               "fs2_ffn_tail_fwd")
    return s, y, st[0], st[1], st[2], st[3]


def ffn_tail_bwd(ds_down, dy, s, gamma2, mean2, rstd2, f2, h, gamma1, mean1, rstd1, dgamma2, dbeta2, dgamma1, dbeta1, p=0.0, rng=None,
                 site1=0, site2=0, dcolsum=None):
    """add_ln_bwd followed by ffn_ln_bwd in one row pass: returns dr (fp32, gradient of the residual r) and g = d(f2) = d(h)"""
    d = s.shape[-1]
    M = s.numel() // d
    dr = torch.empty_like(s)
    g = torch.empty_like(h)
    with _RowQ8(g):
        _check(lib().fs2_ffn_tail_bwd(_p(ds_down), _p(_c(dy)), _dt(dy), _p(_c(s)), _p(gamma2), _p(mean2), _p(rstd2), _p(_c(f2)), _p(_c(h)),
                                      _p(gamma1), _p(mean1), _p(rstd1), _p(dr), _p(g), _p(dgamma2), _p(dbeta2), _p(dgamma1), _p(dbeta1),
                                      _p(dcolsum), M, d, p, _rng_ptr(rng, p), site1, site2, _stream()), "fs2_ffn_tail_bwd")
    return dr, g


# ------------------------------------------------------------------------------------------------ attention softmax
def softmax_fwd(s, p_drop, key_mask, t, p=0.0, rng=None, site=0):
    """s, p_drop: (B,H,t,tp) views (row stride tp, contiguous over (H,t,tp)); in place on s."""
    B, H, _, tp = s.shape
    assert s.stride(3) == 1 and s.stride(2) == tp and s.stride(1) == t * tp
    assert p_drop.stride() == s.stride()
    _check(lib().fs2_softmax_fwd(_p(s), _p(p_drop), _dt(s), _p(_c(key_mask)), B, H, t, tp, s.stride(0), p,
                                 _rng_ptr(rng, p), site, _stream()), "fs2_softmax_fwd")


def attn_probs_supported(t, dk, dtype):
    """whether fs2_attn_probs_fwd takes (t, dk): bf16, dk in {32,64,128}, 64 x tp score strip within 160 KiB of LDS"""
    if os.environ.get("FS2_FUSED_ATTN", "1") == "0":       # A/B switch for measurements (tools/gpu_ci.sh ab)
        return False
    return dtype == torch.bfloat16 and lib().fs2_attn_probs_lds_bytes(int(t), int(dk)) > 0


def attn_second_product_supported(dk):
    """whether attn_probs_fwd / attn_ds_bwd can also produce dropout(P) V / dS K from their LDS strip"""
    return dk == 128 and os.environ.get("FS2_FUSED_ATTN2", "1") != "0"


def attn_probs_fwd(q, k, key_mask, p_out, p_drop, t, alpha, p=0.0, rng=None, site=0, v=None, out=None):
    """p_out = softmax(mask_keys(alpha * q k^T)), p_drop = dropout_p(p_out) in one kernel.  q, k: (B,H,t,dk) views of
    the fused projection (dk contiguous, common strides); p_out, p_drop: (B,H,t,tp) views as softmax_fwd takes them.
    With v (a view like q) and out (B,H,t,dk view of a (B,t,H,dk) tensor): also out = p_drop @ v (dk == 128)."""
    B, H, _, dk = q.shape
    tp = p_out.shape[3]
    assert q.stride() == k.stride() and q.stride(3) == 1 and q.dtype == k.dtype == torch.bfloat16
    assert p_out.stride(3) == 1 and p_out.stride(2) == tp and p_out.stride(1) == t * tp and p_drop.stride() == p_out.stride()
    if out is not None:
        assert v.stride() == q.stride() and out.stride(3) == 1 and out.stride(1) == q.stride(1) and out.dtype == q.dtype
    _check(lib().fs2_attn_probs_fwd(_p(q), _p(k), q.stride(2), q.stride(0), q.stride(1), dk, _p(_c(key_mask)), _p(p_out),
                                    _p(p_drop), p_out.stride(0), B, H, t, tp, float(alpha), p, _rng_ptr(rng, p), site,
                                    _p(v) if out is not None else None, _p(out), out.stride(2) if out is not None else 0,
                                    out.stride(0) if out is not None else 0, _stream()), "fs2_attn_probs_fwd")


def attn_ds_bwd(d_out, v, p_saved, ds, t, p=0.0, rng=None, site=0, k=None, dq=None, alpha=1.0):
    """ds = softmax/dropout backward of dP = d_out v^T in one kernel (dP stays in LDS).  d_out, v: (B,H,t,dk) views
    (dk contiguous, same head stride); p_saved, ds: (B,H,t,tp) views as softmax_bwd takes them.
    With k (a view like v) and dq (B,H,t,dk view): also dq = alpha * ds @ k (dk == 128)."""
    B, H, _, dk = v.shape
    tp = ds.shape[3]
    assert d_out.stride(3) == 1 and v.stride(3) == 1 and d_out.stride(1) == v.stride(1) and d_out.dtype == v.dtype == torch.bfloat16
    assert ds.stride(3) == 1 and ds.stride(2) == tp and ds.stride(1) == t * tp and p_saved.stride()[1:] == ds.stride()[1:]
    if dq is not None:
        assert k.stride() == v.stride() and dq.stride(3) == 1 and dq.stride(1) == v.stride(1) and dq.dtype == v.dtype
    _check(lib().fs2_attn_ds_bwd(_p(d_out), d_out.stride(2), d_out.stride(0), _p(v), v.stride(2), v.stride(0), v.stride(1),
                                 dk, _p(p_saved), p_saved.stride(0), _p(ds), ds.stride(0), B, H, t, tp, p,
                                 _rng_ptr(rng, p), site, _p(k) if dq is not None else None, _p(dq),
                                 dq.stride(2) if dq is not None else 0, dq.stride(0) if dq is not None else 0, float(alpha),
                                 _stream()), "fs2_attn_ds_bwd")


FLASH_MAX_KEYS = 16384


def flash_attn_supported(t, dk, dtype):
    """whether the flash kernels take (keys t, head size dk): bf16, d_k in {64, 96, 128}, up to 16384 keys"""
    return dtype == torch.bfloat16 and dk in (64, 96, 128) and 0 < t <= FLASH_MAX_KEYS


def flash_attn_keep_words_rect(B, H, tq, tk):
    """int16 words of the keep-bit stash of one flash_attention_fwd call (one bit per probability): (B, H, ceil(tk/64), tq, 4)"""
    return int(lib().fs2_flash_attn_keep_words_rect(int(B), int(H), int(tq), int(tk)))


def _flash_desc(q, k, v, key_mask, out, stats, keep, alpha, p_batch, p, causal, key_info):
    B, H, tq, dk = q.shape
    tk = k.shape[2]
    assert k.shape == v.shape == (B, H, tk, dk) and k.stride() == v.stride() and q.stride(3) == k.stride(3) == out.stride(3) == 1
    assert q.stride(1) == k.stride(1) == out.stride(1), "one head stride for q, k, v and the output"
    assert all(x.dtype == torch.bfloat16 for x in (q, k, v, out)) and stats.dtype == torch.float32 and stats.is_contiguous()
    assert stats.numel() == B * H * tq * 2 and key_mask.shape == (B, tk) and (not causal or tq == tk)
    if p > 0:
        assert keep.dtype == torch.int16 and keep.is_contiguous() and keep.numel() >= flash_attn_keep_words_rect(B, H, tq, tk)
    d = FS2FlashAttn()
    d.q, d.k, d.v = _p(q), _p(k), _p(v)
    d.q_row_stride, d.q_batch_stride, d.kv_row_stride, d.kv_batch_stride = q.stride(2), q.stride(0), k.stride(2), k.stride(0)
    d.head_stride, d.dk = q.stride(1), dk
    d.key_mask, d.key_info = _p(_c(key_mask)), _p(key_info)
    d.o, d.o_row_stride, d.o_batch_stride = _p(out), out.stride(2), out.stride(0)
    d.stats, d.keep_bits = _p(stats), (_p(keep) if p > 0 else None)
    d.causal, d.p_batch_stride = int(bool(causal)), int(p_batch)
    d.B, d.H, d.tq, d.tk, d.tkp = B, H, tq, tk, (tk + 7) // 8 * 8
    d.alpha, d.p = float(alpha), float(p)
    return d


def _flash_desc_bwd(d, d_out, aux, dq, dk_, dv, dbias):
    H, dk = d.H, d.dk
    d.d_out, d.do_row_stride, d.do_batch_stride, d.aux = _p(d_out), d_out.stride(2), d_out.stride(0), _p(aux)
    d.dq, d.dk_out, d.dv_out = _p(dq), _p(dk_), _p(dv)
    d.dq_row_stride, d.dq_batch_stride, d.dkv_row_stride, d.dkv_batch_stride = dq.stride(2), dq.stride(0), dk_.stride(2), dk_.stride(0)
    if dbias is not None:
        assert all(x.dtype == torch.float32 and x.is_contiguous() and x.numel() == H * dk for x in dbias)
        d.dbias_q, d.dbias_k, d.dbias_v = (_p(x) for x in dbias)


def _keep_words(d):
    return flash_attn_keep_words_rect(d.B, d.H, d.tq, d.tk) if d.p > 0 else 0


def flash_attention_fwd(q, k, v, key_mask, out, stats, keep, alpha, p_batch, p=0.0, rng=None, site=0, causal=False, key_info=None,
                        pregenerated=False):
    """out = dropout_p(softmax(mask(alpha q k^T))) v without the probabilities in HBM, general form: q (B,H,tq,dk), k / v (B,H,tk,dk)
    strided views with one head stride, key_mask (B,tk), causal (tq == tk): key j > query i masked like a padded key; stats
    (B,H,tq,2) fp32; keep: flash_attn_keep_words_rect(B,H,tq,tk) int16 words (None when p == 0); p_batch: batch stride of the virtual
    (B,[layers],H,tq,tkp) probability tensor (the Philox counters of softmax_rect_fwd / attn_probs_fwd: the same masks)."""
    d = _flash_desc(q, k, v, key_mask, out, stats, keep, alpha, p_batch, p, causal, key_info)
    d.rng, d.site, d.pregenerated = _rng_ptr(rng, p), site, int(bool(pregenerated))
    if _bounds.ENABLED:
        _bounds.check_flash(d, "fs2_flash_attention_fwd", keep_words=_keep_words(d))
    _check(lib().fs2_flash_attention_fwd(ctypes.byref(d), _stream()), "fs2_flash_attention_fwd")


def flash_attention_probs(q, k, v, key_mask, out, stats, keep, probs, alpha, p=0.0, causal=False, key_info=None):
    """probs (B,H,tq,tkp) bf16 view (batch stride free, the rest contiguous) = the post-dropout attention map of a finished
    flash_attention_fwd call with the same q, k, key_mask, stats, keep (the reference's return value, Models/modules.py:19-21)"""
    B, H, tq, dk = q.shape
    tk = k.shape[2]
    tkp = (tk + 7) // 8 * 8
    assert probs.dtype == torch.bfloat16 and probs.shape == (B, H, tq, tkp) and probs.stride()[1:] == (tq * tkp, tkp, 1)
    d = _flash_desc(q, k, v, key_mask, out, stats, keep, alpha, 0, p, causal, key_info)
    d.pregenerated = 1
    if _bounds.ENABLED:
        _bounds.check_flash(d, "fs2_flash_attention_probs", keep_words=_keep_words(d), probs=_p(probs), probs_batch=probs.stride(0))
    _check(lib().fs2_flash_attention_probs(ctypes.byref(d), _p(probs), probs.stride(0), _stream()), "fs2_flash_attention_probs")
    return probs


def flash_attention_bwd(q, k, v, key_mask, out, d_out, stats, keep, aux, dq, dk_, dv, alpha, p=0.0, causal=False, dbias=None, key_info=None):
    """backward of flash_attention_fwd: dq (B,H,tq,dk), dk_ / dv (B,H,tk,dk) views; aux: (B,H,tq,4) fp32 workspace; dbias: optional
    (dbias_q, dbias_k, dbias_v) fp32 vectors of H*dk that receive += the column sums of dq / dk / dv"""
    B, H, tq, dk = q.shape
    d = _flash_desc(q, k, v, key_mask, out, stats, keep, alpha, 0, p, causal, key_info)
    assert d_out.stride(3) == 1 and d_out.stride(1) == q.stride(1) and dq.stride(3) == 1 and dq.stride(1) == q.stride(1)
    assert dk_.stride() == dv.stride() and dk_.stride(3) == 1 and dk_.stride(1) == q.stride(1) and dk_.shape == k.shape
    assert all(x.dtype == torch.bfloat16 for x in (d_out, dq, dk_, dv)) and aux.dtype == torch.float32 and aux.is_contiguous()
    assert aux.numel() == B * H * tq * 4
    _flash_desc_bwd(d, d_out, aux, dq, dk_, dv, dbias)
    if _bounds.ENABLED:
        _bounds.check_flash(d, "fs2_flash_attention_bwd", backward=True, keep_words=_keep_words(d))
    _check(lib().fs2_flash_attention_bwd(ctypes.byref(d), _stream()), "fs2_flash_attention_bwd")


def flash_attn_keep_words(B, H, t):
    """number of int16 words of the keep-bit stash of one flash_attn_fwd call (one bit per probability)"""
    return int(lib().fs2_flash_attn_keep_words(int(B), int(H), int(t)))


def flash_mask_info(key_mask):
    """(B, 3) int32: {number of leading unmasked keys, last unmasked key + 1} of every row of a (B, t) key mask and, in column 2,
    the rows ranked by length (longest first: the order in which the kernels start them); computed once per stack and handed to
    every flash_attn_fwd / _bwd call as key_info"""
    km = _c(key_mask)
    B, t = km.shape
    info = torch.empty((B, 3), dtype=torch.int32, device=km.device)
    _check(lib().fs2_flash_attn_mask_info(_p(km), B, t, _p(info), _stream()), "fs2_flash_attn_mask_info")
    return info


PAD_MASK_MAX_B, PAD_MASK_MAX_T = 1024, 16384
_PAD_TICKET = {}


def pad_mask_info(pos, pad=0):
    """create_masks of the FastSpeech2 task (reference train_fastspeech2.py:55-82) and flash_mask_info in ONE launch:
    pos (B, t) int64 -> (mask (B, t) bool = pos != pad, info (B, 3) int32 as flash_mask_info returns it)"""
    assert pos.dtype == torch.int64 and pos.dim() == 2
    B, t = pos.shape
    if pos.stride(1) != 1 or (B > 1 and pos.stride(0) < t):
        pos = pos.contiguous()      # (rows of a row-strided view -- pos_mel[:, :-1] of the autoregressive trainer -- are read in place)
    mask = torch.empty((B, t), dtype=torch.bool, device=pos.device)
    info = torch.empty((B, 3), dtype=torch.int32, device=pos.device)
    tk = _PAD_TICKET.get(pos.device)
    if tk is None:      # the ticket word of the launch's last-block-ranks-the-rows step: zero once, the kernel leaves it zero
        tk = _PAD_TICKET[pos.device] = torch.zeros(4, dtype=torch.int32, device=pos.device)
    _check(lib().fs2_pad_mask_info(_p(pos), int(pos.stride(0)) if B > 1 else t, int(pad), B, t, _p(mask), _p(info), _p(tk), _stream()), "fs2_pad_mask_info")
    return mask, info


def flash_keep_bits(keep, B, H, t, p_batch, p, rng, site):
    """draw the dropout keep-bits of one flash_attn_fwd call ahead of time (pass pregenerated=True to that call)"""
    assert keep.dtype == torch.int16 and keep.is_contiguous() and keep.numel() >= flash_attn_keep_words(B, H, t) and p > 0
    _check(lib().fs2_flash_attn_keep_bits(_p(keep), int(p_batch), B, H, t, (t + 7) // 8 * 8, p, _rng_ptr(rng, p), site, _stream()),
           "fs2_flash_attn_keep_bits")


def flash_attn_fwd(q, k, v, key_mask, out, stats, keep, t, alpha, p_batch, p=0.0, rng=None, site=0, pregenerated=False, key_info=None):
    """out = dropout_p(softmax(mask_keys(alpha q k^T))) v without the probabilities in HBM; stats (B,H,t,2) fp32 = row maximum
    and sum of exponentials; keep: int16 tensor of flash_attn_keep_words(B,H,t) words receiving the dropout keep-bits (None when
    p == 0).  q, k, v: (B,H,t,128) views of the fused projection; out: (B,H,t,128) view of a (B,t,H,128) tensor; p_batch: batch
    stride of the virtual (B,[layers],H,t,tp) probability tensor (Philox counters of attn_probs_fwd)."""
    B, H, _, dk = q.shape
    assert q.stride() == k.stride() == v.stride() and q.stride(3) == 1 and q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16
    assert out.stride(3) == 1 and out.stride(1) == q.stride(1) and stats.is_contiguous() and stats.dtype == torch.float32
    assert stats.numel() == B * H * t * 2
    if dk != 128:       # the round-2 entry point is the d_k = 128 case of the general one
        return flash_attention_fwd(q, k, v, key_mask, out, stats, keep, alpha, p_batch, p, rng, site, False, key_info, pregenerated)
    if p > 0:
        assert keep.dtype == torch.int16 and keep.is_contiguous() and keep.numel() >= flash_attn_keep_words(B, H, t)
    if _bounds.ENABLED:     # (the same kernels as the general entry point: validate the descriptor that one would build)
        d = _flash_desc(q, k, v, key_mask, out, stats, keep, alpha, p_batch, p, False, key_info)
        d.rng, d.pregenerated = _rng_ptr(rng, p), int(bool(pregenerated))
        _bounds.check_flash(d, "fs2_flash_attn_fwd", keep_words=_keep_words(d))
    _check(lib().fs2_flash_attn_fwd(_p(q), _p(k), _p(v), q.stride(2), q.stride(0), q.stride(1), _p(_c(key_mask)), _p(key_info), _p(out),
                                    out.stride(2), out.stride(0), _p(stats), _p(keep) if p > 0 else None, int(bool(pregenerated)), int(p_batch), B, H, t,
                                    (t + 7) // 8 * 8, float(alpha), p, _rng_ptr(rng, p), site, _stream()), "fs2_flash_attn_fwd")


def flash_attn_bwd(q, k, v, key_mask, out, d_out, stats, keep, aux, dq, dk_, dv, t, alpha, p=0.0, dbias=None, key_info=None):
    """backward of flash_attn_fwd: dq, dk_, dv (B,H,t,128) views with common strides; keep: the forward's keep-bits; aux:
    (B,H,t,4) fp32 workspace; dbias: optional (dbias_q, dbias_k, dbias_v) fp32 vectors of H*128 that receive += the column sums of
    dq / dk / dv (the projections' bias gradients)."""
    B, H, _, dk = q.shape
    assert q.stride() == k.stride() == v.stride() and q.stride(3) == 1
    if dk != 128:
        return flash_attention_bwd(q, k, v, key_mask, out, d_out, stats, keep, aux, dq, dk_, dv, alpha, p, False, dbias, key_info)
    assert out.stride(3) == 1 and d_out.stride(3) == 1 and out.stride(1) == d_out.stride(1) == q.stride(1)
    assert dq.stride() == dk_.stride() == dv.stride() and dq.stride(3) == 1 and dq.stride(1) == q.stride(1)
    assert all(x.dtype == torch.bfloat16 for x in (q, k, v, out, d_out, dq, dk_, dv))
    assert stats.is_contiguous() and aux.is_contiguous() and aux.dtype == torch.float32 and aux.numel() == B * H * t * 4
    if p > 0:
        assert keep.dtype == torch.int16 and keep.is_contiguous() and keep.numel() >= flash_attn_keep_words(B, H, t)
    if dbias is not None:
        assert all(x.dtype == torch.float32 and x.is_contiguous() and x.numel() == H * dk for x in dbias)
    if _bounds.ENABLED:
        d = _flash_desc(q, k, v, key_mask, out, stats, keep, alpha, 0, p, False, key_info)
        _flash_desc_bwd(d, d_out, aux, dq, dk_, dv, dbias)
        _bounds.check_flash(d, "fs2_flash_attn_bwd", backward=True, keep_words=_keep_words(d))
    _check(lib().fs2_flash_attn_bwd(_p(q), _p(k), _p(v), q.stride(2), q.stride(0), q.stride(1), _p(_c(key_mask)), _p(key_info), _p(out),
                                    out.stride(2), out.stride(0), _p(d_out), d_out.stride(2), d_out.stride(0), _p(stats),
                                    _p(keep) if p > 0 else None, _p(aux), _p(dq), _p(dk_), _p(dv), dq.stride(2), dq.stride(0),
                                    *((_p(x) for x in dbias) if dbias is not None else (None, None, None)), B, H, t,
                                    float(alpha), p, _stream()), "fs2_flash_attn_bwd")


def softmax_bwd(dp, p_saved, t, p=0.0, rng=None, site=0):
    B, H, _, tp = dp.shape
    assert dp.stride(3) == 1 and dp.stride(2) == tp and dp.stride(1) == t * tp
    assert p_saved.stride()[1:] == dp.stride()[1:] and p_saved.dtype == dp.dtype
    _check(lib().fs2_softmax_bwd(_p(dp), dp.stride(0), _p(p_saved), p_saved.stride(0), _dt(dp), B, H, t, tp, p,
                                 _rng_ptr(rng, p), site, _stream()), "fs2_softmax_bwd")


def softmax_rect_fwd(s, p_drop, key_mask, tk, causal=False, p=0.0, rng=None, site=0):
    """softmax_fwd for tq query rows per head against tk keys: s, p_drop (B,H,tq,tkp) views, key_mask (B,tk); `causal`
    additionally masks key j > query i (tq == tk).  In place on s; pad columns [tk,tkp) -> 0."""
    B, H, tq, tkp = s.shape
    assert s.stride(3) == 1 and s.stride(2) == tkp and s.stride(1) == tq * tkp and p_drop.stride() == s.stride()
    _check(lib().fs2_softmax_rect_fwd(_p(s), _p(p_drop), _dt(s), _p(_c(key_mask)), B, H, tq, tk, tkp, s.stride(0), int(causal), p,
                                      _rng_ptr(rng, p), site, _stream()), "fs2_softmax_rect_fwd")


def softmax_rect_bwd(dp, p_saved, tk, p=0.0, rng=None, site=0):
    B, H, tq, tkp = dp.shape
    assert dp.stride(3) == 1 and dp.stride(2) == tkp and dp.stride(1) == tq * tkp
    assert p_saved.stride()[1:] == dp.stride()[1:] and p_saved.dtype == dp.dtype
    _check(lib().fs2_softmax_rect_bwd(_p(dp), dp.stride(0), _p(p_saved), p_saved.stride(0), _dt(dp), B, H, tq, tk, tkp, p,
                                      _rng_ptr(rng, p), site, _stream()), "fs2_softmax_rect_bwd")


def dropout(x, p, rng, site, relu_gate=None, out=None):
    """nn.Dropout on a contiguous tensor (same Philox stream layout as the fused dropouts); with relu_gate (same shape) the
    result is also zeroed where relu_gate <= 0: the backward through Dropout(ReLU(.)) given the ReLU's output."""
    x = _c(x)
    if out is None:
        out = torch.empty_like(x)
    _check(lib().fs2_dropout(_p(x), _p(_c(relu_gate)) if relu_gate is not None else None, _p(out), _dt(x), x.numel(), p,
                             _rng_ptr(rng, p), site, _stream()), "fs2_dropout")
    return out


def bce_logits_fwd(x, y, pos_weight, loss):
    """loss += F.binary_cross_entropy_with_logits(x, y, reduction='mean', pos_weight=pos_weight)"""
    _check(lib().fs2_bce_logits_fwd(_p(_c(x)), _dt(x), _p(_c(y)), x.numel(), float(pos_weight), _p(loss), _stream()), "fs2_bce_logits_fwd")


def bce_logits_bwd(x, y, pos_weight, gscale, dx_dtype):
    _same_numel(x, y, "bce_logits_bwd")
    dx = torch.empty(x.shape, dtype=dx_dtype, device=x.device)
    _check(lib().fs2_bce_logits_bwd(_p(_c(x)), _dt(x), _p(_c(y)), x.numel(), float(pos_weight), _p(gscale), _p(dx), _dt(dx),
                                    _stream()), "fs2_bce_logits_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ variance adaptor
def length_regulate_fwd(x, dur, T):
    B, L, d = x.shape
    out = torch.empty((B, T, d), dtype=x.dtype, device=x.device)
    starts = torch.empty((B, L + 1), dtype=torch.int32, device=x.device)
    _check(lib().fs2_length_regulate_fwd(_p(_c(x)), _dt(x), _p(_c(dur)), _p(out), _p(starts), B, L, T, d, _stream()),
           "fs2_length_regulate_fwd")
    return out, starts


def length_regulate_bwd(dout, starts, L, dx=None):
    B, T, d = dout.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty((B, L, d), dtype=dout.dtype, device=dout.device)
    _check(lib().fs2_length_regulate_bwd(_p(_c(dout)), _dt(dout), _p(starts), _p(_c(dx)), B, L, T, d, int(acc), _stream()),
           "fs2_length_regulate_bwd")
    return dx


def bucket_embed_add_fwd(x, f0, energy, pbins, ebins, Ep, Ee):
    """x + Ep[bucketize(f0, pbins)] + Ee[bucketize(energy, ebins)]; f0 (with pbins, Ep) or energy (with ebins, Ee) may be None:
    that term is left out (hp.pitch_pred / hp.energy_pred False) and its row of idx holds -1"""
    d = x.shape[-1]
    M = x.numel() // d
    out = torch.empty_like(x)
    idx = torch.empty((2, M), dtype=torch.int32, device=x.device)
    nb = (pbins if f0 is not None else ebins).numel() if (f0 is not None or energy is not None) else 1
    assert f0 is None or energy is None or pbins.numel() == ebins.numel()
    cp = lambda t: _p(_c(t)) if t is not None else None
    _check(lib().fs2_bucket_embed_add_fwd(_p(_c(x)), _dt(x), cp(f0), cp(energy), _p(pbins) if f0 is not None else None,
                                          _p(ebins) if energy is not None else None, nb, cp(Ep) if f0 is not None else None,
                                          cp(Ee) if energy is not None else None, _p(out), _p(idx), M, d, _stream()),
           "fs2_bucket_embed_add_fwd")
    return out, idx


def bucket_embed_bwd(dout, idx, dEp, dEe):
    d = dout.shape[-1]
    M = dout.numel() // d
    _check(lib().fs2_bucket_embed_bwd(_p(_c(dout)), _dt(dout), _p(idx), _p(dEp), _p(dEe), M, d, _stream()),
           "fs2_bucket_embed_bwd")


def linear1_fwd(x, w, b, mask):
    d = x.shape[-1]
    M = x.numel() // d
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    _check(lib().fs2_linear1_fwd(_p(_c(x)), _dt(x), _p(_c(w)), _p(b), _p(_c(mask)), _p(out), M, d, _stream()),
           "fs2_linear1_fwd")
    return out


def linear1_bwd(dout, x, w, mask, dw, db):
    d = x.shape[-1]
    M = x.numel() // d
    dx = torch.empty_like(x)
    _check(lib().fs2_linear1_bwd(_p(_c(dout)), _p(_c(x)), _dt(x), _p(_c(w)), _p(_c(mask)), _p(dx), _p(dw), _p(db), M, d,
                                 _stream()), "fs2_linear1_bwd")
    return dx


def ln_linear1_fwd(x, gamma, beta, w, b, mask, eps=1e-5, p=0.0, rng=None, site=0):
    """layernorm_fwd followed by linear1_fwd in one row pass (the normalised rows are not stored); returns out, mean, rstd"""
    d = x.shape[-1]
    M = x.numel() // d
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    _check(lib().fs2_ln_linear1_fwd(_p(_c(x)), _dt(x), _p(gamma), _p(beta), _p(_c(w)), _p(b), _p(_c(mask)), _p(out), _p(mean), _p(rstd),
                                    M, d, eps, p, _rng_ptr(rng, p), site, _stream()), "fs2_ln_linear1_fwd")
    return out, mean, rstd


def ln_linear1_bwd(dout, x, gamma, beta, mean, rstd, w, mask, dgamma, dbeta, dw, db, p=0.0, rng=None, site=0, relu_mask=False,
                   dcolsum=None):
    """linear1_bwd followed by layernorm_bwd in one row pass; returns dx (dtype of x)"""
    d = x.shape[-1]
    M = x.numel() // d
    dx = torch.empty_like(x)
    _check(lib().fs2_ln_linear1_bwd(_p(_c(dout)), _p(_c(x)), _dt(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(_c(w)), _p(_c(mask)),
                                    _p(dx), _p(dgamma), _p(dbeta), _p(dw), _p(db), _p(dcolsum), M, d, p, _rng_ptr(rng, p), site,
                                    int(relu_mask), _stream()), "fs2_ln_linear1_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ BatchNorm + tanh
def colstats(x, sums):
    C = x.shape[-1]
    _check(lib().fs2_colstats(_p(_c(x)), _dt(x), x.numel() // C, C, _p(sums), _stream()), "fs2_colstats")


def bn_finalize(sums, count, eps, momentum, running_mean, running_var, num_batches_tracked, count_dev=None):
    """sums: [2C] (+ optional trailing slots); count_dev: 1-element device tensor overriding `count`"""
    C = running_mean.numel()
    mean = torch.empty(C, dtype=torch.float32, device=sums.device)
    rstd = torch.empty(C, dtype=torch.float32, device=sums.device)
    _check(lib().fs2_bn_finalize(_p(sums), float(count), _p(count_dev), eps, momentum, _p(mean), _p(rstd), _p(running_mean),
                                 _p(running_var), _p(num_batches_tracked), C, _stream()), "fs2_bn_finalize")
    return mean, rstd


def bn_tanh_fwd(x, mean, rstd, gamma, beta, p=0.0, rng=None, site=0):
    C = x.shape[-1]
    y = torch.empty_like(x)
    _check(lib().fs2_bn_tanh_fwd(_p(_c(x)), _dt(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), x.numel() // C, C, p,
                                 _rng_ptr(rng, p), site, _stream()), "fs2_bn_tanh_fwd")
    return y


def bn_stats_tanh_fwd(x, sums, count, eps, momentum, running_mean, running_var, num_batches_tracked, gamma, beta, p=0.0, rng=None,
                      site=0, count_dev=None):
    """bn_finalize + bn_tanh_fwd in one launch -> (y, mean, rstd)"""
    C = x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    _check(lib().fs2_bn_stats_tanh_fwd(_p(_c(x)), _dt(x), _p(sums), float(count), _p(count_dev), eps, momentum, _p(gamma), _p(beta), _p(y),
                                       _p(mean), _p(rstd), _p(running_mean), _p(running_var), _p(num_batches_tracked), x.numel() // C, C, p,
                                       _rng_ptr(rng, p), site, _stream()), "fs2_bn_stats_tanh_fwd")
    return y, mean, rstd


def bn_tanh_bwd_reduce(dy, x, mean, rstd, gamma, beta, red, p=0.0, rng=None, site=0):
    C = x.shape[-1]
    _check(lib().fs2_bn_tanh_bwd_reduce(_p(_c(dy)), _p(_c(x)), _dt(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(red),
                                        x.numel() // C, C, p, _rng_ptr(rng, p), site, _stream()),
           "fs2_bn_tanh_bwd_reduce")


def bn_tanh_bwd_apply(dy, x, mean, rstd, gamma, beta, red, count, dgamma, dbeta, p=0.0, rng=None, site=0,
                      count_dev=None, dcolsum=None):
    C = x.shape[-1]
    dx = torch.empty_like(x)
    _check(lib().fs2_bn_tanh_bwd_apply(_p(_c(dy)), _p(_c(x)), _dt(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(red),
                                       float(count), _p(count_dev), _p(dx), _p(dgamma), _p(dbeta), x.numel() // C, C, p,
                                       _rng_ptr(rng, p), site, _p(dcolsum), _stream()), "fs2_bn_tanh_bwd_apply")
    return dx


# ------------------------------------------------------------------------------------------------ losses / optimizer
def _same_numel(pred, target, who):
    """the kernels read `pred.numel()` elements of both operands: a shorter target would be read past its end"""
    if target.numel() != pred.numel():
        raise ValueError(f"{who}: prediction {tuple(pred.shape)} and target {tuple(target.shape)} differ in size")


def l1_fwd(pred, target, loss, log1p_int_target=False):
    """loss[0] (fp32, zeroed by the caller) += mean |pred - target|"""
    _same_numel(pred, target, "l1_fwd")
    _check(lib().fs2_l1_fwd(_p(_c(pred)), _dt(pred), _p(_c(target)), int(log1p_int_target), pred.numel(), _p(loss),
                            _stream()), "fs2_l1_fwd")


def l1_bwd(pred, target, gscale, dpred_dtype, log1p_int_target=False):
    _same_numel(pred, target, "l1_bwd")
    dpred = torch.empty(pred.shape, dtype=dpred_dtype, device=pred.device)
    _check(lib().fs2_l1_bwd(_p(_c(pred)), _dt(pred), _p(_c(target)), int(log1p_int_target), pred.numel(), _p(gscale),
                            _p(dpred), _dt(dpred), _stream()), "fs2_l1_bwd")
    return dpred


def _l1_items(preds, targets, modes, dpreds=None):
    arr = (FS2L1Item * len(preds))()
    keep = []
    for i, (it, pr, tg, md) in enumerate(zip(arr, preds, targets, modes)):
        pr, tg = _c(pr), _c(tg)
        keep += [pr, tg]
        assert tg.numel() == pr.numel() and tg.dtype == (torch.int64 if md else torch.float32)
        it.pred, it.target, it.n, it.pred_dtype, it.target_mode = pr.data_ptr(), tg.data_ptr(), pr.numel(), _dt(pr), int(bool(md))
        if dpreds is not None:
            it.dpred, it.dpred_dtype = dpreds[i].data_ptr(), _dt(dpreds[i])
    return arr, keep


_L1_WS = {}


def l1_multi_fwd(preds, targets, modes, losses):
    """losses[i] = mean |pred_i - target_i| (modes[i]: the target is log(int64 target + 1)) and losses[len(preds)] = their sum;
    results stored (losses needs no zero fill), no float atomics: partial sums + one finishing block"""
    assert losses.dtype == torch.float32 and losses.numel() >= len(preds) + 1 and losses.is_contiguous()
    ws = _L1_WS.get(losses.device)
    if ws is None:      # the partial sums between the two launches (one buffer per device: this process launches on one stream)
        ws = _L1_WS[losses.device] = torch.empty(int(lib().fs2_l1_multi_workspace_floats()), dtype=torch.float32, device=losses.device)
    arr, keep = _l1_items(preds, targets, modes)
    _check(lib().fs2_l1_multi_fwd(ctypes.cast(arr, ctypes.c_void_p), len(preds), _p(losses), _p(ws), _stream()), "fs2_l1_multi_fwd")
    return losses


def l1_multi_bwd(preds, targets, modes, gscale, dpred_dtypes):
    """d(sum of the terms)/d(pred_i) * gscale[0] for every term; ONE launch"""
    dpreds = [torch.empty(pr.shape, dtype=dt, device=pr.device) for pr, dt in zip(preds, dpred_dtypes)]
    arr, keep = _l1_items(preds, targets, modes, dpreds)
    _check(lib().fs2_l1_multi_bwd(ctypes.cast(arr, ctypes.c_void_p), len(preds), _p(gscale), _stream()), "fs2_l1_multi_bwd")
    return dpreds


def sqnorm(x, out):
    _check(lib().fs2_sqnorm(_p(_c(x)), x.numel(), _p(out), _stream()), "fs2_sqnorm")


def adam_step(p, g, m, v, hyper, gsq, beta1, beta2, eps, max_norm, perm=None):
    """perm: int64 tensor (n_segments, 5) = {start, end, O, I, k} of the ranges whose gradient is stored [o][j][i] (optim.ParamArena)"""
    if perm is not None and perm.numel() > 0:
        assert perm.dtype == torch.int64 and perm.is_contiguous() and perm.shape[1] == 5 and perm.device == p.device
        _check(lib().fs2_adam_step_perm(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _p(gsq), beta1, beta2, eps, max_norm,
                                        _p(perm), perm.shape[0], _stream()), "fs2_adam_step_perm")
        return
    _check(lib().fs2_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _p(gsq), beta1, beta2, eps, max_norm,
                               _stream()), "fs2_adam_step")

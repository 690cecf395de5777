"""PyTorch-ROCm custom operators (``torch.library``) over the C ABI -- the registration BASELINE.json's north star names
("Python host code calling hand-written HIP C-ABI kernels through PyTorch-ROCm custom ops").

The model path itself calls ``transformer_tts_amd.ops`` directly from hand-written ``torch.autograd.Function``s (forward and
backward are both kernel sequences; nothing is traced, so the dispatcher has nothing to add there -- INTEGRATION.md).  This module
registers the operator-level entry points a maintainer of the reference would call from ordinary PyTorch code, in the ``fs2``
namespace, each with a fake (meta) implementation so that they compose with ``torch.compile`` / shape propagation, and with
autograd formulas that are themselves kernel calls:

    torch.ops.fs2.linear(x, w, bias, relu)            nn.Linear (+ReLU)            Models/modules.py:32-41,68
    torch.ops.fs2.conv1d_cl(x, w, bias, pad, relu)    nn.Conv1d, channels-last     Models/modules.py:76-84
    torch.ops.fs2.flash_attention(q, k, v, key_mask, causal)   attention() without dropout   Models/modules.py:7-21

``import transformer_tts_amd.torch_ops`` registers them; GPU tensors only (the library has no CPU path).
"""
import math

import torch

from . import ops

_lib = torch.library.Library("fs2", "DEF")
_lib.define("linear(Tensor x, Tensor w, Tensor? bias, bool relu) -> Tensor")
_lib.define("conv1d_cl(Tensor x, Tensor w, Tensor? bias, int pad, bool relu) -> Tensor")
_lib.define("flash_attention(Tensor q, Tensor k, Tensor v, Tensor key_mask, bool causal) -> (Tensor, Tensor)")


def _w_fwd(w):
    """(O, I, k) Conv1d weight -> the kernel layout [o][j*I + i] in the activation dtype"""
    O, I, k = w.shape
    return w.permute(0, 2, 1).reshape(O, k * I).contiguous()


@torch.library.impl(_lib, "linear", "CUDA")
def _linear_cuda(x, w, bias, relu):
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    out = ops.linear(x2, w.contiguous(), bias.float().contiguous() if bias is not None else None, relu=relu)
    return out.view(*x.shape[:-1], w.shape[0])


@torch.library.impl(_lib, "linear", "Meta")
def _linear_meta(x, w, bias, relu):
    return x.new_empty(*x.shape[:-1], w.shape[0])


@torch.library.impl(_lib, "conv1d_cl", "CUDA")
def _conv_cuda(x, w, bias, pad, relu):
    k = w.shape[2]
    return ops.conv(x.contiguous(), _w_fwd(w).to(x.dtype), k, pad, bias=bias.float().contiguous() if bias is not None else None, relu=relu)


@torch.library.impl(_lib, "conv1d_cl", "Meta")
def _conv_meta(x, w, bias, pad, relu):
    return x.new_empty(x.shape[0], x.shape[1], w.shape[0])


def _bthd(x):
    """(B,H,t,dk) tensor re-laid as a (B,t,H,dk) buffer viewed (B,H,t,dk): head stride dk, the layout the model itself uses"""
    return x if (x.stride(3) == 1 and x.stride(1) == x.shape[3]) else x.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)


def _flash_layout(q, k, v):
    """the kernels want ONE head stride for q, k, v and the outputs: head-interleaved (B,t,H,dk) views pass as they are, contiguous
    (B,H,t,dk) tensors of equal length too; anything else is copied into the head-interleaved layout"""
    B, H, tq, dk = q.shape
    same = q.stride(3) == k.stride(3) == 1 and k.stride() == v.stride() and q.stride(1) == k.stride(1)
    if same and q.stride(1) in (dk, tq * dk) and (q.stride(1) == dk or (q.is_contiguous() and k.is_contiguous())):
        return q, k, v
    return _bthd(q), _bthd(k), _bthd(v)


def _like_heads(ref, t):
    """empty (B,H,t,dk) tensor with the head stride of `ref`"""
    B, H, _, dk = ref.shape
    if ref.stride(1) == dk:
        return torch.empty((B, t, H, dk), dtype=ref.dtype, device=ref.device).permute(0, 2, 1, 3)
    return torch.empty((B, H, t, dk), dtype=ref.dtype, device=ref.device)


@torch.library.impl(_lib, "flash_attention", "CUDA")
def _flash_cuda(q, k, v, key_mask, causal):
    B, H, tq, dk = q.shape
    q, k, v = _flash_layout(q, k, v)
    out = _like_heads(q, tq)
    stats = torch.empty((B, H, tq, 2), dtype=torch.float32, device=q.device)
    ops.flash_attention_fwd(q, k, v, key_mask, out, stats, None, 1.0 / math.sqrt(dk), 0, 0.0, None, 0, causal=causal)
    return out, stats


@torch.library.impl(_lib, "flash_attention", "Meta")
def _flash_meta(q, k, v, key_mask, causal):
    B, H, tq, dk = q.shape
    return q.new_empty(B, tq, H, dk).permute(0, 2, 1, 3), q.new_empty(B, H, tq, 2, dtype=torch.float32)


def _linear_backward(ctx, grad):
    x, w, out = ctx.saved_tensors
    g2 = grad.reshape(-1, grad.shape[-1]).contiguous()
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    gm = g2 if not ctx.relu else torch.where(out.reshape(-1, out.shape[-1]) > 0, g2, torch.zeros_like(g2))      # ReLU'
    dx = ops.linear(gm, w.t().contiguous())
    dw = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
    ops.wgrad(gm, x2, dw)
    db = gm.float().sum(0) if ctx.has_bias else None
    return dx.view_as(x), dw.to(w.dtype), db, None


def _linear_setup(ctx, inputs, output):
    x, w, bias, relu = inputs
    ctx.relu, ctx.has_bias = relu, bias is not None
    ctx.save_for_backward(x, w, output)


torch.library.register_autograd("fs2::linear", _linear_backward, setup_context=_linear_setup, lib=_lib)


def _flash_backward(ctx, grad_out, _grad_stats):
    q, k, v, key_mask, out, stats = ctx.saved_tensors
    B, H, tq, dk = q.shape
    q, k, v = _flash_layout(q, k, v)           # (the same layout decision as the forward: `out` was allocated for it)
    dq, dk_, dv = _like_heads(q, tq), _like_heads(q, k.shape[2]), _like_heads(q, k.shape[2])
    aux = torch.empty((B, H, tq, 4), dtype=torch.float32, device=q.device)
    g = grad_out if (grad_out.stride(3) == 1 and grad_out.stride() == out.stride()) else \
        torch.empty_strided(out.shape, out.stride(), dtype=out.dtype, device=out.device).copy_(grad_out)
    ops.flash_attention_bwd(q, k, v, key_mask, out, g, stats, None, aux, dq, dk_, dv, 1.0 / math.sqrt(dk), 0.0, causal=ctx.causal)
    return dq, dk_, dv, None, None


def _flash_setup(ctx, inputs, output):
    q, k, v, key_mask, causal = inputs
    ctx.causal = causal
    ctx.save_for_backward(q, k, v, key_mask, output[0], output[1])


torch.library.register_autograd("fs2::flash_attention", _flash_backward, setup_context=_flash_setup, lib=_lib)


def _conv_backward(ctx, grad):
    """Conv1d backward as kernel calls: data gradient = the same implicit GEMM on the flipped, transposed weight
    (dst[i][j*O + o] = w[o][i][k-1-j], pad k-1-pad), weight gradient = the k-major product of ops.conv_wgrad in the kernel layout
    [o][j*I + i], viewed back as (O, I, k)"""
    x, w, out = ctx.saved_tensors
    O, I, k = w.shape
    g = grad.contiguous()
    if ctx.relu:
        g = torch.where(out > 0, g, torch.zeros_like(g))
    w_d = w.flip(2).permute(1, 2, 0).reshape(I, k * O).contiguous().to(x.dtype)
    dx = ops.conv(g, w_d, k, k - 1 - ctx.pad)
    dw = torch.zeros((O, k * I), dtype=torch.float32, device=w.device)
    ops.conv_wgrad(g, x.contiguous(), k, ctx.pad, dw)
    db = g.float().sum((0, 1)) if ctx.has_bias else None
    return dx, dw.view(O, k, I).permute(0, 2, 1).to(w.dtype), db, None, None


def _conv_setup(ctx, inputs, output):
    x, w, bias, pad, relu = inputs
    ctx.pad, ctx.relu, ctx.has_bias = pad, relu, bias is not None
    ctx.save_for_backward(x, w, output)


torch.library.register_autograd("fs2::conv1d_cl", _conv_backward, setup_context=_conv_setup, lib=_lib)

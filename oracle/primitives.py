"""CPU restatement of every tensor-level op of transformer_tts_amd/ops.py (same names, same
signatures), in plain PyTorch.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Two uses, both in tests/: (1) the per-kernel parity reference for the HIP kernels
(tests/test_kernels_gpu.py runs ops.X on the GPU and primitives.X on CPU copies); (2) a fake
backend that tests monkeypatch over ``transformer_tts_amd.ops`` so the host-side forward/backward
composition in transformer_tts_amd/Models can be checked on a machine without a GPU.

Each op cites the reference arithmetic it stands for (file:line of syoamakase/Transformer_TTS).
Dropout masks reproduce the kernels' Philox4x32-7 stream bit for bit (``drop_scale``), so
dropout-on comparisons are exact rather than statistical.
"""
import numpy as np
import torch
import torch.nn.functional as F

F32, BF16 = 0, 1
_COMPUTE = torch.float64   # internal precision of the restatement; results are cast to the output dtype


def lib():
    raise RuntimeError("oracle.primitives has no native library")


class Rng:
    def __init__(self, seed, device="cpu"):
        self.state = torch.tensor([seed, 0], dtype=torch.int64)

    def advance(self):
        self.state[1] += 1


# ------------------------------------------------------------------------------------------------ Philox dropout
PHILOX_ROUNDS = 7          # fs2_common.h PHILOX_ROUNDS


def _philox4x32(c0, c1, c2, c3, k0, k1):
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(PHILOX_ROUNDS):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
    return c0, c1, c2, c3


def drop_scale(shape, p, rng, site, base_index=None):
    """Per-element factors (0 or 65536/(65536 - thr16)) of the kernels' dropout stream (fs2_common.h drop_scale8/4) for
    a tensor whose element index (row-major, or ``base_index`` if given as an int64 array of that shape) selects
    the Philox call (index >> 3) and the 16-bit half of its output (word (index & 7) >> 1, half index & 1)."""
    if p <= 0.0:
        return torch.ones(shape, dtype=_COMPUTE)
    seed, off = int(rng.state[0]) & (2 ** 64 - 1), int(rng.state[1]) & (2 ** 64 - 1)
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64) if base_index is None else np.asarray(base_index, np.uint64).reshape(-1)
    q = idx >> np.uint64(3)
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) ^ (off >> 32)) & 0xFFFFFFFF
    r = _philox4x32(q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.full(n, site, np.uint64),
                    np.full(n, off & 0xFFFFFFFF, np.uint64), k0, k1)
    word = ((idx & np.uint64(7)) >> np.uint64(1)).astype(np.int64)
    half = (idx & np.uint64(1)).astype(np.uint64)
    bits = np.stack(r, axis=1)[np.arange(n), word].astype(np.uint64)
    bits16 = (bits >> (np.uint64(16) * half)) & np.uint64(0xFFFF)
    thr16 = int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))
    keep = bits16 >= np.uint64(thr16)
    scale = np.float32(65536.0) / (np.float32(65536.0) - np.float32(thr16))
    return torch.from_numpy(np.where(keep, scale, np.float32(0)).astype(np.float64).reshape(shape))


def _f(t):
    return t.to(_COMPUTE)


def _out(v, dtype):
    return v.to(dtype)


# ------------------------------------------------------------------------------------------------ GEMM family
def _epi(v, bias, relu, residual, relu_mask, colstats, out, out_dtype, alpha=1.0, colsum=None):
    v = v * alpha
    if bias is not None:
        v = v + _f(bias)
    if relu:
        v = torch.relu(v)
    if relu_mask is not None:
        v = torch.where(_f(relu_mask) > 0, v, torch.zeros_like(v))
    if residual is not None:
        v = v + _f(residual)
    dt = out.dtype if out is not None else out_dtype
    r = v.to(dt)
    if colstats is not None:
        n = r.shape[-1]
        rr = _f(r).reshape(-1, n)
        colstats[:n] += rr.sum(0).float()
        colstats[n:2 * n] += (rr * rr).sum(0).float()
    if colsum is not None:
        colsum += _f(r).reshape(-1, r.shape[-1]).sum(0).float()
    if out is not None:
        out.copy_(r)
        return out
    return r


def linear(x, w, bias=None, relu=False, residual=None, relu_mask=None, colstats=None, out=None, out_dtype=None, q8_out=False,
           alpha=1.0, colsum=None):
    """nn.Linear: Models/modules.py:32-41,49-51,68; Models/encoder.py:57; Models/postnets.py:43,67."""
    return _epi(_f(x) @ _f(w).t(), bias, relu, residual, relu_mask, colstats, out, out_dtype or x.dtype, alpha, colsum)


def _unfold(x, taps, pad):
    """(B,t,C) -> (B,t,taps*C) with column j*C+c = x[b, t+j-pad, c] (zero outside the sequence)."""
    B, t, C = x.shape
    cols = []
    for j in range(taps):
        s = j - pad
        sh = torch.zeros_like(x)
        lo, hi = max(0, -s), min(t, t - s)
        if hi > lo:
            sh[:, lo:hi] = x[:, lo + s:hi + s]
        cols.append(sh)
    return torch.cat(cols, dim=2)


def conv(x, w, taps, pad, bias=None, relu=False, residual=None, relu_mask=None, colstats=None, out=None, q8_out=False,
         out_dtype=None, colsum=None):
    """nn.Conv1d over time (Models/modules.py:76-84; varianceadaptor.py:203-209; postnets.py:28-39,71-75)
    in channels-last form; w is the kernel-layout shadow [n][j*C + c] = weight[n][c][j]."""
    return _epi(_unfold(_f(x), taps, pad) @ _f(w).t(), bias, relu, residual, relu_mask, colstats, out,
                out_dtype or x.dtype, 1.0, colsum)


def wgrad(dy, x, out, split=None, defer=False):
    """(defer: the product may leave partial tiles for ops.wgrad_flush(); the restatement adds at once)"""
    out += (_f(dy).t() @ _f(x)).float()
    return out


def wgrad_batched(dy, x, outs, defer=False):
    N = dy.shape[1] // len(outs)
    for j, o in enumerate(outs):
        wgrad(dy[:, j * N:(j + 1) * N], x, o)
    return outs


def view2d(x, M, d):
    """ops.view2d (a view that keeps a producer's fp8 copy attached): a plain view here"""
    return x.view(M, d)


def zero(t):
    return t.zero_()


def wgrad_launch():
    pass


def wgrad_flush():
    """the sum of deferred weight-gradient partial tiles (ops.wgrad_flush): nothing is deferred here"""


def conv_wgrad(dy, x, taps, pad, out, defer=False):
    B, t, N = dy.shape
    out += (_f(dy).reshape(B * t, N).t() @ _unfold(_f(x), taps, pad).reshape(B * t, -1)).float()
    return out


def bmm(a, b, out, trans_a=False, trans_b=True, alpha=1.0):
    """torch.matmul of attention() (Models/modules.py:8,20) and its backward products."""
    M, N = out.shape[2], out.shape[3]
    if trans_a:
        v = _f(a)[..., :M].transpose(-1, -2) @ _f(b)
    elif trans_b:
        v = _f(a) @ _f(b).transpose(-1, -2)
    else:
        v = _f(a)[..., : b.shape[2]] @ _f(b)
    out.copy_((alpha * v[..., :M, :N]).to(out.dtype))
    return out


# ------------------------------------------------------------------------------------------------ shadows / casts
def cast_permute(src, dst, mode):
    s = src if src.dim() == 3 else src.unsqueeze(-1)
    O, I, k = s.shape
    if mode == 0:
        v = s.permute(0, 2, 1).reshape(O, k * I)
    else:
        v = s.flip(2).permute(1, 2, 0).reshape(I, k * O)
    dst[:, : v.shape[1]] = v.to(dst.dtype)
    return dst


def permute_add(scratch, grad, rezero=False):
    g = grad if grad.dim() == 3 else grad.unsqueeze(-1)
    O, I, k = g.shape
    g += scratch.reshape(O, k, I).permute(0, 2, 1)
    if rezero:
        scratch.zero_()


def cast(src, dtype, out=None):
    if out is None:
        return src.to(dtype)
    out.copy_(src.to(out.dtype))
    return out


def add_cast(a, b, dtype):
    """the two gradient terms of the post-net's mel_pred (Models/postnets.py:67,74-75) as one tensor"""
    return (_f(a) + _f(b)).to(dtype)


def make_cast_table(entries, device):
    return list(entries)


def cast_permute_batched(table, n, dtype):
    for src, dst, mode in table:
        if mode == 2:
            dst.copy_(src.reshape(dst.shape))
        else:
            cast_permute(src, dst, mode)


def onehot(idx, nb, dtype):
    return torch.nn.functional.one_hot(idx.long(), nb).to(dtype)


def colsum(x, out):
    out += _f(x).sum(0).float()
    return out


def colsum_blocks(x, outs):
    d = x.shape[1] // len(outs)
    for j, o in enumerate(outs):
        colsum(x[:, j * d:(j + 1) * d], o)
    return outs


# ------------------------------------------------------------------------------------------------ embedding / PE
def embedding_fwd(ids, table, out_dtype):
    """nn.Embedding (Models/encoder.py:55,84)."""
    return table[ids].to(out_dtype)


def embedding_bwd(ids, dout, dtable, padding_idx=-1):
    flat, g = ids.reshape(-1), dout.reshape(-1, dout.shape[-1]).float()
    keep = flat != padding_idx
    dtable.index_add_(0, flat[keep], g[keep])


def pe_add_fwd(a, pe, alpha, p, rng, site):
    """PositionalEncoder.forward (Models/modules.py:107-111)."""
    B, t, d = a.shape
    v = (_f(a) + _f(alpha) * _f(pe[:t])) * drop_scale((B, t, d), p, rng, site)
    return v.float()


def pe_add_bwd(dout, pe, da_dtype, dalpha, p, rng, site, need_da=True, dcolsum=None):
    B, t, d = dout.shape
    g = _f(dout) * drop_scale((B, t, d), p, rng, site)
    dalpha += (g * _f(pe[:t])).sum().float()
    if dcolsum is not None:
        dcolsum += g.reshape(-1, d).sum(0).float()
    return g.to(da_dtype) if need_da else None


def pe_add_ln_fwd(a, pe, alpha, gamma, beta, out_dtype, p, rng, site, ids=None, eps=1e-5):
    """the head of an FFT stack: nn.Embedding (ids given; Models/encoder.py:55,84), PositionalEncoder (Models/modules.py:107-111) and
    norm_1 of the first layer (Models/layers.py:31), composed from the primitives of each"""
    a0 = embedding_fwd(ids, a, torch.float32) if ids is not None else a
    x = pe_add_fwd(a0, pe, alpha, p, rng, site)
    y, mean, rstd = layernorm_fwd(x, gamma, beta, out_dtype, eps)
    return x, y, mean, rstd


def ln_pe_add_bwd(dy, x, gamma, mean, rstd, ds, pe, da_dtype, dgamma, dbeta, dalpha, p, rng, site, dcolsum=None):
    dx = layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dx=None if ds is None else ds.clone())
    return pe_add_bwd(dx, pe, da_dtype, dalpha, p, rng, site, dcolsum=dcolsum)


# ------------------------------------------------------------------------------------------------ LayerNorm family
def _ln(x, gamma, beta, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    return (x - mu) * rstd * _f(gamma) + _f(beta), mu.squeeze(-1), rstd.squeeze(-1)


def _ln_bwd(dy, x, gamma, mean, rstd):
    xh = (x - _f(mean).unsqueeze(-1)) * _f(rstd).unsqueeze(-1)
    dg = dy * _f(gamma)
    c1 = dg.mean(-1, keepdim=True)
    c2 = (dg * xh).mean(-1, keepdim=True)
    dx = _f(rstd).unsqueeze(-1) * (dg - c1 - xh * c2)
    d = x.shape[-1]
    return dx, (dy * xh).reshape(-1, d).sum(0), dy.reshape(-1, d).sum(0)


def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-5, p=0.0, rng=None, site=0):
    """nn.LayerNorm (+ nn.Dropout after it: Models/varianceadaptor.py:219,222)."""
    y, mu, rstd = _ln(_f(x), gamma, beta, eps)
    y = y * drop_scale(tuple(x.shape), p, rng, site)
    return y.to(out_dtype), mu.reshape(-1).float(), rstd.reshape(-1).float()


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, p=0.0, rng=None, site=0, relu_mask=False, dx=None,
                  dcolsum=None):
    g = _f(dy) * drop_scale(tuple(x.shape), p, rng, site)
    v, dg, db = _ln_bwd(g, _f(x), gamma, mean.reshape(x.shape[:-1]), rstd.reshape(x.shape[:-1]))
    dgamma += dg.float()
    dbeta += db.float()
    if relu_mask:
        v = torch.where(_f(x) > 0, v, torch.zeros_like(v))
    if dcolsum is not None:
        dcolsum += v.reshape(-1, v.shape[-1]).sum(0).float()
    if dx is not None:
        dx.copy_((_f(dx) + v).to(dx.dtype))
        return dx
    return v.to(x.dtype)


def add_ln_fwd(r, a, gamma, beta, eps=1e-5, p=0.0, rng=None, site=0):
    """x = x + dropout(branch); norm(x)  (Models/layers.py:31-35,40 with the following norm)."""
    s = (_f(r) + _f(a) * drop_scale(tuple(a.shape), p, rng, site)).float()
    y, mu, rstd = _ln(_f(s), gamma, beta, eps)
    return s, y.to(a.dtype), mu.reshape(-1).float(), rstd.reshape(-1).float()


def add_ln_bwd(ds_down, dy, s, gamma, mean, rstd, dgamma, dbeta, p=0.0, rng=None, site=0, dcolsum=None):
    v, dg, db = _ln_bwd(_f(dy), _f(s), gamma, mean.reshape(s.shape[:-1]), rstd.reshape(s.shape[:-1]))
    dgamma += dg.float()
    dbeta += db.float()
    if ds_down is not None:
        v = v + _f(ds_down)
    dr = v.float()
    da64 = _f(dr) * drop_scale(tuple(s.shape), p, rng, site)
    if dcolsum is not None:
        dcolsum += da64.reshape(-1, s.shape[-1]).sum(0).float()
    return dr, da64.to(dy.dtype)


def ffn_ln_fwd(f2, h, gamma, beta, eps=1e-5, p=0.0, rng=None, site=0):
    """FeedForward tail: layer_norm(dropout(x + res)) (Models/modules.py:85-87)."""
    u = (_f(f2) + _f(h)) * drop_scale(tuple(h.shape), p, rng, site)
    y, mu, rstd = _ln(u, gamma, beta, eps)
    return y.to(h.dtype), mu.reshape(-1).float(), rstd.reshape(-1).float()


def ffn_ln_bwd(dy, f2, h, gamma, mean, rstd, dgamma, dbeta, p=0.0, rng=None, site=0, dcolsum=None):
    ds = drop_scale(tuple(h.shape), p, rng, site)
    u = (_f(f2) + _f(h)) * ds
    v, dg, db = _ln_bwd(_f(dy), u, gamma, mean.reshape(h.shape[:-1]), rstd.reshape(h.shape[:-1]))
    dgamma += dg.float()
    dbeta += db.float()
    if dcolsum is not None:
        dcolsum += (v * ds).reshape(-1, h.shape[-1]).sum(0).float()
    return (v * ds).to(h.dtype)


def ffn_tail_fwd(f2, h, r, gamma1, beta1, gamma2, beta2, eps=1e-5, p=0.0, rng=None, site1=0, site2=0):
    """Models/modules.py:85-87 then Models/layers.py:40,31: the two steps one after the other"""
    yff, m1, r1 = ffn_ln_fwd(f2, h, gamma1, beta1, eps, p, rng, site1)
    s, y, m2, r2 = add_ln_fwd(r, yff, gamma2, beta2, eps, p, rng, site2)
    return s, y, m1, r1, m2, r2


def ffn_tail_bwd(ds_down, dy, s, gamma2, mean2, rstd2, f2, h, gamma1, mean1, rstd1, dgamma2, dbeta2, dgamma1, dbeta1, p=0.0, rng=None,
                 site1=0, site2=0, dcolsum=None):
    dr, da = add_ln_bwd(ds_down, dy, s, gamma2, mean2, rstd2, dgamma2, dbeta2, p, rng, site2)
    g = ffn_ln_bwd(da, f2, h, gamma1, mean1, rstd1, dgamma1, dbeta1, p, rng, site1, dcolsum=dcolsum)
    return dr, g


# ------------------------------------------------------------------------------------------------ attention softmax
def _strided_index(t):
    """element offsets (relative to the view's first element) of a strided view, as int64 array"""
    idx = torch.zeros(t.shape, dtype=torch.int64)
    for dim, (n, s) in enumerate(zip(t.shape, t.stride())):
        shape = [1] * t.dim()
        shape[dim] = n
        idx = idx + (torch.arange(n, dtype=torch.int64) * s).view(shape)
    return idx.numpy()


def softmax_fwd(s, p_drop, key_mask, t, p=0.0, rng=None, site=0):
    """attention() of Models/modules.py:9-19: masked_fill(mask==0, -1e4) on keys, softmax, F.dropout
    (training=True always).  s already holds QK^T/sqrt(d_k); in place; pad columns [t,tp) -> 0."""
    B, H, _, tp = s.shape
    v = _f(s)[..., :t]
    v = v.masked_fill(key_mask.view(B, 1, 1, t) == 0, -1e4)
    pr = torch.softmax(v, dim=-1)
    full = torch.zeros((B, H, t, tp), dtype=_COMPUTE)
    full[..., :t] = pr
    s.copy_(full.to(s.dtype))
    if p_drop.data_ptr() != s.data_ptr() or p > 0:
        sc = drop_scale((B, H, t, tp), p, rng, site, base_index=_strided_index(s))
        p_drop.copy_((_f(s) * sc).to(s.dtype))


def attn_probs_supported(t, dk, dtype):
    """the (t, dk) envelope of fs2_attn_probs_fwd (attention.hip: strip_ld, fs2_attn_probs_lds_bytes)"""
    tp = (t + 7) // 8 * 8
    sld = tp + 8 if (tp // 8) % 2 == 0 else tp
    return dtype == torch.bfloat16 and dk in (32, 64, 128) and 0 < t <= 1024 and \
        64 * sld * 2 + 2 * 64 * dk * 2 + 1024 <= 160 * 1024


def attn_second_product_supported(dk):
    return dk == 128


def attn_probs_fwd(q, k, key_mask, p_out, p_drop, t, alpha, p=0.0, rng=None, site=0, v=None, out=None):
    """Models/modules.py:8-20 in one call: scores (stored in the tensors' dtype), key mask, softmax, dropout
    (and, with v/out, the product with the values)."""
    bmm(q, k, p_out[..., :t], trans_b=True, alpha=alpha)
    softmax_fwd(p_out, p_drop, key_mask, t, p, rng, site)
    if out is not None:
        bmm(p_drop, v, out, trans_b=False)


def attn_ds_bwd(d_out, v, p_saved, ds, t, p=0.0, rng=None, site=0, k=None, dq=None, alpha=1.0):
    """backward of attn_probs_fwd's softmax/dropout: dP = d_out v^T (stored in the tensors' dtype), then softmax_bwd
    (and, with k/dq, dQ = alpha dS K)."""
    bmm(d_out, v, ds[..., :t], trans_b=True)
    softmax_bwd(ds, p_saved, t, p, rng, site)
    if dq is not None:
        bmm(ds, k, dq, trans_b=False, alpha=alpha)


def softmax_bwd(dp, p_saved, t, p=0.0, rng=None, site=0):
    B, H, _, tp = dp.shape
    sc = drop_scale((B, H, t, tp), p, rng, site, base_index=_strided_index(p_saved))   # the forward's offsets
    g = torch.zeros((B, H, t, tp), dtype=_COMPUTE)
    g[..., :t] = (_f(dp) * sc)[..., :t]          # pad columns of dP are undefined on input
    pr = torch.zeros_like(g)
    pr[..., :t] = _f(p_saved)[..., :t]
    dot = (g * pr).sum(-1, keepdim=True)
    dp.copy_((pr * (g - dot)).to(dp.dtype))


def softmax_rect_fwd(s, p_drop, key_mask, tk, causal=False, p=0.0, rng=None, site=0):
    """attention() of Models/modules.py:9-19 for the autoregressive decoder: tq query rows against tk keys, key mask (B,tk),
    optionally the no-peak mask of train.py:26-36 (key j > query i masked); in place; pad columns [tk,tkp) -> 0."""
    B, H, tq, tkp = s.shape
    v = _f(s)[..., :tk]
    keep = (key_mask.view(B, 1, 1, tk) != 0)
    if causal:
        keep = keep & (torch.arange(tk).view(1, 1, 1, tk) <= torch.arange(tq).view(1, 1, tq, 1))
    v = v.masked_fill(~keep, -1e4)
    pr = torch.softmax(v, dim=-1)
    full = torch.zeros((B, H, tq, tkp), dtype=_COMPUTE)
    full[..., :tk] = pr
    s.copy_(full.to(s.dtype))
    if p_drop.data_ptr() != s.data_ptr() or p > 0:
        sc = drop_scale((B, H, tq, tkp), p, rng, site, base_index=_strided_index(s))
        p_drop.copy_((_f(s) * sc).to(s.dtype))


def softmax_rect_bwd(dp, p_saved, tk, p=0.0, rng=None, site=0):
    B, H, tq, tkp = dp.shape
    sc = drop_scale((B, H, tq, tkp), p, rng, site, base_index=_strided_index(p_saved))
    g = torch.zeros((B, H, tq, tkp), dtype=_COMPUTE)
    g[..., :tk] = (_f(dp) * sc)[..., :tk]
    pr = torch.zeros_like(g)
    pr[..., :tk] = _f(p_saved)[..., :tk]
    dot = (g * pr).sum(-1, keepdim=True)
    dp.copy_((pr * (g - dot)).to(dp.dtype))


def dropout(x, p, rng, site, relu_gate=None, out=None):
    """nn.Dropout (Models/prenets.py:32,35) with the kernels' Philox stream; relu_gate: also zero where relu_gate <= 0."""
    v = _f(x) * drop_scale(tuple(x.shape), p, rng, site)
    if relu_gate is not None:
        v = torch.where(_f(relu_gate) > 0, v, torch.zeros_like(v))
    r = v.to(x.dtype)
    if out is not None:
        out.copy_(r)
        return out
    return r


def bce_logits_fwd(x, y, pos_weight, loss):
    """F.binary_cross_entropy_with_logits(x, y, reduction='mean', pos_weight) (train.py:217)."""
    import torch.nn.functional as F
    loss += F.binary_cross_entropy_with_logits(_f(x), _f(y), reduction="mean",
                                               pos_weight=torch.tensor(float(pos_weight), dtype=_COMPUTE)).float()


def bce_logits_bwd(x, y, pos_weight, gscale, dx_dtype):
    xv, yv = _f(x), _f(y)
    d = (1 - yv) - (1 + (float(pos_weight) - 1) * yv) * torch.sigmoid(-xv)
    return (d * _f(gscale) / x.numel()).to(dx_dtype)


def quantize_fp8(x, bf8=False):
    """fs2_amax + fs2_quantize_fp8 (csrc/fp8.hip): per-tensor current scaling with a power-of-two scale 2^k, k the largest
    integer with amax * 2^k < 2^8 (e4m3) / 2^15 (e5m2); round-to-nearest-even OCP fp8 codes as uint8; state = [amax, 2^-k]."""
    xf = x.detach().float()
    amax = xf.abs().max()
    log2max, fmax, dt = (15, 57344.0, torch.float8_e5m2) if bf8 else (8, 448.0, torch.float8_e4m3fn)
    if float(amax) > 0:
        _, ex = torch.frexp(amax)                 # amax = m * 2^ex, m in [0.5, 1)  ->  exponent e = ex - 1
        k = log2max - 1 - (int(ex) - 1)
    else:
        k = 0
    k = max(-126, min(126, k))
    y = (xf * (2.0 ** k)).clamp(-fmax, fmax)
    q = y.to(dt).view(torch.uint8)
    return q, torch.tensor([float(amax), 2.0 ** (-k)], dtype=torch.float32)


def dequantize_fp8(q, state, bf8=False):
    dt = torch.float8_e5m2 if bf8 else torch.float8_e4m3fn
    return q.view(dt).to(_COMPUTE) * float(state[1])


# ------------------------------------------------------------------------------------------------ variance adaptor
def length_regulate_fwd(x, dur, T):
    """LengthRegulator.LR/expand + pad (Models/varianceadaptor.py:141-184,233-249)."""
    B, L, d = x.shape
    out = torch.zeros((B, T, d), dtype=x.dtype)
    starts = torch.zeros((B, L + 1), dtype=torch.int32)
    dd = dur.clamp(min=0)
    starts[:, 1:] = torch.cumsum(dd, dim=1).to(torch.int32)
    for b in range(B):
        rep = torch.repeat_interleave(x[b], dd[b], dim=0)[:T]
        out[b, : rep.shape[0]] = rep
    return out, starts


def length_regulate_bwd(dout, starts, L, dx=None):
    B, T, d = dout.shape
    v = torch.zeros((B, L, d), dtype=_COMPUTE)
    for b in range(B):
        for i in range(L):
            f0, f1 = int(starts[b, i]), min(int(starts[b, i + 1]), T)
            if f1 > f0:
                v[b, i] = _f(dout[b, f0:f1]).sum(0)
    if dx is not None:
        dx.copy_((_f(dx) + v).to(dx.dtype))
        return dx
    return v.to(dout.dtype)


def bucket_embed_add_fwd(x, f0, energy, pbins, ebins, Ep, Ee):
    """x + pitch_embedding(bucketize(f0)) + energy_embedding(bucketize(e)) (varianceadaptor.py:100,116,123-126)."""
    M = x.numel() // x.shape[-1]
    out = _f(x)
    none = torch.full((M,), -1, dtype=torch.int64, device=x.device)
    ip = ie = none
    if f0 is not None:              # hp.pitch_pred (:93,122-123)
        ip = torch.bucketize(f0, pbins)
        out = out + _f(Ep[ip])
    if energy is not None:          # hp.energy_pred (:112,124-125)
        ie = torch.bucketize(energy, ebins)
        out = out + _f(Ee[ie])
    return out.to(x.dtype), torch.stack([ip.reshape(-1), ie.reshape(-1)]).to(torch.int32)


def bucket_embed_bwd(dout, idx, dEp, dEe):
    g = dout.reshape(-1, dout.shape[-1]).float()
    if dEp is not None:
        dEp.index_add_(0, idx[0].long(), g)
    if dEe is not None:
        dEe.index_add_(0, idx[1].long(), g)


def linear1_fwd(x, w, b, mask):
    """linear_layer + squeeze + masked_fill(mask==0, 0) (Models/varianceadaptor.py:223-229)."""
    out = _f(x) @ _f(w).reshape(-1) + _f(b)
    return out.masked_fill(mask.reshape(out.shape) == 0, 0.0).float()


def linear1_bwd(dout, x, w, mask, dw, db):
    g = _f(dout) * (mask.reshape(dout.shape) != 0)
    dw += (g.unsqueeze(-1) * _f(x)).reshape(-1, x.shape[-1]).sum(0).float().reshape(dw.shape)
    db += g.sum().float()
    return (g.unsqueeze(-1) * _f(w).reshape(-1)).to(x.dtype)


def ln_linear1_fwd(x, gamma, beta, w, b, mask, eps=1e-5, p=0.0, rng=None, site=0):
    """the tail of a VariancePredictor (Models/varianceadaptor.py:226-231): LayerNorm, dropout, Linear(d -> 1), mask"""
    n, mean, rstd = layernorm_fwd(x, gamma, beta, x.dtype, eps, p, rng, site)
    return linear1_fwd(n, w, b, mask), mean, rstd


def ln_linear1_bwd(dout, x, gamma, beta, mean, rstd, w, mask, dgamma, dbeta, dw, db, p=0.0, rng=None, site=0, relu_mask=False,
                   dcolsum=None):
    shape = x.shape[:-1]
    n = (_f(x) - mean.reshape(shape).unsqueeze(-1)) * rstd.reshape(shape).unsqueeze(-1) * _f(gamma) + _f(beta)
    n = (n * drop_scale(tuple(x.shape), p, rng, site)).to(x.dtype)            # what the forward's Linear saw
    dn = linear1_bwd(dout, n, w, mask, dw, db)
    return layernorm_bwd(dn, x, gamma, mean, rstd, dgamma, dbeta, p, rng, site, relu_mask=relu_mask, dcolsum=dcolsum)


# ------------------------------------------------------------------------------------------------ BatchNorm + tanh
def colstats(x, sums):
    C = x.shape[-1]
    v = _f(x).reshape(-1, C)
    sums[:C] += v.sum(0).float()
    sums[C:2 * C] += (v * v).sum(0).float()


def bn_finalize(sums, count, eps, momentum, running_mean, running_var, num_batches_tracked, count_dev=None):
    """nn.BatchNorm1d in training mode (Models/postnets.py:58-59): biased batch variance for the
    normalisation, unbiased for running_var, momentum 0.1."""
    C = running_mean.numel()
    if count_dev is not None:
        count = float(count_dev[0])
    mu = _f(sums[:C]) / count
    var = (_f(sums[C:2 * C]) / count - mu * mu).clamp(min=0)
    if running_mean is not None:
        unb = var * count / (count - 1) if count > 1 else var
        running_mean.copy_(((1 - momentum) * _f(running_mean) + momentum * mu).float())
        running_var.copy_(((1 - momentum) * _f(running_var) + momentum * unb).float())
    if num_batches_tracked is not None:
        num_batches_tracked += 1
    return mu.float(), (1.0 / torch.sqrt(var + eps)).float()


def _bn_z(x, mean, rstd, gamma, beta):
    xh = (_f(x) - _f(mean)) * _f(rstd)
    return xh, torch.tanh(xh * _f(gamma) + _f(beta))


def bn_tanh_fwd(x, mean, rstd, gamma, beta, p=0.0, rng=None, site=0):
    """dropout(tanh(batch_norm(x))) (Models/postnets.py:71-73)."""
    _, th = _bn_z(x, mean, rstd, gamma, beta)
    return (th * drop_scale(tuple(x.shape), p, rng, site)).to(x.dtype)


def bn_stats_tanh_fwd(x, sums, count, eps, momentum, running_mean, running_var, num_batches_tracked, gamma, beta, p=0.0, rng=None,
                      site=0, count_dev=None):
    """the two steps above, one after the other (Models/postnets.py:58-59,71-73) -> (y, mean, rstd)"""
    mean, rstd = bn_finalize(sums, count, eps, momentum, running_mean, running_var, num_batches_tracked, count_dev)
    return bn_tanh_fwd(x, mean, rstd, gamma, beta, p, rng, site), mean, rstd


def bn_tanh_bwd_reduce(dy, x, mean, rstd, gamma, beta, red, p=0.0, rng=None, site=0):
    C = x.shape[-1]
    xh, th = _bn_z(x, mean, rstd, gamma, beta)
    dz = _f(dy) * drop_scale(tuple(x.shape), p, rng, site) * (1 - th * th)
    red[:C] += dz.reshape(-1, C).sum(0).float()
    red[C:2 * C] += (dz * xh).reshape(-1, C).sum(0).float()


def bn_tanh_bwd_apply(dy, x, mean, rstd, gamma, beta, red, count, dgamma, dbeta, p=0.0, rng=None, site=0,
                      count_dev=None, dcolsum=None):
    C = x.shape[-1]
    if count_dev is not None:
        count = float(count_dev[0])
    xh, th = _bn_z(x, mean, rstd, gamma, beta)
    dz = _f(dy) * drop_scale(tuple(x.shape), p, rng, site) * (1 - th * th)
    r0, r1 = _f(red[:C]) / count, _f(red[C:2 * C]) / count
    if dgamma is not None:
        dbeta += red[:C]
        dgamma += red[C:2 * C]
    dx = _f(gamma) * _f(rstd) * (dz - r0 - xh * r1)
    if dcolsum is not None:
        dcolsum += dx.reshape(-1, C).sum(0).float()
    return dx.to(x.dtype)


# ------------------------------------------------------------------------------------------------ losses / optimizer
def _l1_target(target, log1p_int_target):
    return torch.log(target.float() + 1) if log1p_int_target else target


def l1_fwd(pred, target, loss, log1p_int_target=False):
    """nn.L1Loss() over every element (train_fastspeech2.py:212-259)."""
    loss += (_f(pred) - _f(_l1_target(target, log1p_int_target))).abs().mean().float()


def l1_bwd(pred, target, gscale, dpred_dtype, log1p_int_target=False):
    d = _f(pred) - _f(_l1_target(target, log1p_int_target))
    return (torch.sign(d) * _f(gscale) / pred.numel()).to(dpred_dtype)


def l1_multi_fwd(preds, targets, modes, losses):
    """the same terms, one after the other, and their sum in the slot behind them (train_fastspeech2.py:212-259); stored, not added"""
    total = 0.0
    for i, (pr, tg, md) in enumerate(zip(preds, targets, modes)):
        term = (_f(pr) - _f(_l1_target(tg, md))).abs().mean().float()
        losses[i] = term
        total = total + term
    losses[len(preds)] = total
    return losses


def l1_multi_bwd(preds, targets, modes, gscale, dpred_dtypes):
    return [l1_bwd(pr, tg, gscale, dt, md) for pr, tg, md, dt in zip(preds, targets, modes, dpred_dtypes)]


def sqnorm(x, out):
    out += (_f(x) ** 2).sum().float()


def adam_step(p, g, m, v, hyper, gsq, beta1, beta2, eps, max_norm, perm=None):
    """clip_grad_norm_ + torch.optim.Adam.step (train_fastspeech2.py:312-315,416).  perm: rows {start, end, O, I, k} of the arena
    ranges whose gradient is stored [o][j][i] (the weight-gradient GEMM's layout) instead of the parameter's (O,I,k)."""
    if perm is not None and perm.numel() > 0:
        g = g.clone()
        for start, end, O, I, k in perm.tolist():
            g[start:end] = g[start:end].view(O, k, I).permute(0, 2, 1).reshape(-1)
    lr, bc1, bc2, gs = (float(h) for h in hyper)
    gmul = gs
    if max_norm > 0 and gsq is not None:
        gmul *= min(1.0, max_norm / (float(gsq[0]) ** 0.5 * gs + 1e-6))
    gg = g * gmul
    m.lerp_(gg, 1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p.addcdiv_(m, denom, value=-lr / bc1)

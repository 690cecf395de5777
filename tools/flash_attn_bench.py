"""Flash attention kernels against the LDS-strip path at the bench shapes (measurement only).
usage: python tools/flash_attn_bench.py [B H t]"""
import sys

import torch

sys.path.insert(0, ".")
from transformer_tts_amd import ops  # noqa: E402

dev = "cuda"
B, H, t = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (48, 2, 925)
dk = 128
tp = (t + 7) // 8 * 8
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, t, 3, H, dk, device=dev, generator=g).to(torch.bfloat16)
q, v, k = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
km = torch.ones(B, t, dtype=torch.bool, device=dev)
if len(sys.argv) >= 5 and sys.argv[4] == "ragged":       # prefix masks, lengths uniform in [0.45 t, t] (mean ~0.72 t as in the bench)
    lens = torch.linspace(0.45 * t, t, B).round().long()
    km = (torch.arange(t)[None, :] < lens[:, None]).to(dev)
P = torch.empty(B, H, t, tp, device=dev, dtype=torch.bfloat16)
Pd = torch.empty_like(P)
dS = torch.empty_like(P)
O = torch.empty(B, t, H, dk, device=dev, dtype=torch.bfloat16)
O4 = O.permute(0, 2, 1, 3)
dO = torch.randn(B, t, H, dk, device=dev, generator=g).to(torch.bfloat16)
dO4 = dO.permute(0, 2, 1, 3)
dqkv = torch.empty_like(qkv)
dq, dv, dk_ = (dqkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
stats = torch.empty(B, H, t, 2, device=dev)
aux = torch.empty(B, H, t, 4, device=dev)
keep = torch.empty(ops.flash_attn_keep_words(B, H, t), dtype=torch.int16, device=dev)
rng = ops.Rng(1, dev)
alpha = dk ** -0.5
import os
PD = float(os.environ.get('FLASH_P', '0.1'))
pb = H * t * tp


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def strip_fwd():
    ops.attn_probs_fwd(q, k, km, P, Pd, t, alpha, PD, rng, 3, v=v, out=O4)


def strip_bwd():
    ops.bmm(Pd, dO4, dv, trans_a=True, trans_b=False)
    ops.attn_ds_bwd(dO4, v, P, dS, t, PD, rng, 3, k=k, dq=dq, alpha=alpha)
    ops.bmm(dS, q, dk_, trans_a=True, trans_b=False, alpha=alpha)


def flash_fwd():
    ops.flash_attn_fwd(q, k, v, km, O4, stats, keep, t, alpha, pb, PD, rng, 3)


def flash_fwd_pregen():
    ops.flash_attn_fwd(q, k, v, km, O4, stats, keep, t, alpha, pb, PD, rng, 3, pregenerated=True)


def keep_bits():
    ops.flash_keep_bits(keep, B, H, t, pb, PD, rng, 3)


def flash_bwd():
    ops.flash_attn_bwd(q, k, v, km, O4, dO4, stats, keep, aux, dq, dk_, dv, t, alpha, PD)


dbias = [torch.zeros(H * dk, device=dev) for _ in range(3)]


def flash_bwd_bias():
    ops.flash_attn_bwd(q, k, v, km, O4, dO4, stats, keep, aux, dq, dk_, dv, t, alpha, PD, dbias=dbias)


flops = 2.0 * B * H * t * t * dk
for name, fn, units in (("strip fwd", strip_fwd, 2), ("strip bwd (+2 bmm)", strip_bwd, 4), ("flash fwd", flash_fwd, 2),
                        ("flash fwd, bits pre-drawn", flash_fwd_pregen, 2), ("keep-bits generator", keep_bits, 0),
                        ("flash bwd (dQ + dK/dV)", flash_bwd, 7), ("flash bwd + bias sums", flash_bwd_bias, 7)):
    if PD == 0 and name in ("flash fwd, bits pre-drawn", "keep-bits generator"):
        continue
    us = timeit(fn)
    print(f"{name:26s} {us:8.1f} us   {units * flops / us * 1e-6:7.1f} TFLOP/s (executed products)")

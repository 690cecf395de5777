"""Phase timing of the attention strip kernels (shader-clock stamps per workgroup; measurement only)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from transformer_tts_amd import ops  # noqa: E402

dev = "cuda"
B, H, t, dk = 48, 2, 925, 128
tp = (t + 7) // 8 * 8
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, t, 3, H, dk, device=dev, generator=g).to(torch.bfloat16)
q, v, k = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
km = torch.ones(B, t, dtype=torch.bool, device=dev)
P = torch.empty(B, H, t, tp, device=dev, dtype=torch.bfloat16)
Pd = torch.empty_like(P)
dS = torch.empty_like(P)
O = torch.empty(B, t, H, dk, device=dev, dtype=torch.bfloat16)
dO = torch.randn(B, t, H, dk, device=dev, generator=g).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
rng = ops.Rng(1, dev)
import os
QBT = int(os.environ.get('FS2_ATTN_QB', '64'))
nblk = ((t + QBT - 1) // QBT) * H * B
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
lib = ops.lib()
lib.fs2_debug_attn_timer.argtypes = [ctypes.c_void_p]
lib.fs2_debug_attn_timer.restype = None


def run(mode):
    if mode == 0:
        ops.attn_probs_fwd(q, k, km, P, Pd, t, dk ** -0.5, 0.1, rng, 3, v=v, out=O.permute(0, 2, 1, 3))
    else:
        ops.attn_ds_bwd(dO.permute(0, 2, 1, 3), v, P, dS, t, 0.1, rng, 3, k=k, dq=dqkv[:, :, 0].permute(0, 2, 1, 3), alpha=dk ** -0.5)


for mode in (0, 1):
    for _ in range(3):
        run(mode)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run(mode)
    e.record()
    torch.cuda.synchronize()
    print(f"mode {mode}: {s.elapsed_time(e) * 100:.1f} us per launch (warm)")
    lib.fs2_debug_attn_timer(buf.data_ptr())
    run(mode)
    torch.cuda.synchronize()
    lib.fs2_debug_attn_timer(None)
    st8 = buf.view(nblk, 8).double().cpu()
    st = st8[:, :4]
    d = st[:, 1:] - st[:, :-1]
    tot = st[:, 3] - st[:, 0]
    print(f"   phase 1 detail (wave 0): prologue issue {(st8[:, 4] - st8[:, 0]).mean():.0f}, first super-tile {(st8[:, 5] - st8[:, 4]).mean():.0f}, "
          f"remaining super-tiles {(st8[:, 6] - st8[:, 5]).mean():.0f}, barrier wait {(st8[:, 1] - st8[:, 6]).mean():.0f}")
    print(f"   per workgroup, shader-clock ticks (100 MHz = 10 ns each?): phase1 {d[:, 0].mean():.0f}  phase2 {d[:, 1].mean():.0f}  "
          f"phase3 {d[:, 2].mean():.0f}  total {tot.mean():.0f}; span of all blocks {st[:, 3].max() - st[:, 0].min():.0f}")

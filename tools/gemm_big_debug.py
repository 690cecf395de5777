"""Where does the large-tile GEMM differ from the 128-tile kernel?  Prefilled outputs (holes show as the fill value),
one-hot A (C[m][n] = B[n][m % K]) so that a wrong row / column mapping is readable from the error positions."""
import os
import sys

import torch

sys.path.insert(0, ".")
from transformer_tts_amd import ops  # noqa: E402

dev, T = "cuda", torch.bfloat16
torch.manual_seed(0)
for (M, N, K, bm) in ((256, 256, 64, "256"), (192, 256, 64, "192"), (1000, 512, 256, "256"), (44400, 256, 1024, "256")):
    k_idx = torch.arange(M, device=dev) % K
    A = torch.zeros(M, K, device=dev, dtype=T)
    A[torch.arange(M, device=dev), k_idx] = 1
    B = (torch.arange(N, device=dev).float()[:, None] * 1.0 + torch.arange(K, device=dev).float()[None, :] / 1024.0).to(T)
    ref = B.float().t()[k_idx]                     # (M, N): C[m][n] = B[n][m % K]
    for mode in ("0", "2"):
        os.environ["FS2_GEMM_BIG"], os.environ["FS2_GEMM_BIG_BM"] = mode, bm
        out = torch.full((M, N), -7.0, device=dev, dtype=T)
        ops.linear(A, B, out=out)
        torch.cuda.synchronize()
        o = out.float()
        bad = (o != ref)
        holes = (o == -7.0) & (ref != -7.0)
        print(f"M{M} N{N} K{K} bm{bm} mode{mode}: bad {int(bad.sum())} holes {int(holes.sum())}")
        if mode == "2" and int(bad.sum()):
            idx = bad.nonzero()[:12]
            for m, n in idx.tolist():
                print(f"    C[{m}][{n}] = {o[m, n].item():.4f} expected {ref[m, n].item():.4f}")
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("    bad rows:", rows[:40].tolist(), "... count", len(rows))
            print("    bad cols:", cols[:40].tolist(), "... count", len(cols))

"""Sliced split-K of the ring kernel (FS2Gemm.accumulate = 2 + fs2_splitk_reduce) on the encoder convolutions of config 2
(6144 x 256 x (9 x 1024)): result against the un-split product, time per split count / row-slab height."""
import os
import sys

import torch

sys.path.insert(0, ".")
from transformer_tts_amd import ops  # noqa: E402
from tools.gemm_big_bench import timeit  # noqa: E402

dev, T = "cuda", torch.bfloat16


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(T)
    x, w, b = r(48, 128, 1024), r(256, 9 * 1024) * 0.02, torch.randn(256, device=dev)
    res = torch.randn(48, 128, 256, device=dev)
    fl = 2.0 * 6144 * 256 * 9216
    fn = lambda: ops.conv(x, w, 9, 4, bias=b, relu=True, residual=res, out_dtype=torch.float32)
    os.environ["FS2_SPLITK_FWD"] = "0"
    ref = fn().float()
    t0w, t0c = timeit(fn, False), timeit(fn, True)
    print(f"un-split: {t0w:7.1f}/{t0c:7.1f} us ({fl / t0c / 1e6:6.0f} TF cold)")
    os.environ["FS2_SPLITK_FWD"] = "1"
    for cfg in sys.argv[1:] or ["5:128", "8:192", "10:256", "4:128", "6:192"]:
        n, bm = cfg.split(":")
        os.environ["FS2_SPLITK_N"], os.environ["FS2_GEMM_BIG_BM"] = n, bm
        out = fn().float()
        rel = float((out - ref).abs().max() / ref.abs().max())
        tw, tc = timeit(fn, False), timeit(fn, True)
        print(f"splits {n:>2s} bm {bm}: {tw:7.1f}/{tc:7.1f} us ({fl / tc / 1e6:6.0f} TF cold) maxrel {rel:.1e}", flush=True)


if __name__ == "__main__":
    main()

"""Row kernels of an FFT layer at the config-2 decoder size (44400 rows x 256 channels, bf16 mode, dropout 0.1):
microseconds per launch and algorithmic GB/s (every operand element once) against the 8 TB/s HBM peak."""
import sys

import torch

sys.path.insert(0, ".")
from transformer_tts_amd import ops  # noqa: E402
from tools.gemm_big_bench import timeit  # noqa: E402

dev, T = "cuda", torch.bfloat16


def main():
    M, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 44400), 256
    g = torch.Generator(device=dev).manual_seed(0)
    rb = lambda *s: torch.randn(*s, device=dev, generator=g).to(T)
    rf = lambda *s: torch.randn(*s, device=dev, generator=g)
    rng = ops.Rng(1, dev)
    f2, h, a, dy = rb(M, d), rb(M, d), rb(M, d), rb(M, d)
    r, ds_down = rf(M, d), rf(M, d)
    g1, b1, g2, b2 = rf(d), rf(d), rf(d), rf(d)
    z = lambda: torch.zeros(d, device=dev)
    s, y, m1, r1, m2, r2 = ops.ffn_tail_fwd(f2, h, r, g1, b1, g2, b2, p=0.1, rng=rng, site1=1, site2=2)
    s2, y2, ma, ra = ops.add_ln_fwd(r, a, g1, b1, p=0.1, rng=rng, site=3)
    dg = [z() for _ in range(5)]
    cases = {
        "ffn_tail_fwd": (lambda: ops.ffn_tail_fwd(f2, h, r, g1, b1, g2, b2, p=0.1, rng=rng, site1=1, site2=2), M * d * (2 + 2 + 4 + 4 + 2)),
        "ffn_tail_bwd": (lambda: ops.ffn_tail_bwd(ds_down, dy, s, g2, m2, r2, f2, h, g1, m1, r1, dg[0], dg[1], dg[2], dg[3], p=0.1, rng=rng,
                                                  site1=1, site2=2, dcolsum=dg[4]), M * d * (4 + 2 + 4 + 2 + 2 + 4 + 2)),
        "add_ln_fwd": (lambda: ops.add_ln_fwd(r, a, g1, b1, p=0.1, rng=rng, site=3), M * d * (4 + 2 + 4 + 2)),
        "add_ln_bwd": (lambda: ops.add_ln_bwd(ds_down, dy, s2, g1, ma, ra, dg[0], dg[1], p=0.1, rng=rng, site=3, dcolsum=dg[4]), M * d * (4 + 2 + 4 + 4 + 2)),
    }
    for name, (fn, by) in cases.items():
        tw, tc = timeit(fn, False, 20), timeit(fn, True, 10)
        print(f"{name:14s} {tw:7.1f} / {tc:7.1f} us warm / cold   {by / tw / 1e3:7.0f} / {by / tc / 1e3:7.0f} GB/s  ({by / tc / 1e3 / 8000:.2f} of 8 TB/s cold)", flush=True)


if __name__ == "__main__":
    main()

"""the 256 x 128 form of the weight-gradient kernel (FS2_KM_WIDE=1, default) against the 128 x 128 form (=0): stand-alone product kernels of
the configs[1] decoder shapes, warm / cold, and the result of both against each other"""
import os
import sys

import torch

sys.path.insert(0, ".")
from tools.gemm_big_bench import timeit  # noqa: E402
from transformer_tts_amd import ops  # noqa: E402


def run(M, N, K):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    outs = {}
    line = f"dW {N}x{K} from {M} rows:"
    for wide in ("0", "1"):
        os.environ["FS2_KM_WIDE"] = wide
        out = torch.zeros(N, K, device="cuda")
        ops.wgrad(dy, x, out)
        outs[wide] = out

        def f():
            ops.wgrad(dy, x, out, defer=True)
            ops._WG._launch_pending()
            ops._WG.parts, ops._WG.keep, ops._WG.spans, ops._WG.off = [], [], [], 0      # (product kernel alone: drop the reduce)
        tw, tc = timeit(f, False, 20), timeit(f, True, 10)
        fl = 2.0 * M * N * K
        line += f" | wide={wide} warm {tw:6.1f} us ({fl / tw / 1e6:5.0f} TF) cold {tc:6.1f} us ({fl / tc / 1e6:5.0f} TF)"
    ref = (dy.float().t() @ x.float())
    e0 = float((outs["0"] - ref).abs().max() / ref.abs().max())
    e1 = float((outs["1"] - ref).abs().max() / ref.abs().max())
    print(line + f" | max rel err vs fp32 matmul: {e0:.1e} / {e1:.1e}, wide vs narrow max |diff| {float((outs['0'] - outs['1']).abs().max()):.2e}", flush=True)


for (M, N, K) in ((44496, 1024, 256), (44496, 256, 1024), (44496, 256, 256), (44496, 768, 256), (6144, 256, 256), (1000, 384, 200)):
    run(M, N, K)
os.environ.pop("FS2_KM_WIDE", None)

"""weight gradients stand-alone, warm / cold time of the product kernel (partial tiles, no reduce): the configs[1] decoder FFN shape
(dW 1024 x 256 from 44.5 k rows) and the configs[4] shapes (d_model 512, 59.2 k rows), bf16 operands against their fp8 copies"""
import sys, os, torch
sys.path.insert(0, ".")
from tools.gemm_big_bench import timeit
from transformer_tts_amd import ops

def run(M, N, K, fp8):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.zeros(N, K, device="cuda")
    if fp8:
        for t, bf8 in ((dy, True), (x, False)):
            q, st = ops.quantize_fp8(t, bf8)
            t._fs2_q8 = (q, st, bf8)
    ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = fp8, fp8
    def f():
        ops.wgrad(dy, x, out, defer=True)
        ops._WG._launch_pending()
        ops._WG.parts, ops._WG.keep, ops._WG.spans, ops._WG.off = [], [], [], 0      # (product kernel alone: drop the reduce)
    try:
        tw, tc = timeit(f, False, 20), timeit(f, True, 10)
    finally:
        ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = False, False
    fl = 2.0 * M * N * K
    print(f"{'fp8 ' if fp8 else 'bf16'} dW {N}x{K} from {M} rows: warm {tw:6.1f} us ({fl/tw/1e6:5.0f} TF)  cold {tc:6.1f} us ({fl/tc/1e6:5.0f} TF)", flush=True)

for (M, N, K) in ((44496, 1024, 256), (44496, 256, 256), (59200, 2048, 512), (59200, 512, 2048), (59200, 1536, 512), (59200, 512, 512)):
    for fp8 in (False, True):
        run(M, N, K, fp8)

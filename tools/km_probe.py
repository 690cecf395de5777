"""decoder FFN weight gradient (1024 x 256 from 44.4 k rows) stand-alone: warm / cold time of the product kernel"""
import sys, os, torch
sys.path.insert(0, ".")
from tools.gemm_big_bench import timeit
from transformer_tts_amd import ops
M = 44496
for (N, K) in ((1024, 256), (256, 1024), (256, 256)):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.zeros(N, K, device="cuda")
    def f():
        ops.wgrad(dy, x, out, defer=True)
        ops._WG.parts, ops._WG.keep, ops._WG.spans, ops._WG.off = [], [], [], 0      # (product kernel alone: drop the reduce)
    tw, tc = timeit(f, False, 20), timeit(f, True, 10)
    fl = 2.0 * M * N * K
    print(f"FS2_KM_DBG={os.environ.get('FS2_KM_DBG','0')} dW {N}x{K}: warm {tw:6.1f} us ({fl/tw/1e6:5.0f} TF)  cold {tc:6.1f} us ({fl/tc/1e6:5.0f} TF)")

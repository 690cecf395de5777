"""ring depth of the 128-row tile of the ring kernel (FS2_RING_S, read once per process): python tools/ring_depth_probe.py 3|2"""
import os, sys
os.environ["FS2_RING_S"] = sys.argv[1] if len(sys.argv) > 1 else "3"
sys.path.insert(0, ".")
import torch
from tools.gemm_big_bench import timeit
from transformer_tts_amd import ops
dev, T = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(T)
M = 44400
x1024, w = r(M, 1024), r(256, 1024)
xp, wp = r(48, 925, 256), r(256, 5 * 256)
bias = torch.randn(256, device=dev)
os.environ["FS2_GEMM_RING"], os.environ["FS2_GEMM_WS"], os.environ["FS2_GEMM_BIG_BM"] = "2", "0", "128"
for name, fn in (("ffn2 44400x256x1024", lambda: ops.linear(x1024, w, bias)), ("post_conv k5", lambda: ops.conv(xp, wp, 5, 4, bias=bias))):
    line = name + " bm128:"
    S = os.environ["FS2_RING_S"]
    line += f" | S={S} warm {timeit(fn, False, 20):6.1f} cold {timeit(fn, True, 10):6.1f} us"
    print(line, flush=True)

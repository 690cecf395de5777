"""Timing-only probes of the weights-stationary GEMM (FS2_WS_DBG: stores dropped / activations not fetched), per row count."""
import os, sys, torch
sys.path.insert(0, ".")
from transformer_tts_amd import ops
from tools.gemm_big_bench import timeit
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
os.environ["FS2_GEMM_WS"] = "2"
for M in (11100, 44400, 88800, 177600):
    x = r(M, 256)
    for N in (768,):
        w = r(N, 256)
        fn = lambda: ops.linear(x, w)
        line = f"M={M} N={N}"
        for dbg, name in ((0, "full"), (3, "neither")):
            os.environ["FS2_WS_DBG"] = str(dbg)
            line += f" | {name} {timeit(fn, False, 20):7.1f} us"
        print(line, flush=True)
os.environ["FS2_WS_DBG"] = "0"

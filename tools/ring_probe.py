"""Where does the ring GEMM's time go?  Stand-alone products of the configs[1] step on the 16-wave ring kernel with one buffer's traffic
switched off at a time (FS2_RING_DBG: 1 = no C stores, 2 = no activation (A) loads, 4 = no weight (W) loads; the instruction stream is
unchanged, results are wrong), warm (back-to-back) and cold (1 GiB fill in between), per forced tile height.
    python tools/ring_probe.py [name filter ...]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from tools.gemm_big_bench import timeit  # noqa: E402
from transformer_tts_amd import ops  # noqa: E402

dev, T = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(T)
M = 44400
x1024, w_256_1024 = r(M, 1024), r(256, 1024)
x256, w_1024_256 = r(M, 256), r(1024, 256)
xp, wp = r(48, 925, 256), r(256, 5 * 256)
xe, we = r(48, 128, 256), r(1024, 9 * 256)
xe2, we2 = r(48, 128, 1024), r(256, 9 * 1024)
bias256, bias1024 = torch.randn(256, device=dev), torch.randn(1024, device=dev)
cases = {
    "ffn2 44400x256x1024": (lambda: ops.linear(x1024, w_256_1024, bias256), 2.0 * M * 256 * 1024),
    "ffn1 44400x1024x256": (lambda: ops.linear(x256, w_1024_256, bias1024, relu=True), 2.0 * M * 1024 * 256),
    "post_conv k5 44400x256x1280": (lambda: ops.conv(xp, wp, 5, 4, bias=bias256), 2.0 * M * 256 * 1280),
    "enc_conv1 k9 6144x1024x2304": (lambda: ops.conv(xe, we, 9, 4, bias=bias1024, relu=True), 2.0 * 6144 * 1024 * 2304),
    "enc_conv2 k9 6144x256x9216": (lambda: ops.conv(xe2, we2, 9, 4, bias=bias256), 2.0 * 6144 * 256 * 9216),
}
only = sys.argv[1:]
os.environ["FS2_GEMM_RING"], os.environ["FS2_GEMM_WS"] = "2", "0"
for name, (fn, fl) in cases.items():
    if only and not any(o in name for o in only):
        continue
    for bm in ("128", "192", "256"):
        os.environ["FS2_GEMM_BIG_BM"] = bm
        line = f"{name:30s} bm {bm}:"
        for dbg, label in ((0, "all"), (1, "-C"), (2, "-A"), (4, "-W"), (6, "-A-W"), (7, "none")):
            os.environ["FS2_RING_DBG"] = str(dbg)
            tw, tc = timeit(fn, False), timeit(fn, True)
            line += f" | {label} {tw:6.1f}/{tc:6.1f} us" + (f" ({fl / tw / 1e6:4.0f}/{fl / tc / 1e6:4.0f} TF)" if dbg == 0 else "")
        print(line, flush=True)
os.environ.pop("FS2_RING_DBG", None)

# ---- K sweep at M = 44400, N = 256 on the 192-row tile: slope = time per 64-deep slot, intercept = fixed time of a launch
if not only or "ksweep" in only:
    os.environ["FS2_GEMM_BIG_BM"] = "192"
    for K in (256, 512, 1024, 2048, 4096):
        xa, wa = r(M, K), r(256, K)
        fn = lambda: ops.linear(xa, wa, bias256)
        line = f"ksweep 44400x256x{K:<5d} bm 192:"
        for dbg, label in ((0, "all"), (7, "none")):
            os.environ["FS2_RING_DBG"] = str(dbg)
            tw, tc = timeit(fn, False, iters=20), timeit(fn, True)
            line += f" | {label} {tw:6.1f}/{tc:6.1f} us"
        print(line, flush=True)
    # an empty kernel of the same launch geometry would be the floor of the intercept: event pair around a tiny product
    xs, wsm = r(192 * 232, 64), r(256, 64)
    os.environ["FS2_RING_DBG"] = "7"
    print(f"one slot (K = 64), 232 tiles, no traffic: {timeit(lambda: ops.linear(xs, wsm, bias256), False, iters=20):6.1f} us", flush=True)
    os.environ.pop("FS2_RING_DBG", None)

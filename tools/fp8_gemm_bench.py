"""configs[4] products (d_model 512, FFN 2048, 64 x ~925 frames): bf16 ring kernel against the fp8 kernels with the operands quantised
ONCE outside the timed region (kernel against kernel), and against the full fp8 call (amax + quantise of both operands per call)."""
import os, sys, torch
sys.path.insert(0, ".")
from tools.gemm_big_bench import timeit
from transformer_tts_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
M = 64 * 925
x512, x2048, x1536 = r(M, 512), r(M, 2048), r(M, 1536)
w = {(n, k): (r(n, k) * k ** -0.5) for (n, k) in ((2048, 512), (512, 2048), (1536, 512), (512, 512), (512, 1536))}
b2048, b512, b1536 = (torch.randn(n, device=dev) for n in (2048, 512, 1536))
mask2048 = r(M, 2048)
res512 = torch.randn(M, 512, device=dev)
cs = torch.zeros(4096, device=dev)
xp, wp = r(64, 925, 512), r(512, 5 * 512) * (5 * 512) ** -0.5
cases = {
    "ffn1 bias+relu Mx2048x512": (lambda: ops.linear(x512, w[(2048, 512)], b2048, relu=True), 2.0 * M * 2048 * 512, False),
    "ffn2 bias+res f32 Mx512x2048": (lambda: ops.linear(x2048, w[(512, 2048)], b512, residual=res512, out_dtype=torch.float32), 2.0 * M * 512 * 2048, False),
    "qkv Mx1536x512": (lambda: ops.linear(x512, w[(1536, 512)], b1536), 2.0 * M * 1536 * 512, False),
    "proj Mx512x512": (lambda: ops.linear(x512, w[(512, 512)], b512), 2.0 * M * 512 * 512, False),
    "ffn1-dgrad mask+colsum Mx512x2048": (lambda: ops.linear(x2048, w[(512, 2048)], relu_mask=None, colsum=cs[:512].zero_()), 2.0 * M * 512 * 2048, True),
    "ffn2-dgrad mask Mx2048x512": (lambda: ops.linear(x512, w[(2048, 512)], relu_mask=mask2048), 2.0 * M * 2048 * 512, True),
    "qkv-dgrad Mx512x1536": (lambda: ops.linear(x1536, w[(512, 1536)]), 2.0 * M * 512 * 1536, True),
    "post_conv k5 Mx512x2560": (lambda: ops.conv(xp, wp, 5, 4, bias=b512), 2.0 * M * 512 * 2560, False),
}
orig = ops._fp8_operands
cache = {}


def cached(gd, x2, wt):
    key = (x2.data_ptr(), wt.data_ptr(), ops.FP8_MODE["backward"])
    if key not in cache:
        xq, sx = ops.quantize_fp8(x2, bf8=ops.FP8_MODE["backward"])
        wq, sw = ops.quantize_fp8(wt, bf8=False)
        cache[key] = (xq, sx.clone(), wq, sw.clone())
    xq, sx, wq, sw = cache[key]
    gd.A, gd.B, gd.lda, gd.ldb = xq.data_ptr(), wq.data_ptr(), xq.stride(-2), wq.stride(-2)
    gd.dtype = ops.BF8_FP8 if ops.FP8_MODE["backward"] else ops.FP8
    gd.scale_a, gd.scale_b = sx[1:].data_ptr(), sw[1:].data_ptr()
    return cache[key]


only = sys.argv[1:]
for name, (fn, fl, backward) in cases.items():
    if only and not any(o in name for o in only):
        continue
    ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = False, False
    ref = fn().float()
    tw, tc = timeit(fn, False), timeit(fn, True)
    tile = ops.lib().fs2_gemm_last_tile()
    line = f"{name:36s} bf16[{tile}] {tw:6.1f}/{tc:6.1f} us ({fl / tw / 1e6:5.0f} TF)"
    ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = True, backward
    try:
        for label, quant in (("f8-ring", cached), ("f8-ring+quantise", orig)):
            ops._fp8_operands = quant
            ops.fp8_begin_step(torch.device(dev))
            out = fn().float()
            rel = float((out - ref).norm() / ref.norm())
            def run():
                ops.fp8_begin_step(torch.device(dev)) if quant is orig else None
                fn()
            tw, tc = timeit(run, False), timeit(run, True)
            line += f" | {label}[{ops.lib().fs2_gemm_last_tile()}] {tw:6.1f}/{tc:6.1f} us ({fl / tw / 1e6:5.0f} TF) rel {rel:.3f}"
    finally:
        ops._fp8_operands = orig
        ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = False, False
    print(line, flush=True)

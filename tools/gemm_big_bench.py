"""The 16-wave GEMM kernels (gemm_ring.hip: tiled; gemm_ws.hip: weights-stationary stream for K = 256; gemm_big_km.hip: weight
gradients) against the 4-wave kernel (gemm.hip) on config-2 product shapes: results compared element-wise, both timed warm
(back-to-back launches) and cold (1 GiB fill between launches).  Arguments: tile heights (128 / 192 / 256) and name filters."""
import os
import sys

import torch

sys.path.insert(0, ".")
from transformer_tts_amd import ops  # noqa: E402

dev, T = "cuda", torch.bfloat16
_flush = None


def timeit(fn, cold, iters=10):
    global _flush
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if not cold:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / iters
    if _flush is None:
        _flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    tot = 0.0
    for i in range(iters):
        _flush.fill_(float(i))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot * 1e3 / iters


def main():
    cfgs = [a for a in sys.argv[1:] if a.isdigit() or ":" in a] or ["256"]
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(T)
    M = 44400
    x256, x1024, x768 = r(M, 256), r(M, 1024), r(M, 768)
    w_1024_256, w_256_1024, w_768_256, w_256_256 = r(1024, 256), r(256, 1024), r(768, 256), r(256, 256)
    bias1024, bias256 = torch.randn(1024, device=dev), torch.randn(256, device=dev)
    mask1024 = r(M, 1024)
    res256 = torch.randn(M, 256, device=dev)
    cs1024 = torch.zeros(2048, device=dev)
    xp, wp = r(48, 925, 256), r(256, 5 * 256)
    xv, wv = r(48, 925, 256), r(256, 3 * 256)
    xe, we = r(48, 128, 256), r(1024, 9 * 256)
    big_a, big_b = r(8192, 4096), r(4096, 4096)
    cases = {
        "ffn1 bias+relu 44400x1024x256": (lambda: ops.linear(x256, w_1024_256, bias1024, relu=True), 2.0 * M * 1024 * 256),
        "ffn1-dgrad mask+colsum 44400x1024x256": (lambda: ops.linear(x256, w_1024_256, relu_mask=mask1024, colsum=cs1024[:1024].zero_()), 2.0 * M * 1024 * 256),
        "ffn2 bias+res f32out 44400x256x1024": (lambda: ops.linear(x1024, w_256_1024, bias256, residual=res256, out_dtype=torch.float32), 2.0 * M * 256 * 1024),
        "ffn2 plain 44400x256x1024": (lambda: ops.linear(x1024, w_256_1024), 2.0 * M * 256 * 1024),
        "qkv 44400x768x256": (lambda: ops.linear(x256, w_768_256), 2.0 * M * 768 * 256),
        "proj 44400x256x256": (lambda: ops.linear(x256, w_256_256, bias256), 2.0 * M * 256 * 256),
        "proj-dgrad res-f32 44400x256x256": (lambda: ops.linear(x256, w_256_256, residual=res256, out_dtype=torch.float32), 2.0 * M * 256 * 256),
        "post_conv k5 44400x256x1280": (lambda: ops.conv(xp, wp, 5, 4, bias=bias256), 2.0 * M * 256 * 1280),
        "va_conv k3 44400x256x768": (lambda: ops.conv(xv, wv, 3, 1, bias=bias256, relu=True), 2.0 * M * 256 * 768),
        "enc_conv1 k9 6144x1024x2304": (lambda: ops.conv(xe, we, 9, 4, bias=bias1024, relu=True), 2.0 * 6144 * 1024 * 2304),
        "square 8192x4096x4096": (lambda: ops.linear(big_a, big_b), 2.0 * 8192 * 4096 * 4096),
    }
    # weight gradients (k-major operands): FS2_GEMM_BIG_KM 0 (4-wave kernel) against 2 (16-wave kernel, gemm_big_km.hip)
    dy1024, dy256, dy768 = r(M, 1024), r(M, 256), r(M, 768)
    xk3, dyk = r(48, 925, 256), r(48, 925, 256)
    xe9, dye = r(48, 128, 256), r(48, 128, 1024)
    wcases = {
        "wgrad ffn1 1024x256 red 44400": (lambda: ops.wgrad(dy1024, x256, torch.zeros(1024, 256, device=dev)), 2.0 * M * 1024 * 256),
        "wgrad ffn2 256x1024 red 44400": (lambda: ops.wgrad(dy256, x1024, torch.zeros(256, 1024, device=dev)), 2.0 * M * 1024 * 256),
        "wgrad proj 256x256 red 44400": (lambda: ops.wgrad(dy256, x256, torch.zeros(256, 256, device=dev)), 2.0 * M * 256 * 256),
        "wgrad qkv 3x(256x256) red 44400": (lambda: ops.wgrad(dy768, x256, torch.zeros(768, 256, device=dev)), 2.0 * M * 768 * 256),
        "conv_wgrad k3 256x(3x256) red 44400": (lambda: ops.conv_wgrad(dyk, xk3, 3, 1, torch.zeros(256, 768, device=dev)), 2.0 * M * 256 * 768),
        "conv_wgrad k5 256x(5x256) red 44400": (lambda: ops.conv_wgrad(dyk, xk3, 5, 4, torch.zeros(256, 1280, device=dev)), 2.0 * M * 256 * 1280),
        "conv_wgrad k9 1024x(9x256) red 6144": (lambda: ops.conv_wgrad(dye, xe9, 9, 4, torch.zeros(1024, 2304, device=dev)), 2.0 * 6144 * 1024 * 2304),
    }
    only = [a for a in sys.argv[1:] if not a.isdigit() and ":" not in a]
    for name, (fn, fl) in wcases.items():
        if only and not any(o in name for o in only):
            continue
        os.environ["FS2_GEMM_BIG_KM"] = "0"
        ref = fn().float()
        t0w, t0c = timeit(fn, False), timeit(fn, True)
        os.environ["FS2_GEMM_BIG_KM"] = "2"
        out = fn().float()
        rel = float((out - ref).abs().max() / ref.abs().max())
        tw, tc = timeit(fn, False), timeit(fn, True)
        print(f"{name:40s} 4-wave {t0w:7.1f}/{t0c:7.1f} us ({fl / t0c / 1e6:6.0f} TF cold) | 16-wave {tw:7.1f}/{tc:7.1f} us ({fl / tc / 1e6:6.0f} TF) maxrel {rel:.1e}", flush=True)
    os.environ["FS2_GEMM_BIG_KM"] = "1"
    for name, (fn, fl) in cases.items():
        if only and not any(o in name for o in only):
            continue
        os.environ["FS2_GEMM_BIG"], os.environ["FS2_GEMM_RING"], os.environ["FS2_GEMM_WS"] = "0", "0", "0"
        ref = fn().float()
        t0w, t0c = timeit(fn, False), timeit(fn, True)
        line = f"{name:40s} 4-wave {t0w:7.1f}/{t0c:7.1f} us ({fl / t0c / 1e6:6.0f} TF cold)"
        for cfg in cfgs:
            os.environ["FS2_GEMM_BIG_BM"] = cfg.split(":")[0]
            for kern, env in (("ring", ("0", "2", "0")), ("ws", ("0", "2", "2"))):      # 16-wave tiled kernel, weights-stationary stream (K = 256)
                os.environ["FS2_GEMM_BIG"], os.environ["FS2_GEMM_RING"], os.environ["FS2_GEMM_WS"] = env
                out = fn().float()
                err = float((out - ref).abs().max())
                rel = err / float(ref.abs().max())
                tw, tc = timeit(fn, False), timeit(fn, True)
                line += f" | {kern}{cfg} {tw:7.1f}/{tc:7.1f} us ({fl / tc / 1e6:6.0f} TF) maxrel {rel:.1e}"
        os.environ.pop("FS2_GEMM_BIG_BM", None)
        print(line, flush=True)
        os.environ["FS2_GEMM_BIG"], os.environ["FS2_GEMM_RING"] = "2", "0"
    os.environ["FS2_GEMM_BIG"], os.environ["FS2_GEMM_RING"], os.environ["FS2_GEMM_WS"] = "1", "1", "1"


if __name__ == "__main__":
    main()

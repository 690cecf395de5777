"""GEMM microbenchmark over the product shapes of BASELINE.json configs[1] (for rocprofv3 / tuning)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from transformer_tts_amd import ops  # noqa: E402

dev = "cuda"
T = torch.bfloat16


def bench(name, fn, flops, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{name:34s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)


_flush = None


def bench_cold(name, fn, flops, iters=8):
    """every launch preceded by a 1 GiB fill, so that no operand is left in L2 / Infinity Cache (in-model condition)"""
    global _flush
    if _flush is None:
        _flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    fn()
    tot = 0.0
    for i in range(iters):
        _flush.fill_(float(i))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    us = tot * 1e3 / iters
    print(f"{name:34s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s   (cold)", flush=True)


def main():
    which = sys.argv[1:] or ["all"]
    cold = "cold" in which
    which = [w for w in which if w != "cold"] or ["all"]
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(T)
    M = 44400
    cases = {}
    x, w = r(M, 256), r(1024, 256)
    cases["dec_ffn1 44400x1024x256"] = (lambda: ops.linear(x, w), 2.0 * M * 1024 * 256)
    bias_, mask_, cs_ = torch.randn(1024, device=dev), r(M, 1024), torch.zeros(2 * 1024 + 4, device=dev)
    outb = torch.empty(M, 1024, device=dev, dtype=T)
    cases["epi ffn1 plain"] = (lambda: ops.linear(x, w, out=outb), 2.0 * M * 1024 * 256)
    cases["epi ffn1 bias+relu"] = (lambda: ops.linear(x, w, bias_, relu=True, out=outb), 2.0 * M * 1024 * 256)
    cases["epi ffn1 relu_mask"] = (lambda: ops.linear(x, w, relu_mask=mask_, out=outb), 2.0 * M * 1024 * 256)
    cases["epi ffn1 colsum"] = (lambda: ops.linear(x, w, colsum=cs_, out=outb), 2.0 * M * 1024 * 256)
    cases["epi ffn1 relu_mask+colsum"] = (lambda: ops.linear(x, w, relu_mask=mask_, colsum=cs_, out=outb), 2.0 * M * 1024 * 256)
    cases["epi ffn1 colstats(sum+sumsq)"] = (lambda: ops.linear(x, w, colstats=cs_, out=outb), 2.0 * M * 1024 * 256)
    cases["epi colsum kernel alone (M,1024)"] = (lambda: ops.colsum(outb, cs_[:1024]), 2.0 * M * 1024)
    x2, w2 = r(M, 1024), r(256, 1024)
    cases["dec_ffn2 44400x256x1024"] = (lambda: ops.linear(x2, w2), 2.0 * M * 256 * 1024)
    xe, we = r(48, 128, 256), r(1024, 9 * 256)
    cases["enc_conv1 6144x1024x(9x256)"] = (lambda: ops.conv(xe, we, 9, 4), 2.0 * 6144 * 1024 * 2304)
    xe2, we2 = r(48, 128, 1024), r(256, 9 * 1024)
    cases["enc_conv2 6144x256x(9x1024)"] = (lambda: ops.conv(xe2, we2, 9, 4), 2.0 * 6144 * 256 * 9216)
    xp, wp = r(48, 925, 256), r(256, 5 * 256)
    cases["post_conv 44400x256x(5x256)"] = (lambda: ops.conv(xp, wp, 5, 4), 2.0 * M * 256 * 1280)
    big_a, big_b = r(8192, 4096), r(4096, 4096)
    cases["square 8192x4096x4096"] = (lambda: ops.linear(big_a, big_b), 2.0 * 8192 * 4096 * 4096)
    t, tp = 925, 928
    qkv = r(48, t, 768)
    q, k, v = (qkv.view(48, t, 3, 2, 128)[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    S = torch.zeros(48, 2, t, tp, device=dev, dtype=T)
    cases["attn_qk 96x(925x925x128)"] = (lambda: ops.bmm(q, k, S[..., :t], trans_b=True), 2.0 * 96 * t * t * 128)
    O = torch.empty(48, t, 2, 128, device=dev, dtype=T)
    cases["attn_pv 96x(925x128x928)"] = (lambda: ops.bmm(S, v, O.permute(0, 2, 1, 3), trans_b=False), 2.0 * 96 * t * 128 * tp)
    dy, xx = r(M, 1024), r(M, 256)
    gw = torch.zeros(1024, 256, device=dev)
    cases["wgrad 1024x256 red 44400"] = (lambda: ops.wgrad(dy, xx, gw), 2.0 * M * 1024 * 256)
    dye, xxe = r(48, 128, 1024), r(48, 128, 256)
    gwe = torch.zeros(1024, 9 * 256, device=dev)
    cases["conv_wgrad 1024x(9x256) red 6144"] = (lambda: ops.conv_wgrad(dye, xxe, 9, 4, gwe), 2.0 * 6144 * 1024 * 2304)
    for (n_, k_, m_) in ((256, 256, 6144), (256, 256, 44400), (768, 256, 44400), (256, 1024, 44400), (80, 256, 44400)):
        dyw, xw = r(m_, n_), r(m_, k_)
        gww = torch.zeros(n_, k_, device=dev)
        for sp in (4, 8, 16, 32, 64):
            cases[f"wsplit {n_}x{k_} red {m_} split {sp}"] = (lambda a=dyw, b=xw, c=gww, sp=sp: ops.wgrad(a, b, c, split=sp), 2.0 * m_ * n_ * k_)
    # calibration only: the vendor library (hipBLASLt through torch.matmul) on the same plain-GEMM shapes
    cases["  lib dec_ffn1"] = (lambda: torch.nn.functional.linear(x, w), 2.0 * M * 1024 * 256)
    cases["  lib dec_ffn2"] = (lambda: torch.nn.functional.linear(x2, w2), 2.0 * M * 256 * 1024)
    cases["  lib square"] = (lambda: torch.nn.functional.linear(big_a, big_b), 2.0 * 8192 * 4096 * 4096)
    qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
    cases["  lib attn_qk"] = (lambda: torch.matmul(qc, kc.transpose(-1, -2)), 2.0 * 96 * t * t * 128)
    Sc = S[..., :t].contiguous()
    cases["  lib attn_pv"] = (lambda: torch.matmul(Sc, vc), 2.0 * 96 * t * 128 * t)
    xim = r(6144, 2304)
    cases["  lib enc_conv1 as plain 6144x1024x2304"] = (lambda: torch.nn.functional.linear(xim, we), 2.0 * 6144 * 1024 * 2304)
    xim2 = r(6144, 9216)
    cases["  lib enc_conv2 as plain 6144x256x9216"] = (lambda: torch.nn.functional.linear(xim2, we2), 2.0 * 6144 * 256 * 9216)
    dyt = dy.t().contiguous()
    cases["  lib wgrad 1024x256 red 44400"] = (lambda: torch.matmul(dyt, xx), 2.0 * M * 1024 * 256)
    for name, (fn, fl) in cases.items():
        if "all" in which or any(wn in name for wn in which):
            (bench_cold if cold else bench)(name, fn, fl)


if __name__ == "__main__":
    main()

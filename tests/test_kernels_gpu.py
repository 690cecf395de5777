"""Per-kernel parity on the MI355X: every C-ABI entry point (through transformer_tts_amd.ops) against its
CPU restatement in oracle/primitives.py on the same seeded inputs.

Tolerances: exact-fp32 mode (f32-input MFMA, fp32 everywhere): rtol 2e-5 / atol 2e-5 (+ accumulation
order); bf16 mode: both sides see identical bf16 inputs, the oracle computes in fp64 and rounds once, the
kernels accumulate in fp32 and round once -> within 2 bf16 ulps (rtol 1.6e-2) plus a small absolute term
for sums that cancel.  Integer / index outputs are bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import primitives as P

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dtype, k=1):
    if dtype == torch.float32:
        return dict(rtol=2e-5, atol=2e-5 * k)
    return dict(rtol=1.6e-2, atol=1.6e-2 * k)


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = np.random.default_rng(hash((shape, seed)) % (2 ** 32))
    return (torch.from_numpy(g.standard_normal(shape).astype(np.float32)) * scale).to(dtype)


def gpu(t):
    return None if t is None else t.cuda()


def close(a, b, what="", **kw):
    torch.testing.assert_close(a.detach().cpu().double(), b.detach().cpu().double(), msg=lambda m: f"{what}: {m}", **kw)


@pytest.fixture(scope="module")
def ops():
    from transformer_tts_amd import ops as o
    o.lib()
    return o


# ------------------------------------------------------------------------------------------------ GEMM family
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(300, 80, 72), (129, 256, 256), (40, 768, 64), (1000, 1024, 8)])
def test_linear_epilogues(ops, dtype, M, N, K):
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias, res, mask = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, dtype=dtype, seed=5)
    for kw in (dict(), dict(bias=True), dict(bias=True, relu=True), dict(residual=True, out_f32=True),
               dict(relu_mask=True), dict(bias=True, colstats=True), dict(alpha=0.25)):
        def call(o, dev):
            mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
            cs = torch.zeros(2 * N, dtype=torch.float32, device=dev) if kw.get("colstats") else None
            out = o.linear(mv(x), mv(w), bias=mv(bias) if kw.get("bias") else None, relu=kw.get("relu", False),
                           residual=mv(res) if kw.get("residual") else None,
                           relu_mask=mv(mask) if kw.get("relu_mask") else None, colstats=cs,
                           out_dtype=torch.float32 if kw.get("out_f32") else None, alpha=kw.get("alpha", 1.0))
            return out, cs
        (a, acs), (b, bcs) = call(ops, "cuda"), call(P, "cpu")
        close(a, b, f"linear {kw}", **tol(a.dtype))
        if acs is not None:
            close(acs, bcs, "colstats", rtol=2e-3, atol=2e-2 * M ** 0.5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,t,C,N,taps,pad", [(3, 37, 32, 128, 3, 1), (2, 50, 80, 256, 5, 4), (3, 37, 64, 64, 9, 4),
                                               (2, 131, 256, 80, 5, 0), (4, 20, 32, 32, 1, 0)])
def test_conv_forward_dgrad_geometry(ops, dtype, B, t, C, N, taps, pad):
    x, w = rnd(B, t, C, dtype=dtype, seed=1), rnd(N, taps * C, dtype=dtype, seed=2, scale=(taps * C) ** -0.5)
    bias = rnd(N, seed=3)
    a = ops.conv(x.cuda(), w.cuda(), taps, pad, bias=bias.cuda(), relu=True)
    b = P.conv(x, w, taps, pad, bias=bias, relu=True)
    close(a, b, "conv", **tol(dtype))
    res = rnd(B, t, N, seed=4)
    a = ops.conv(x.cuda(), w.cuda(), taps, pad, residual=res.cuda(), out_dtype=torch.float32)
    b = P.conv(x, w, taps, pad, residual=res, out_dtype=torch.float32)
    close(a, b, "conv+res f32 out", **tol(dtype))


def test_conv_matches_torch_conv1d(ops):
    """the implicit-GEMM geometry against nn.functional.conv1d itself (symmetric and causal padding)"""
    import torch.nn.functional as F
    for k, pad, crop in ((9, 4, 0), (3, 1, 0), (5, 4, 4)):
        x, w, bias = rnd(2, 45, 16, seed=1), rnd(24, 16, k, seed=2, scale=0.2), rnd(24, seed=3)
        ref = F.conv1d(x.transpose(1, 2), w, bias, padding=pad)
        ref = (ref[:, :, :-crop] if crop else ref).transpose(1, 2)
        shadow = torch.empty(24, k * 16).cuda()
        ops.cast_permute(w.cuda(), shadow, 0)
        out = ops.conv(x.cuda(), shadow, k, pad, bias=bias.cuda())
        close(out, ref, f"conv1d k={k}", rtol=2e-5, atol=2e-5)
        # dgrad: d/dx of sum(out * g)  ==  conv of g with the flipped/transposed shadow, pad' = k-1-pad
        g = rnd(2, 45, 24, seed=4)
        xr = x.clone().requires_grad_(True)
        o2 = F.conv1d(xr.transpose(1, 2), w, bias, padding=pad)
        o2 = (o2[:, :, :-crop] if crop else o2).transpose(1, 2)
        (o2 * g).sum().backward()
        sd = torch.empty(16, k * 24).cuda()
        ops.cast_permute(w.cuda(), sd, 1)
        dx = ops.conv(g.cuda(), sd, k, k - 1 - pad)
        close(dx, xr.grad, f"conv1d dgrad k={k}", rtol=2e-5, atol=2e-5)
        # wgrad in kernel layout, permuted back into the (O,I,k) gradient
        wr = w.clone().requires_grad_(True)
        o3 = F.conv1d(x.transpose(1, 2), wr, bias, padding=pad)
        o3 = (o3[:, :, :-crop] if crop else o3).transpose(1, 2)
        (o3 * g).sum().backward()
        scratch = torch.zeros(24, k * 16).cuda()
        ops.conv_wgrad(g.cuda(), x.cuda(), k, pad, scratch)
        gw = torch.zeros(24, 16, k).cuda()
        ops.permute_add(scratch, gw)
        close(gw, wr.grad, f"conv1d wgrad k={k}", rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("O,I,k", [(24, 16, 5), (70, 130, 9), (33, 65, 3), (256, 1024, 9), (5, 7, 12), (80, 256, 5)])
def test_permute_add_exact(ops, O, I, k):
    """grad[o][i][j] += scratch[o][j*I + i] (tiled kernel for k <= 9, flat kernel otherwise), with and without re-zeroing"""
    scr, g0 = rnd(O, k * I, seed=1), rnd(O, I, k, seed=2)
    want = g0 + scr.view(O, k, I).permute(0, 2, 1)
    for rezero in (False, True):
        s_, g_ = scr.cuda(), g0.cuda()
        ops.permute_add(s_, g_, rezero=rezero)
        assert torch.equal(g_.cpu(), want)
        assert torch.equal(s_.cpu(), torch.zeros_like(scr) if rezero else scr)


def test_linear_random_shapes_and_epilogues(ops):
    """seeded sweep of ragged products (M from 1 row, N and K at their alignment minima, every epilogue combination)
    in both compute modes against the oracle primitive"""
    rs = np.random.default_rng(2024)
    for case in range(24):
        dtype = torch.bfloat16 if case % 2 else torch.float32
        M = int(rs.choice([1, 3, 17, 64, 129, 300, 641]))
        N = int(rs.choice([4, 8, 12, 80, 132, 256, 388])) if dtype == torch.float32 else int(rs.choice([8, 16, 80, 136, 256, 392]))
        K = int(rs.choice([8, 24, 64, 72, 200, 512]))
        x, w = rnd(M, K, dtype=dtype, seed=case), rnd(N, K, dtype=dtype, seed=100 + case, scale=0.3)
        kw = {}
        if rs.random() < 0.6:
            kw["bias"] = rnd(N, seed=200 + case)
        if rs.random() < 0.4:
            kw["relu"] = True
        if rs.random() < 0.4:
            kw["residual"] = rnd(M, N, dtype=dtype if rs.random() < 0.5 else torch.float32, seed=300 + case)
        if rs.random() < 0.3:
            kw["relu_mask"] = rnd(M, N, dtype=dtype, seed=400 + case)
        ref = P.linear(x, w, **kw)
        got = ops.linear(x.cuda(), w.cuda(), **{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()})
        close(got, ref, f"case {case}: {M}x{N}x{K} {dtype} {sorted(kw)}", **tol(dtype, k=K))


def test_gemm_tile_orders_agree(ops):
    """FS2Gemm.tile_order: the n-fastest walk and the XCD-aligned m-fastest walk compute every tile with the same
    arithmetic -> bit-identical outputs; the automatic choice (0) picks one of them; invalid uses are rejected."""
    import ctypes
    from transformer_tts_amd import ops as real_ops
    dtype = torch.bfloat16
    M, N, K = 12800, 512, 192                     # 100 x 4 tiles of 128: "tall" for the automatic choice
    x, w, bias = rnd(M, K, dtype=dtype, seed=1).cuda(), rnd(N, K, dtype=dtype, seed=2).cuda(), rnd(N, seed=3).cuda()
    outs = []
    for order in (1, 2, 0):
        out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
        g = real_ops.FS2Gemm()
        g.A, g.B, g.lda, g.ldb = x.data_ptr(), w.data_ptr(), K, K
        g.M, g.N, g.K, g.dtype = M, N, K, real_ops.BF16
        g.split_k = g.batch1 = g.batch2 = 1
        g.C, g.ldc, g.c_dtype, g.bias, g.alpha, g.relu = out.data_ptr(), N, real_ops.BF16, bias.data_ptr(), 1.0, 1
        g.tile_order = order
        real_ops._gemm_call(g)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    close(outs[1], P.linear(x.cpu(), w.cpu(), bias.cpu(), relu=True), "m-fastest tile walk", rtol=2e-2, atol=2e-2)
    g.tile_order, g.split_k, g.accumulate, g.c_dtype, g.relu, g.bias = 2, 2, 1, real_ops.F32, 0, None
    assert real_ops.lib().fs2_gemm(ctypes.byref(g), None) < 0 and b"tile_order" in real_ops.lib().fs2_last_error()


def test_splitk_forward_products(ops):
    """few output tiles + long reduction: ops.conv / ops.linear run sliced split-K (FS2Gemm.accumulate = 2: one fp32 workspace
    slice per split, plain stores) and finish with fs2_splitk_reduce (bias / ReLU / residual / cast).  Same results as the oracle,
    and a second call (workspace reused, never zeroed) gives bit for bit the same answer."""
    dtype = torch.bfloat16
    B, t, C, N, k = 16, 128, 512, 256, 9
    from transformer_tts_amd import ops as real_ops
    g = real_ops.FS2Gemm(); g.dtype = real_ops.BF16
    assert real_ops._splitk_plan(B * t, N, k * C, g, None, None, None, 1.0) > 1
    x, w = rnd(B, t, C, dtype=dtype, seed=1), rnd(N, k * C, dtype=dtype, seed=2, scale=0.05)
    bias, res = rnd(N, seed=3), rnd(B, t, N, dtype=dtype, seed=4)
    for kw in (dict(bias=bias), dict(bias=bias, relu=True), dict(residual=res), dict(bias=bias, residual=res.float(), out_dtype=torch.float32)):
        ref = P.conv(x, w, k, k // 2, **kw)
        cu = {kk: (v.cuda() if torch.is_tensor(v) else v) for kk, v in kw.items()}
        gots = []
        for _ in range(2):
            got = ops.conv(x.cuda(), w.cuda(), k, k // 2, **cu)
            close(got, ref, f"split-K conv {sorted(kw)}", rtol=2e-2, atol=2e-2)
            gots.append(got)
            for ws in real_ops._splitk_scratch.values():
                ws.fill_(float("nan"))                      # the workspace needs no clean state
        assert torch.equal(gots[0], gots[1])
    assert real_ops.lib().fs2_gemm_last_splits() > 1
    xl, wl = rnd(1000, 4096, dtype=dtype, seed=5), rnd(128, 4096, dtype=dtype, seed=6, scale=0.05)
    close(ops.linear(xl.cuda(), wl.cuda(), bias[:128].cuda()), P.linear(xl, wl, bias[:128]), "split-K linear", rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", DT)
def test_wgrad_and_colsum(ops, dtype):
    for M, N, K in ((300, 80, 72), (2000, 256, 1024), (77, 768, 32)):
        dy, x = rnd(M, N, dtype=dtype, seed=1), rnd(M, K, dtype=dtype, seed=2)
        out0 = rnd(N, K, seed=3)
        a = ops.wgrad(dy.cuda(), x.cuda(), out0.cuda())
        b = P.wgrad(dy, x, out0.clone())
        close(a, b, "wgrad", rtol=1e-4, atol=1e-4 * M ** 0.5)
        # strided column slice as A (the q/k/v slices of dqkv) and colsum on it
        dq = rnd(M, 3 * N, dtype=dtype, seed=4)
        sl = slice(N, 2 * N)
        a = ops.wgrad(dq.cuda()[:, sl], x.cuda(), torch.zeros(N, K).cuda())
        b = P.wgrad(dq[:, sl], x, torch.zeros(N, K))
        close(a, b, "wgrad slice", rtol=1e-4, atol=1e-4 * M ** 0.5)
        a = ops.colsum(dq.cuda()[:, sl], torch.zeros(N).cuda())
        b = P.colsum(dq[:, sl], torch.zeros(N))
        close(a, b, "colsum", rtol=1e-4, atol=1e-4 * M ** 0.5)
        arena = rnd(3 * (N * K + N), seed=5)
        # column sums of the three blocks straight into vectors at a constant stride (q/v/k bias gradients), and fallback
        for regular in (True, False):
            ba, bb = arena.clone().cuda(), arena.clone()
            offs = [N * K + j * (N * K + N) for j in range(3)] if regular else [N * K, 3 * (N * K + N) - N, 2 * N * K + N]
            ops.colsum_blocks(dq.cuda(), [ba[o:o + N] for o in offs])
            P.colsum_blocks(dq, [bb[o:o + N] for o in offs])
            close(ba, bb, f"colsum_blocks regular={regular}", rtol=1e-4, atol=1e-4 * M ** 0.5)
        # the three column blocks of dqkv in one batched launch: outputs at a constant stride inside one arena
        # (weights interleaved with their biases, as the parameter arena lays q/v/k out), and the irregular fallback
        for regular in (True, False):
            ga, gb = arena.clone().cuda(), arena.clone()
            offs = [j * (N * K + N) for j in range(3)] if regular else [0, 2 * (N * K + N), N * K + N]
            outs_a = [ga[o:o + N * K].view(N, K) for o in offs]
            outs_b = [gb[o:o + N * K].view(N, K) for o in offs]
            ops.wgrad_batched(dq.cuda(), x.cuda(), outs_a)
            P.wgrad_batched(dq, x, outs_b)
            close(ga, gb, f"wgrad_batched regular={regular}", rtol=1e-4, atol=1e-4 * M ** 0.5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,t,C,N,taps,pad", [(3, 37, 32, 64, 3, 1), (2, 50, 80, 256, 5, 4), (3, 37, 64, 256, 9, 4)])
def test_conv_wgrad(ops, dtype, B, t, C, N, taps, pad):
    dy, x = rnd(B, t, N, dtype=dtype, seed=1), rnd(B, t, C, dtype=dtype, seed=2)
    a = ops.conv_wgrad(dy.cuda(), x.cuda(), taps, pad, torch.zeros(N, taps * C).cuda())
    b = P.conv_wgrad(dy, x, taps, pad, torch.zeros(N, taps * C))
    close(a, b, "conv_wgrad", rtol=1e-4, atol=1e-4 * (B * t) ** 0.5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("t", [37, 128, 200])
def test_attention_products(ops, dtype, t):
    """the five batched products of attention forward/backward on the strided q/k/v views and the
    padded (B,N,H,t,tp) probability buffer"""
    B, H, dk, NL = 2, 2, 16, 2
    d = H * dk
    tp = (t + 7) // 8 * 8

    def views(dev):
        mv = (lambda x: x.cuda()) if dev == "cuda" else (lambda x: x.clone())
        qkv = mv(rnd(B, t, 3 * d, dtype=dtype, seed=1))
        q, k, v = (qkv.view(B, t, 3, H, dk)[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        pbuf = mv(rnd(B, NL, H, t, tp, dtype=dtype, seed=2))
        pbuf[..., t:] = 0
        return q, k, v, pbuf

    res = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        q, k, v, pbuf = views(dev)
        S = torch.full((B, NL, H, t, tp), 7.0, dtype=dtype, device=dev)
        o.bmm(q, k, S[:, 1][..., :t], trans_b=True, alpha=0.25)
        Pm = pbuf[:, 1]
        O = torch.zeros((B, t, H, dk), dtype=dtype, device=dev)
        o.bmm(Pm, v, O.permute(0, 2, 1, 3), trans_b=False)
        dqkv = torch.zeros((B, t, 3 * d), dtype=dtype, device=dev)
        dq, dk_, dv = (dqkv.view(B, t, 3, H, dk)[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        dO4 = O.permute(0, 2, 1, 3)
        o.bmm(Pm, dO4, dv, trans_a=True, trans_b=False)
        o.bmm(Pm, k, dq, trans_b=False, alpha=0.5)
        o.bmm(Pm, q, dk_, trans_a=True, trans_b=False, alpha=0.5)
        res[dev] = (S[:, 1][..., :t].clone(), O, dqkv, S[:, 0].clone())
    for a, b, n in zip(res["cuda"], res["cpu"], ("QK^T", "PV", "dqkv", "untouched layer slice")):
        close(a, b, n, **tol(dtype, k=16))     # sums of up to 200 O(1) terms: fp32 accumulation-order noise


# ------------------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("d", [32, 256, 384])
def test_layernorm_family(ops, dtype, p, d):
    M = 203
    x32, xt = rnd(M, d, seed=1), rnd(M, d, dtype=dtype, seed=2)
    a, h = rnd(M, d, dtype=dtype, seed=3), rnd(M, d, dtype=dtype, seed=4)
    gm, bt = 1 + 0.1 * rnd(d, seed=5), 0.1 * rnd(d, seed=6)
    dy, dsd = rnd(M, d, dtype=dtype, seed=7), rnd(M, d, seed=8)
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        rng = o.Rng(99, dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        r = []
        # plain LN fp32 -> T and T -> T (+dropout), backward with relu mask / accumulate
        y, mu, rs = o.layernorm_fwd(mv(x32), mv(gm), mv(bt), dtype, 1e-5, p, rng, 5)
        dg, db = z(d), z(d)
        dx = o.layernorm_bwd(mv(dy), mv(x32), mv(gm), mu, rs, dg, db, p, rng, 5, dx=mv(dsd))
        r += [y, mu, rs, dx, dg, db]
        y, mu, rs = o.layernorm_fwd(mv(xt), mv(gm), mv(bt), dtype, 1e-5, p, rng, 6)
        dg, db = z(d), z(d)
        dx = o.layernorm_bwd(mv(dy), mv(xt), mv(gm), mu, rs, dg, db, p, rng, 6, relu_mask=True)
        r += [y, dx, dg, db]
        # residual + dropout + LN
        s, y, mu, rs = o.add_ln_fwd(mv(x32), mv(a), mv(gm), mv(bt), 1e-5, p, rng, 7)
        dg, db = z(d), z(d)
        dr, da = o.add_ln_bwd(mv(dsd), mv(dy), s, mv(gm), mu, rs, dg, db, p, rng, 7)
        dr2, _ = o.add_ln_bwd(None, mv(dy), s, mv(gm), mu, rs, z(d), z(d), p, rng, 7)
        r += [s, y, dr, da, dg, db, dr2]
        # FFN tail
        y, mu, rs = o.ffn_ln_fwd(mv(a), mv(h), mv(gm), mv(bt), 1e-5, p, rng, 8)
        dg, db = z(d), z(d)
        g = o.ffn_ln_bwd(mv(dy), mv(a), mv(h), mv(gm), mu, rs, dg, db, p, rng, 8)
        r += [y, g, dg, db]
        out[dev] = r
    for i, (a_, b_) in enumerate(zip(out["cuda"], out["cpu"])):
        k = 30 if a_.dim() == 1 and a_.numel() == d else 2     # per-channel sums over 203 rows
        close(a_, b_, f"layernorm family output #{i}", **tol(a_.dtype if a_.dtype != torch.float32 else dtype, k=k))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("t", [37, 128, 300, 925, 1100])      # bf16: 1, 2 and 3 sixteen-byte groups per lane
def test_softmax_fwd_bwd(ops, dtype, p, t):
    B, H, NL = 3, 2, 2
    tp = (t + 7) // 8 * 8
    lens = [t, max(1, t // 2), max(1, t - 5)]
    km = torch.zeros(B, t, dtype=torch.bool)
    for b, n in enumerate(lens):
        km[b, :n] = True
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda x: x.cuda()) if dev == "cuda" else (lambda x: x.clone())
        rng = o.Rng(5, dev)
        buf = mv(rnd(B, NL, H, t, tp, dtype=dtype, seed=1, scale=3.0))
        bufd = torch.zeros_like(buf) if p > 0 else buf
        S, Pd = buf[:, 1], bufd[:, 1]
        o.softmax_fwd(S, Pd, mv(km), t, p, rng, 11)
        dP = mv(rnd(B, H, t, tp, dtype=dtype, seed=2))
        dP[..., t:] = float("nan")          # pad columns are never written by the producer GEMM
        o.softmax_bwd(dP, S, t, p, rng, 11)
        out[dev] = (S.clone(), Pd.clone(), dP)
    for a, b, n in zip(out["cuda"], out["cpu"], ("P", "P_drop", "dS")):
        close(a, b, n, **tol(dtype))
    Pc = out["cuda"][0].float().cpu()
    assert torch.all(Pc[..., t:] == 0), "pad columns must be written as zero"
    close(Pc[..., :t].sum(-1), torch.ones(B, H, t), "rows sum to one", rtol=1e-2, atol=1e-2)
    assert float(Pc[1, :, :, lens[1]:t].abs().max()) < 1e-6, "masked keys get ~0 probability (-1e4 fill)"


@pytest.mark.parametrize("t,H,dk", [(1, 2, 128), (8, 1, 32), (63, 2, 64), (65, 2, 128), (127, 1, 128), (129, 1, 32)])
def test_attn_strip_kernels_at_tile_boundaries(ops, t, H, dk):
    """single-frame sequences, one key short of / one key past the 64-key tiles and 128-key super-tiles: forward
    probabilities (+ P V) and backward dS (+ dQ) of the LDS-strip kernels against the oracle composition"""
    dtype = torch.bfloat16
    B = 2
    tp = (t + 7) // 8 * 8
    km = torch.ones(B, t, dtype=torch.bool)
    km[1, max(1, t // 2):] = False
    qkv = rnd(B, t, 3, H, dk, dtype=dtype, seed=1, scale=1.2)
    dO = rnd(B, t, H, dk, dtype=dtype, seed=2)
    res = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda x: x.cuda()) if dev == "cuda" else (lambda x: x.clone())
        rng = o.Rng(9, dev)
        x, g = mv(qkv), mv(dO)
        q, v, k = (x[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        Pb, Pd, dS = (mv(torch.full((B, H, t, tp), float("nan"), dtype=dtype)) for _ in range(3))
        second = o.attn_second_product_supported(dk)
        O = mv(torch.zeros(B, t, H, dk, dtype=dtype))
        dq = mv(torch.zeros(B, t, H, dk, dtype=dtype))
        o.attn_probs_fwd(q, k, mv(km), Pb, Pd, t, dk ** -0.5, 0.1, rng, 4, v=v if second else None,
                         out=O.permute(0, 2, 1, 3) if second else None)
        res[dev] = [Pb.float().cpu(), Pd.float().cpu(), O.float().cpu()]
        Pref = mv(res["cuda"][0].to(dtype)) if dev == "cpu" else Pb     # same saved probabilities on both sides
        o.attn_ds_bwd(g.permute(0, 2, 1, 3), v, Pref, dS, t, 0.1, rng, 4, k=k if second else None,
                      dq=dq.permute(0, 2, 1, 3) if second else None, alpha=dk ** -0.5)
        res[dev] += [dS.float().cpu(), dq.float().cpu()]
    for a, b, n in zip(res["cuda"], res["cpu"], ("P", "P_drop", "P_drop @ V", "dS", "dQ")):
        scale = b.abs().amax().clamp_min(1e-3)
        assert float((a - b).abs().max() / scale) < 5e-2, (n, t, float((a - b).abs().max()), float(scale))
    assert torch.all(res["cuda"][0][..., t:] == 0) and torch.all(res["cuda"][3][..., t:] == 0)


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("t,H,dk", [(37, 2, 32), (64, 1, 64), (130, 2, 128), (925, 2, 128), (1013, 1, 64)])
def test_attn_probs_fused_vs_gemm_softmax(ops, p, t, H, dk):
    """fs2_attn_probs_fwd (scores kept in LDS) against the oracle's bmm + softmax_fwd on the same fused-qkv layout,
    key masks of different lengths, the same Philox counters (so the dropout masks must coincide)."""
    dtype = torch.bfloat16
    B, NL = 3, 2
    tp = (t + 7) // 8 * 8
    assert P.attn_probs_supported(t, dk, dtype) and ops.attn_probs_supported(t, dk, dtype)
    lens = [t, max(1, t // 2), max(1, t - 5)]
    km = torch.zeros(B, t, dtype=torch.bool)
    for b, n in enumerate(lens):
        km[b, :n] = True
    qkv = rnd(B, t, 3, H, dk, dtype=dtype, seed=1, scale=1.5)
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda x: x.cuda()) if dev == "cuda" else (lambda x: x.clone())
        rng = o.Rng(5, dev)
        x = mv(qkv)
        q, k = (x[:, :, j].permute(0, 2, 1, 3) for j in (0, 2))
        buf = mv(torch.full((B, NL, H, t, tp), float("nan"), dtype=dtype))
        bufd = mv(torch.full((B, NL, H, t, tp), float("nan"), dtype=dtype)) if p > 0 else buf
        second = o.attn_second_product_supported(dk)
        O = mv(torch.full((B, t, H, dk), float("nan"), dtype=dtype))
        o.attn_probs_fwd(q, k, mv(km), buf[:, 1], bufd[:, 1], t, 1.0 / dk ** 0.5, p, rng, 11,
                         v=x[:, :, 1].permute(0, 2, 1, 3) if second else None, out=O.permute(0, 2, 1, 3) if second else None)
        out[dev] = (buf[:, 1].clone(), bufd[:, 1].clone(), O if second else None)
    # scores are rounded to bf16 before the softmax on both sides; a 1-ulp difference there moves a probability by ~1 %
    for a, b, n in zip(out["cuda"][:2], out["cpu"][:2], ("P", "P_drop")):
        close(a, b, n, rtol=4e-2, atol=2e-3)
    if out["cuda"][2] is not None:        # dropout(P) V from the same LDS strip
        close(out["cuda"][2], out["cpu"][2], "P_drop @ V", rtol=3e-2, atol=3e-2)
    Pc, Pdc = (x.float().cpu() for x in out["cuda"][:2])
    assert torch.all(Pc[..., t:] == 0) and torch.all(Pdc[..., t:] == 0), "pad columns must be written as zero"
    close(Pc[..., :t].sum(-1), torch.ones(B, H, t), "rows sum to one", rtol=1e-2, atol=1e-2)
    assert float(Pc[1, :, :, lens[1]:t].abs().max()) < 1e-6, "masked keys get ~0 probability"
    if p > 0:      # identical dropout pattern: zeros of P_drop where P is not tiny must coincide with the oracle's
        big = out["cpu"][0].float() > 1e-3
        assert torch.equal((Pdc == 0) & big, (out["cpu"][1].float() == 0) & big)
    assert not ops.attn_probs_supported(1100, 128, dtype) and not ops.attn_probs_supported(100, 48, dtype)


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("t,H,dk", [(37, 2, 32), (64, 1, 64), (130, 2, 128), (925, 2, 128), (1013, 1, 64)])
def test_attn_ds_fused_vs_gemm_softmax_bwd(ops, p, t, H, dk):
    """fs2_attn_ds_bwd (dP = dO V^T kept in LDS) against the oracle's bmm + softmax_bwd: dO in the (B,t,H,dk) layout
    of the output projection's data gradient, V inside the fused qkv tensor, P from a real forward."""
    dtype = torch.bfloat16
    B, NL = 3, 2
    tp = (t + 7) // 8 * 8
    lens = [t, max(1, t // 2), max(1, t - 5)]
    km = torch.zeros(B, t, dtype=torch.bool)
    for b, n in enumerate(lens):
        km[b, :n] = True
    qkv = rnd(B, t, 3, H, dk, dtype=dtype, seed=1, scale=1.5)
    dO = rnd(B, t, H, dk, dtype=dtype, seed=2)
    # saved probabilities: the oracle's forward (shared by both sides, so only the backward is compared)
    rng0 = P.Rng(5, "cpu")
    Pbuf = torch.zeros(B, NL, H, t, tp, dtype=dtype)
    Pdbuf = torch.zeros(B, NL, H, t, tp, dtype=dtype)
    q, v, k = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    P.attn_probs_fwd(q, k, km, Pbuf[:, 1], Pdbuf[:, 1], t, 1.0 / dk ** 0.5, p, rng0, 11)
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda x: x.cuda()) if dev == "cuda" else (lambda x: x.clone())
        rng = o.Rng(5, dev)
        x, g, pb = mv(qkv), mv(dO), mv(Pbuf)
        vv = x[:, :, 1].permute(0, 2, 1, 3)
        dS = mv(torch.full((B, H, t, tp), float("nan"), dtype=dtype))
        second = o.attn_second_product_supported(dk)
        dqkv = mv(torch.full((B, t, 3, H, dk), float("nan"), dtype=dtype))
        o.attn_ds_bwd(g.permute(0, 2, 1, 3), vv, pb[:, 1], dS, t, p, rng, 11,
                      k=x[:, :, 2].permute(0, 2, 1, 3) if second else None,
                      dq=dqkv[:, :, 0].permute(0, 2, 1, 3) if second else None, alpha=1.0 / dk ** 0.5)
        out[dev] = dS
        out[dev + "_dq"] = dqkv[:, :, 0].float().cpu() if second else None
    if out["cuda_dq"] is not None:        # dQ = dS K / sqrt(dk) from the same LDS strip
        ref = out["cpu_dq"]
        err = (out["cuda_dq"] - ref).abs().max() / ref.abs().max().clamp_min(1e-6)
        assert float(err) < 3e-2, float(err)
    a, b_ = out["cuda"].float().cpu(), out["cpu"].float()
    assert torch.all(a[..., t:] == 0), "pad columns must be written as zero"
    # dP is rounded to bf16 on both sides before the softmax backward; compare against the scale of each row
    scale = b_.abs().amax(-1, keepdim=True).clamp_min(1e-6)
    assert float(((a - b_).abs() / scale).max()) < 4e-2, float(((a - b_).abs() / scale).max())


def test_flash_mask_info_ranks_rows_longest_first(ops):
    """fs2_flash_attn_mask_info on a batch larger than one block of its ranking kernel: {kfull, kmax} per row (holes in a mask make
    them differ) and the rank list the flash kernels start their workgroups from"""
    B, t = 300, 200
    g = np.random.default_rng(5)
    lens = g.integers(1, t + 1, size=B)
    lens[:3] = [t, 1, t]                      # ties and the extremes
    km = torch.from_numpy(np.arange(t)[None, :] < lens[:, None])
    km[7, : int(lens[7])] = True
    if lens[7] > 4:
        km[7, 2] = False                      # a hole: kfull = 2, kmax = lens[7]
    info = ops.flash_mask_info(km.cuda()).cpu()
    kfull = [2 if (i == 7 and lens[7] > 4) else int(n) for i, n in enumerate(lens)]
    assert info[:, 0].tolist() == kfull and info[:, 1].tolist() == [int(n) for n in lens]
    assert info[:, 2].tolist() == sorted(range(B), key=lambda i: (-int(lens[i]), i))


@pytest.mark.parametrize("B,t", [(300, 200), (1100, 37), (48, 925), (1, 1)])
def test_flash_mask_info_one_launch_and_two_launch_forms(ops, B, t):
    """row bounds and ranking of the key masks for a few batch sizes (up to more rows than one block of the ranking kernel has threads) against the host computation"""
    g = np.random.default_rng(B + t)
    lens = g.integers(1, t + 1, size=B)
    km = torch.from_numpy(np.arange(t)[None, :] < lens[:, None])
    info = ops.flash_mask_info(km.cuda()).cpu()
    assert info[:, 0].tolist() == info[:, 1].tolist() == [int(n) for n in lens]
    assert info[:, 2].tolist() == sorted(range(B), key=lambda i: (-int(lens[i]), i))


@pytest.mark.parametrize("B,t,pad", [(48, 925, 0), (3, 7, 0), (1, 1, 0), (130, 129, 5), (1024, 33, 0)])
def test_pad_mask_info_is_create_masks_plus_mask_info(ops, B, t, pad):
    """fs2_pad_mask_info = the reference's (pos != pad) of create_masks (train_fastspeech2.py:55-82) + fs2_flash_attn_mask_info, bit for
    bit, including positions that hit the pad value in the middle of a row and all-pad rows"""
    g = np.random.default_rng(B * 7 + t)
    lens = g.integers(0, t + 1, size=B)
    lens[0] = t
    pos = torch.from_numpy(np.where(np.arange(t)[None, :] < lens[:, None], np.arange(1, t + 1)[None, :] + (pad if pad else 0), pad)).long()
    if t > 4:
        pos[B // 2, 2] = pad                  # a hole
    mask, info = ops.pad_mask_info(pos.cuda(), pad)
    want = pos != pad
    assert mask.dtype == torch.bool and torch.equal(mask.cpu(), want)
    ref = ops.flash_mask_info(want.cuda()).cpu()
    assert torch.equal(info.cpu(), ref)
    from transformer_tts_amd.train_fastspeech2 import create_masks
    s, m = create_masks(pos.cuda(), pos.cuda(), task="fastspeech2", src_pad=pad, trg_pad=pad)
    assert s.shape == (B, 1, t) and torch.equal(s.cpu(), want.unsqueeze(-2)) and torch.equal(m.cpu(), want.unsqueeze(-2))
    assert torch.equal(s._fs2_kinfo.cpu(), ref)
    if t > 1:       # a row-strided view (the autoregressive trainer's pos_mel[:, :-1]) is read in place
        sub = pos.cuda()[:, :-1]
        mask2, info2 = ops.pad_mask_info(sub, pad)
        assert torch.equal(mask2.cpu(), want[:, :-1]) and torch.equal(info2.cpu(), ops.flash_mask_info(want[:, :-1].contiguous().cuda()).cpu())


def _oracle_attention(qkv, dO, km, t, p, seed, site, NL=2, layer=1):
    """attention() forward + backward composed from the oracle's primitives on the fused-qkv layout: O, dqkv."""
    B, _, _, H, dk = qkv.shape
    tp = (t + 7) // 8 * 8
    rng = P.Rng(seed, "cpu")
    q, v, k = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    Pb = torch.zeros(B, NL, H, t, tp, dtype=qkv.dtype)
    Pd = torch.zeros(B, NL, H, t, tp, dtype=qkv.dtype) if p > 0 else Pb
    O = torch.zeros(B, t, H, dk, dtype=qkv.dtype)
    P.attn_probs_fwd(q, k, km, Pb[:, layer], Pd[:, layer], t, dk ** -0.5, p, rng, site, v=v, out=O.permute(0, 2, 1, 3))
    dqkv = torch.zeros_like(qkv)
    dq, dv, dk_ = (dqkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    g4 = dO.permute(0, 2, 1, 3)
    P.bmm(Pd[:, layer], g4, dv, trans_a=True, trans_b=False)
    dS = torch.zeros(B, H, t, tp, dtype=qkv.dtype)
    P.attn_ds_bwd(g4, v, Pb[:, layer], dS, t, p, rng, site, k=k, dq=dq, alpha=dk ** -0.5)
    P.bmm(dS, q, dk_, trans_a=True, trans_b=False, alpha=dk ** -0.5)
    return O, dqkv, Pb[:, layer], Pd[:, layer]


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("t,H", [(1, 2), (63, 1), (65, 2), (130, 2), (925, 2), (1013, 1), (1024, 1)])
def test_flash_attention_fwd_bwd(ops, p, t, H):
    """fs2_flash_attn_fwd / _bwd (no probabilities in HBM) against attention() composed from the oracle's primitives, on the
    fused-qkv layout of the FFT block, ragged key masks, the dropout mask of the strip path (same Philox counters)."""
    dtype, dk, B, NL = torch.bfloat16, 128, 3, 2
    tp = (t + 7) // 8 * 8
    assert ops.flash_attn_supported(t, dk, dtype) and ops.flash_attn_supported(1025, dk, dtype) and not ops.flash_attn_supported(t, 48, dtype)
    lens = [t, max(1, t // 2), max(1, t - 5)]
    km = torch.zeros(B, t, dtype=torch.bool)
    for b, n in enumerate(lens):
        km[b, :n] = True
    qkv = rnd(B, t, 3, H, dk, dtype=dtype, seed=1, scale=1.5)
    dO = rnd(B, t, H, dk, dtype=dtype, seed=2)
    O_ref, dqkv_ref, P_ref, Pd_ref = _oracle_attention(qkv, dO, km, t, p, 5, 11, NL)
    x, g = qkv.cuda(), dO.cuda()
    q, v, k = (x[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    rng = ops.Rng(5, "cuda")
    O = torch.full((B, t, H, dk), float("nan"), dtype=dtype, device="cuda")
    stats = torch.full((B, H, t, 2), float("nan"), device="cuda")
    p_batch = NL * H * t * tp
    keep = torch.empty(ops.flash_attn_keep_words(B, H, t), dtype=torch.int16, device="cuda") if p > 0 else None
    # with dropout: the mask rows scanned once (fs2_flash_attn_mask_info); without: every workgroup scans its own row
    kinfo = ops.flash_mask_info(km.cuda()) if p > 0 else None
    if kinfo is not None:
        ki = kinfo.cpu()
        assert ki[:, :2].tolist() == [[n, n] for n in lens]
        order = ki[:, 2].tolist()       # rows ranked by length, longest first, ties in batch order
        assert order == sorted(range(len(lens)), key=lambda i: (-lens[i], i)), order
    ops.flash_attn_fwd(q, k, v, km.cuda(), O.permute(0, 2, 1, 3), stats, keep, t, dk ** -0.5, p_batch, p, rng, 11, key_info=kinfo)
    # the oracle composition rounds the scores to bf16 before the softmax (as the unfused reference does in bf16), this path
    # keeps them in fp32: compare against the scale of the output
    err = float((O.float().cpu() - O_ref.float()).abs().max() / O_ref.float().abs().amax().clamp_min(1e-2))
    assert err < 3e-2, ("O", t, err)
    # the attention map written after the fact (fs2_flash_attention_probs): the post-dropout probabilities of the strip path, the SAME
    # dropped positions, zeros in the pad columns [t, tp) and beyond every row's last visible key
    maps = torch.full((B, NL, H, t, tp), float("nan"), dtype=dtype, device="cuda")
    ops.flash_attention_probs(q, k, v, km.cuda(), O.permute(0, 2, 1, 3), stats, keep, maps[:, 1], dk ** -0.5, p, key_info=kinfo)
    got, want = maps[:, 1].float().cpu(), Pd_ref.float()
    assert torch.isnan(maps[:, 0]).all() and not torch.isnan(got).any()
    assert float(got[..., t:].abs().max() if tp > t else 0.0) == 0.0
    for b, n in enumerate(lens):
        assert float(got[b, :, :, n:].abs().max() if n < tp else 0.0) <= 1e-30, ("masked keys", b)
    assert float((got - want).abs().max()) < 3e-2 * float(want.max()), ("map", t, float((got - want).abs().max()), float(want.max()))
    if p > 0:
        vis = torch.zeros(B, 1, 1, tp, dtype=torch.bool)
        for b, n in enumerate(lens):
            vis[b, ..., :n] = True
        big = vis & (P_ref.float() > 1e-3)        # (pre-dropout probability well away from underflow: a zero there is a dropped position)
        assert bool(((got == 0) == (want == 0))[big.expand_as(want)].all()), "the map's dropped positions differ from the strip path's"
    close(O.float().cpu().mean(), O_ref.float().mean(), "mean of O", rtol=2e-2, atol=2e-3)
    # statistics: the row maximum of the masked scaled scores and the sum of exponentials
    s = torch.einsum("bthd,bshd->bhts", qkv[:, :, 0].float(), qkv[:, :, 2].float()) * dk ** -0.5
    s = s.masked_fill(~km[:, None, None, :], -1e4)
    close(stats[..., 0].cpu(), s.amax(-1), "row maximum", rtol=1e-3, atol=2e-3)
    close(stats[..., 1].cpu(), torch.exp(s - s.amax(-1, keepdim=True)).sum(-1), "sum of exponentials", rtol=2e-3, atol=1e-3)
    if p > 0:       # masks drawn ahead of time by fs2_flash_attn_keep_bits: the same bits, the same output
        keep2 = torch.empty_like(keep)
        ops.flash_keep_bits(keep2, B, H, t, p_batch, p, rng, 11)
        O2, stats2 = torch.empty_like(O), torch.empty_like(stats)
        ops.flash_attn_fwd(q, k, v, km.cuda(), O2.permute(0, 2, 1, 3), stats2, keep2, t, dk ** -0.5, p_batch, p, rng, 11, pregenerated=True)
        assert torch.equal(O2, O) and torch.equal(stats2, stats)
        nkt = (t + 63) // 64
        kk = torch.arange(nkt * 4, device="cuda").view(nkt, 1, 4) * 16 < torch.tensor([t, max(1, t // 2), max(1, t - 5)], device="cuda").view(B, 1, 1, 1, 1)
        a_, b_ = keep.view(B, H, nkt, t, 4), keep2.view(B, H, nkt, t, 4)      # words of key tiles the forward skipped are never written
        assert torch.equal(a_[kk.expand_as(a_)], b_[kk.expand_as(b_)])
    dqkv = torch.full((B, t, 3, H, dk), float("nan"), dtype=dtype, device="cuda")
    dq, dv, dk_ = (dqkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    aux = torch.empty((B, H, t, 4), device="cuda")
    dbias = [torch.full((H * dk,), 0.5, device="cuda") for _ in range(3)]         # accumulated into: += column sums
    ops.flash_attn_bwd(q, k, v, km.cuda(), O.permute(0, 2, 1, 3), g.permute(0, 2, 1, 3), stats, keep, aux, dq, dk_, dv, t, dk ** -0.5, p,
                       dbias=dbias, key_info=kinfo)
    for x, j, n in zip(dbias, (0, 2, 1), ("dbias_q", "dbias_k", "dbias_v")):
        # the kernel sums its fp32 accumulators, the comparison sums the bf16 rows it stored: B*t rounding errors of 2^-9 relative
        # (the k column sums are exactly zero in real arithmetic -- sum_k dS = 0 -- so there this noise is all there is)
        rows = dqkv[:, :, j].float()
        ref_sum = rows.sum((0, 1)).reshape(-1).cpu() + 0.5
        noise = 4.0 * (B * t) ** 0.5 * 2.0 ** -9 * float(rows.abs().max())
        close(x.cpu(), ref_sum, n, rtol=2e-2, atol=2e-2 + noise)
    got, ref = dqkv.float().cpu(), dqkv_ref.float()
    assert torch.isfinite(got).all()
    for j, n in enumerate(("dQ", "dV", "dK")):
        # t = 1: dS = 0 exactly and the reference holds rounding noise; here delta = rowsum(dO * bf16(O)) leaves ~3e-2 |dO| |K| / sqrt(dk)
        scale = ref[:, :, j].abs().amax().clamp_min(1.0 if t == 1 else 0.05)
        err = float((got[:, :, j] - ref[:, :, j]).abs().max() / scale)
        assert err < 3e-2, (n, t, err)
    if p > 0 and t >= 63:
        # the same dropout mask as the strip path: with V = identity columns the output IS dropout(P)[:, :128]
        eye = torch.zeros(B, t, 3, H, dk, dtype=dtype)
        eye[:, :, 0], eye[:, :, 2] = qkv[:, :, 0], qkv[:, :, 2]
        n = min(t, dk)
        eye[:, torch.arange(n), 1, :, torch.arange(n)] = 1.0
        xe = eye.cuda()
        qe, ve, ke = (xe[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        Oe = torch.zeros((B, t, H, dk), dtype=dtype, device="cuda")
        ops.flash_attn_fwd(qe, ke, ve, km.cuda(), Oe.permute(0, 2, 1, 3), stats, torch.empty_like(keep), t, dk ** -0.5, p_batch, p, rng, 11)
        pd_flash = Oe.permute(0, 2, 1, 3)[..., :n].float().cpu()
        big = P_ref[..., :n].float() > 1e-3
        assert torch.equal((pd_flash == 0) & big, (Pd_ref[..., :n].float() == 0) & big)


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("tq,tk,dk,causal", [(1025, 1025, 128, False), (1500, 1500, 64, False), (2047, 2047, 96, True), (37, 37, 64, True),
                                             (65, 65, 128, True), (200, 200, 96, False), (300, 77, 128, False), (130, 925, 96, False),
                                             (9, 130, 64, False), (1, 1, 128, True)])
def test_flash_attention_general(ops, p, tq, tk, dk, causal):
    """fs2_flash_attention_fwd / _bwd (general descriptor): more than 1024 keys (the key-mask row streams through a dynamically sized
    LDS image), d_k in {64, 96, 128}, the causal (no-peak) self-attention and the rectangular encoder-decoder attention of the
    autoregressive decoder (Models/layers.py:108-118, masks train.py:26-58) -- against fp32 attention() with the keep-mask the
    oracle's rectangular softmax draws from the same Philox counters.  Tolerance: 3e-2 of each tensor's scale (bf16 operands)."""
    dtype, B, H, NL = torch.bfloat16, 2, 2, 2
    tkp = (tk + 7) // 8 * 8
    lens = [tk, max(1, tk - 5 if tk > 8 else tk // 2)]
    km = torch.zeros(B, tk, dtype=torch.bool)
    for b, n in enumerate(lens):
        km[b, :n] = True
    q, dO = rnd(B, tq, H, dk, dtype=dtype, seed=1, scale=1.5), rnd(B, tq, H, dk, dtype=dtype, seed=2)
    vk = rnd(B, tk, 2, H, dk, dtype=dtype, seed=3, scale=1.5)              # fused [v | k] projection of the key side
    alpha = dk ** -0.5
    # ---- reference (fp32, CPU)
    keepf = torch.ones(B, H, tq, tk)
    if p > 0:
        rng = P.Rng(5, "cpu")
        buf = torch.zeros(B, NL, H, tq, tkp, dtype=dtype)
        bufd = torch.zeros_like(buf)
        P.softmax_rect_fwd(buf[:, 1], bufd[:, 1], torch.ones(B, tk, dtype=torch.bool), tk, False, p, rng, 11)
        thr = int(p * 65536.0 + 0.5)
        keepf = (bufd[:, 1, ..., :tk].float() != 0).float() * (65536.0 / (65536.0 - thr))
    qf, kf, vf = (x.float().requires_grad_(True) for x in (q.permute(0, 2, 1, 3), vk[:, :, 1].permute(0, 2, 1, 3), vk[:, :, 0].permute(0, 2, 1, 3)))
    S = torch.einsum("bhqd,bhkd->bhqk", qf, kf) * alpha
    vis = km[:, None, None, :].expand(B, H, tq, tk).clone()
    if causal:
        vis &= torch.tril(torch.ones(tq, tk, dtype=torch.bool))
    Sm = S.masked_fill(~vis, -1e4)
    Pm = torch.softmax(Sm, -1)
    O_ref = torch.einsum("bhqk,bhkd->bhqd", Pm * keepf, vf)
    O_ref.backward(dO.permute(0, 2, 1, 3).float())
    # ---- kernels
    qc, vkc, g = q.cuda(), vk.cuda(), dO.cuda()
    q4, v4, k4 = qc.permute(0, 2, 1, 3), vkc[:, :, 0].permute(0, 2, 1, 3), vkc[:, :, 1].permute(0, 2, 1, 3)
    O = torch.full((B, tq, H, dk), float("nan"), dtype=dtype, device="cuda")
    stats = torch.full((B, H, tq, 2), float("nan"), device="cuda")
    keep = torch.empty(ops.flash_attn_keep_words_rect(B, H, tq, tk), dtype=torch.int16, device="cuda") if p > 0 else None
    rngc = ops.Rng(5, "cuda")
    p_batch = NL * H * tq * tkp
    kinfo = ops.flash_mask_info(km.cuda()) if p > 0 else None
    ops.flash_attention_fwd(q4, k4, v4, km.cuda(), O.permute(0, 2, 1, 3), stats, keep, alpha, p_batch, p, rngc, 11, causal=causal, key_info=kinfo)
    got = O.permute(0, 2, 1, 3).float().cpu()
    assert torch.isfinite(got).all()
    err = float((got - O_ref.detach()).abs().max() / O_ref.detach().abs().amax().clamp_min(1e-2))
    assert err < 3e-2, ("O", err)
    close(stats[..., 0].cpu(), Sm.detach().amax(-1), "row maximum", rtol=1e-2, atol=3e-2)
    close(stats[..., 1].cpu(), torch.exp(Sm.detach() - Sm.detach().amax(-1, keepdim=True)).sum(-1), "sum of exponentials", rtol=2e-2, atol=1e-2)
    dq = torch.full((B, tq, H, dk), float("nan"), dtype=dtype, device="cuda")
    dvk = torch.full((B, tk, 2, H, dk), float("nan"), dtype=dtype, device="cuda")
    aux = torch.empty((B, H, tq, 4), device="cuda")
    dbias = [torch.full((H * dk,), 0.5, device="cuda") for _ in range(3)]
    ops.flash_attention_bwd(q4, k4, v4, km.cuda(), O.permute(0, 2, 1, 3), g.permute(0, 2, 1, 3), stats, keep, aux, dq.permute(0, 2, 1, 3),
                            dvk[:, :, 1].permute(0, 2, 1, 3), dvk[:, :, 0].permute(0, 2, 1, 3), alpha, p, causal=causal, dbias=dbias, key_info=kinfo)
    for name, a, r in (("dQ", dq.permute(0, 2, 1, 3), qf.grad), ("dK", dvk[:, :, 1].permute(0, 2, 1, 3), kf.grad), ("dV", dvk[:, :, 0].permute(0, 2, 1, 3), vf.grad)):
        a = a.float().cpu()
        assert torch.isfinite(a).all(), name
        # (one key: dS = 0 exactly; what is left is delta = rowsum(dO * bf16(O)) against the fp32 product, ~2^-9 |dO| |V| sqrt(dk))
        scale = r.abs().amax().clamp_min(1.0 if tk == 1 else 0.05)
        err = float((a - r).abs().max() / scale)
        assert err < (1e-1 if tk == 1 else 3e-2), (name, err)
    for x, rows, n in zip(dbias, (dq, dvk[:, :, 1], dvk[:, :, 0]), ("dbias_q", "dbias_k", "dbias_v")):
        rows = rows.float()
        noise = 4.0 * (B * max(tq, tk)) ** 0.5 * 2.0 ** -9 * float(rows.abs().max())
        close(x.cpu(), rows.sum((0, 1)).reshape(-1).cpu() + 0.5, n, rtol=2e-2, atol=2e-2 + noise)


@pytest.mark.parametrize("dtype", DT)
def test_pe_embedding_linear1(ops, dtype):
    B, t, d, V = 3, 29, 64, 40
    pe, alpha = rnd(100, d, seed=1), torch.tensor([1.3])
    a, dout = rnd(B, t, d, dtype=dtype, seed=2), rnd(B, t, d, seed=3)
    ids = torch.from_numpy(np.random.default_rng(0).integers(0, V, size=(B, t)))
    table = rnd(V, d, seed=4)
    w, bb = rnd(d, seed=5), torch.tensor([0.37])
    mask = torch.from_numpy(np.random.default_rng(1).random((B, t)) > 0.3)
    dvec = rnd(B, t, seed=6)
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda x: x.cuda()) if dev == "cuda" else (lambda x: x.clone())
        rng = o.Rng(3, dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        r = []
        for p in (0.0, 0.25):
            x = o.pe_add_fwd(mv(a), mv(pe), mv(alpha), p, rng, 21)
            dal = z(1)
            da = o.pe_add_bwd(mv(dout), mv(pe), dtype, dal, p, rng, 21)
            r += [x, da, dal]
        e = o.embedding_fwd(mv(ids), mv(table), torch.float32)
        dt_ = z(V, d)
        o.embedding_bwd(mv(ids), mv(dout), dt_, padding_idx=0)
        r += [e, dt_]
        y = o.linear1_fwd(mv(a), mv(w), mv(bb), mv(mask))
        dw, db = z(d), z(1)
        dx = o.linear1_bwd(mv(dvec), mv(a), mv(w), mv(mask), dw, db)
        r += [y, dx, dw, db]
        out[dev] = r
    for i, (a_, b_) in enumerate(zip(out["cuda"], out["cpu"])):
        close(a_, b_, f"pe/embedding/linear1 output #{i}", **tol(dtype, k=20))
    assert float(out["cuda"][7][0].abs().sum()) == 0.0, "padding_idx row receives no gradient"


@pytest.mark.parametrize("dtype", DT)
def test_length_regulator_and_bucket_embed(ops, dtype):
    B, L, d, T = 4, 13, 32, 40
    g = np.random.default_rng(3)
    dur = torch.from_numpy(g.integers(0, 6, size=(B, L)))
    dur[0] = 0
    dur[0, 4] = 3                     # mostly zero durations
    dur[1] = 5                        # sum 65 > T: cropped
    dur[2, 0] = -2                    # negative duration is clamped
    x = rnd(B, L, d, dtype=dtype, seed=1)
    dout = rnd(B, T, d, dtype=dtype, seed=2)
    f0 = torch.from_numpy(g.uniform(0, 900, size=(B, T)).astype(np.float32))
    en = torch.from_numpy(g.uniform(-5, 420, size=(B, T)).astype(np.float32))
    pb = torch.exp(torch.linspace(np.log(71.0), np.log(799.8), 255))
    eb = torch.linspace(0.0, 403.8, 255)
    f0[0, :5] = pb[[0, 10, 100, 254, 77]]      # exactly on a boundary
    f0[0, 5:8] = torch.tensor([0.0, 71.0, 1e6])
    en[0, :3] = eb[[0, 128, 254]]
    Ep, Ee = rnd(256, d, seed=3), rnd(256, d, seed=4)
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        y, starts = o.length_regulate_fwd(mv(x), mv(dur), T)
        dx = o.length_regulate_bwd(mv(dout), starts, L)
        dx2 = o.length_regulate_bwd(mv(dout), starts, L, dx=mv(x))
        v, idx = o.bucket_embed_add_fwd(mv(dout), mv(f0), mv(en), mv(pb), mv(eb), mv(Ep), mv(Ee))
        dEp = torch.zeros(256, d, device=dev)
        dEe = torch.zeros(256, d, device=dev)
        o.bucket_embed_bwd(mv(dout), idx, dEp, dEe)
        out[dev] = (y, starts, dx, dx2, v, idx, dEp, dEe)
    a, b = out["cuda"], out["cpu"]
    assert torch.equal(a[0].cpu(), b[0]), "length regulator output is a pure copy: bit-exact"
    assert torch.equal(a[1].cpu(), b[1]) and torch.equal(a[5].cpu(), b[5]), "scan / bucket indices bit-exact"
    for i in (2, 3, 4, 6, 7):
        close(a[i], b[i], f"LR/bucket output #{i}", **tol(dtype, k=4))
    # one term only (hp.pitch_pred / hp.energy_pred False): the missing term's pointers are null, its idx row holds -1
    for which in ("pitch", "energy"):
        single = {}
        for o, dev in ((ops, "cuda"), (P, "cpu")):
            mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
            args = (mv(f0), None, mv(pb), None, mv(Ep), None) if which == "pitch" else (None, mv(en), None, mv(eb), None, mv(Ee))
            v, idx = o.bucket_embed_add_fwd(mv(dout), *args)
            dE = torch.zeros(256, d, device=dev)
            o.bucket_embed_bwd(mv(dout), idx, dE if which == "pitch" else None, None if which == "pitch" else dE)
            single[dev] = (v, idx, dE)
        assert torch.equal(single["cuda"][1].cpu(), single["cpu"][1]), which
        assert bool((single["cpu"][1][1 if which == "pitch" else 0] == -1).all())
        close(single["cuda"][0], single["cpu"][0], f"bucket embed, {which} term only", **tol(dtype, k=4))
        close(single["cuda"][2], single["cpu"][2], f"bucket embed gradient, {which} term only", **tol(dtype, k=4))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("C", [256, 512, 80])
def test_batchnorm_tanh(ops, dtype, p, C):
    M = 333
    x, dy = rnd(M, C, dtype=dtype, seed=1, scale=2.0) + 0.3, rnd(M, C, dtype=dtype, seed=2)
    gm, bt = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        rng = o.Rng(8, dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        sums = z(2 * C)
        o.colstats(mv(x), sums)
        rm, rv, nbt = z(C) + 0.1, z(C) + 1.0, torch.zeros((), dtype=torch.int64, device=dev)
        mean, rstd = o.bn_finalize(sums, M, 1e-5, 0.1, rm, rv, nbt)
        y = o.bn_tanh_fwd(mv(x), mean, rstd, mv(gm), mv(bt), p, rng, 31)
        red = z(2 * C)
        o.bn_tanh_bwd_reduce(mv(dy), mv(x), mean, rstd, mv(gm), mv(bt), red, p, rng, 31)
        cd = torch.tensor([float(M)], device=dev)
        dgm, dbt = z(C) + 0.25, z(C) - 0.5          # the apply kernel adds the affine gradients (the reduced sums) itself
        dx = o.bn_tanh_bwd_apply(mv(dy), mv(x), mean, rstd, mv(gm), mv(bt), red, 1.0, dgm, dbt, p, rng, 31, count_dev=cd)
        out[dev] = (sums, mean, rstd, rm, rv, nbt.float(), y, red, dx, dgm, dbt)
        # the two forward steps as ONE launch (fs2_bn_stats_tanh_fwd): same statistics, running statistics and output as the pair
        rm2, rv2, nbt2 = z(C) + 0.1, z(C) + 1.0, torch.zeros((), dtype=torch.int64, device=dev)
        y2, mean2, rstd2 = o.bn_stats_tanh_fwd(mv(x), sums, 0.0 if dev == "cuda" else M, 1e-5, 0.1, rm2, rv2, nbt2, mv(gm), mv(bt), p, rng, 31, count_dev=cd)
        assert int(nbt2) == 1
        for nm, u, v in (("mean", mean2, mean), ("rstd", rstd2, rstd), ("running_mean", rm2, rm), ("running_var", rv2, rv)):
            close(u, v, f"fused bn {nm} ({dev})", rtol=1e-6, atol=1e-7)
        close(y2, y, f"fused bn output ({dev})", rtol=1e-5 if dtype == torch.float32 else 1e-2, atol=1e-6 if dtype == torch.float32 else 1e-2)
    for i, (a_, b_) in enumerate(zip(out["cuda"], out["cpu"])):
        close(a_, b_, f"bn output #{i}", rtol=2e-3 if i in (0, 7, 9, 10) else tol(dtype)["rtol"],
              atol=(5e-2 if i in (0, 7, 9, 10) else tol(dtype, k=2)["atol"]))
    # against torch's own BatchNorm1d in training mode (fp32)
    if dtype == torch.float32 and p == 0.0:
        bn = torch.nn.BatchNorm1d(C)
        with torch.no_grad():
            bn.weight.copy_(gm); bn.bias.copy_(bt); bn.running_mean.fill_(0.1); bn.running_var.fill_(1.0)
        ref = torch.tanh(bn(x.t().unsqueeze(0))).squeeze(0).t()
        close(out["cuda"][6], ref, "vs nn.BatchNorm1d", rtol=1e-4, atol=1e-5)
        close(out["cuda"][3], bn.running_mean, "running_mean", rtol=1e-5, atol=1e-6)
        close(out["cuda"][4], bn.running_var, "running_var", rtol=1e-4, atol=1e-6)


def test_l1_cast_permute_optimizer(ops):
    n = 5003
    pred, tgt = rnd(n, seed=1), rnd(n, seed=2)
    pred[:7] = tgt[:7]                                   # sign(0) = 0
    itgt = torch.from_numpy(np.random.default_rng(0).integers(0, 9, size=n))
    gs = torch.tensor([0.7])
    res = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        l1, l2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        o.l1_fwd(mv(pred), mv(tgt), l1)
        o.l1_fwd(mv(pred), mv(itgt), l2, True)
        d1 = o.l1_bwd(mv(pred), mv(tgt), mv(gs), torch.float32)
        d2 = o.l1_bwd(mv(pred), mv(itgt), mv(gs), torch.float32, True)
        w = mv(rnd(24, 16, 5, seed=3))
        sf, sd = torch.zeros(24, 80, device=dev), torch.zeros(16, 120, device=dev)
        o.cast_permute(w, sf, 0)
        o.cast_permute(w, sd, 1)
        gsq = torch.zeros(1, device=dev)
        g = mv(rnd(n, seed=4))
        o.sqnorm(g, gsq)
        p_, m_, v_ = mv(rnd(n, seed=5)), mv(0.1 * rnd(n, seed=6)), mv(0.01 * rnd(n, seed=7).abs())
        hyper = torch.tensor([3e-3, 1 - 0.9 ** 3, 1 - 0.98 ** 3, 1.0], device=dev)
        o.adam_step(p_, g, m_, v_, hyper, gsq, 0.9, 0.98, 1e-9, 1.0)
        res[dev] = (l1, l2, d1, d2, sf, sd, gsq, p_, m_, v_)
    for i, (a, b) in enumerate(zip(res["cuda"], res["cpu"])):
        close(a, b, f"l1/cast/adam output #{i}", rtol=2e-5, atol=1e-6)
    # the optimizer against torch.optim.Adam + clip_grad_norm_ (3rd step of a state with history)
    assert float(res["cuda"][2][:7].abs().sum()) == 0.0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("d", [64, 256, 512])
def test_ffn_tail_fused_equals_two_kernels(ops, dtype, p, d):
    """fs2_ffn_tail_fwd / _bwd (FeedForward LayerNorm + residual + next LayerNorm in one row pass) against fs2_ffn_ln_* followed by
    fs2_add_ln_*: the same values up to the last place (the intermediate is rounded where the two-kernel form stores it), and
    against the oracle's composition within the row-kernel tolerance"""
    M = 203
    f2, h, r = rnd(M, d, dtype=dtype, seed=1), rnd(M, d, dtype=dtype, seed=2), rnd(M, d, seed=3)
    g1, b1, g2, b2 = 1 + 0.1 * rnd(d, seed=4), 0.1 * rnd(d, seed=5), 1 + 0.1 * rnd(d, seed=6), 0.1 * rnd(d, seed=7)
    dy, dsd = rnd(M, d, dtype=dtype, seed=8), rnd(M, d, seed=9)
    res = {}
    for name, o, dev in (("fused", ops, "cuda"), ("two", ops, "cuda"), ("oracle", P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        rng = o.Rng(8, dev)
        z = lambda: torch.zeros(d, device=dev)
        dg2, db2, dg1, db1, cs = z(), z(), z(), z(), z()
        if name == "two":
            yff, m1, r1 = o.ffn_ln_fwd(mv(f2), mv(h), mv(g1), mv(b1), 1e-5, p, rng, 21)
            s_, y, m2, r2 = o.add_ln_fwd(mv(r), yff, mv(g2), mv(b2), 1e-5, p, rng, 22)
            dr, da = o.add_ln_bwd(mv(dsd), mv(dy), s_, mv(g2), m2, r2, dg2, db2, p, rng, 22)
            g = o.ffn_ln_bwd(da, mv(f2), mv(h), mv(g1), m1, r1, dg1, db1, p, rng, 21, dcolsum=cs)
        else:
            s_, y, m1, r1, m2, r2 = o.ffn_tail_fwd(mv(f2), mv(h), mv(r), mv(g1), mv(b1), mv(g2), mv(b2), 1e-5, p, rng, 21, 22)
            dr, g = o.ffn_tail_bwd(mv(dsd), mv(dy), s_, mv(g2), m2, r2, mv(f2), mv(h), mv(g1), m1, r1, dg2, db2, dg1, db1, p, rng, 21, 22,
                                   dcolsum=cs)
        res[name] = [s_, y, m1, r1, m2, r2, dr, g, dg2, db2, dg1, db1, cs]
    names = ["s", "y", "mean1", "rstd1", "mean2", "rstd2", "dr", "g", "dgamma2", "dbeta2", "dgamma1", "dbeta1", "dcolsum"]
    for n, a, b in zip(names[:8], res["fused"], res["two"]):      # same arithmetic; hipcc contracts multiply-adds differently in the
        if a.dtype == torch.float32:                               # two contexts, so fp32 results may differ in the last place
            close(a, b, f"{n} vs the two-kernel form", rtol=2e-6, atol=2e-6)
        else:
            bad = (a.float() - b.float()).abs() > 2.0 ** -7 * b.float().abs().clamp_min(2.0 ** -6)       # at most one bf16 ulp
            assert not bool(bad.any()), f"{n}: fused differs from the two-kernel form by {(a.float() - b.float()).abs().max().item()}"
            assert float((a != b).float().mean()) < 2e-3, f"{n}: too many last-place differences"
    for n, a, b in zip(names[8:], res["fused"][8:], res["two"][8:]):      # per-channel sums: float atomics, order varies
        close(a, b, n, rtol=1e-4, atol=1e-4)
    for n, a, b in zip(names, res["fused"], res["oracle"]):
        k = 30 if a.dim() == 1 and a.numel() == d else 2
        close(a, b, f"{n} vs oracle", **tol(a.dtype if a.dtype != torch.float32 else dtype, k=k))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("d", [64, 256])
def test_ln_linear1_fused_equals_two_kernels(ops, dtype, p, d):
    """fs2_ln_linear1_fwd / _bwd (the tail of a VariancePredictor in one row pass, normalised rows recomputed in the backward)
    against fs2_layernorm_* + fs2_linear1_*, and against the oracle's composition"""
    M = 203
    x = torch.relu(rnd(M, d, dtype=dtype, seed=1))
    gm, bt, w, b = 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3), 0.3 * rnd(d, seed=4), torch.tensor([0.2])
    mask = (torch.arange(M) % 7 != 3)
    dout = rnd(M, seed=5)
    res = {}
    for name, o, dev in (("fused", ops, "cuda"), ("two", ops, "cuda"), ("oracle", P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        rng = o.Rng(8, dev)
        z = lambda *sh: torch.zeros(*sh, device=dev)
        dg, db_, dw, dbias, cs = z(d), z(d), z(d), z(1), z(d)
        if name == "two":
            n, mean, rstd = o.layernorm_fwd(mv(x), mv(gm), mv(bt), dtype, 1e-5, p, rng, 9)
            out = o.linear1_fwd(n, mv(w), mv(b), mv(mask))
            dn = o.linear1_bwd(mv(dout), n, mv(w), mv(mask), dw, dbias)
            dx = o.layernorm_bwd(dn, mv(x), mv(gm), mean, rstd, dg, db_, p, rng, 9, relu_mask=True, dcolsum=cs)
        else:
            out, mean, rstd = o.ln_linear1_fwd(mv(x), mv(gm), mv(bt), mv(w), mv(b), mv(mask), 1e-5, p, rng, 9)
            dx = o.ln_linear1_bwd(mv(dout), mv(x), mv(gm), mv(bt), mean, rstd, mv(w), mv(mask), dg, db_, dw, dbias, p, rng, 9,
                                  relu_mask=True, dcolsum=cs)
        res[name] = [out, mean, rstd, dx, dg, db_, dw, dbias, cs]
    names = ["out", "mean", "rstd", "dx", "dgamma", "dbeta", "dw", "db", "dcolsum"]
    for other in ("two", "oracle"):
        for n, a, b_ in zip(names, res["fused"], res[other]):
            k = 30 if n in ("dgamma", "dbeta", "dw", "db", "dcolsum") else 2
            close(a, b_, f"{n} vs {other}", **tol(a.dtype if a.dtype != torch.float32 else dtype, k=k))
    assert torch.all(res["fused"][0].cpu()[~mask] == 0)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("emb", [False, True])
@pytest.mark.parametrize("B,t,d", [(3, 37, 256), (2, 5, 32), (4, 130, 512)])
def test_stack_head_one_pass_each_way(ops, dtype, p, emb, B, t, d):
    """embedding gather / input activations + positional encoder + first LayerNorm in one row pass, and its backward (LayerNorm backward
    + residual gradient + positional encoder backward), against the composition of the single-step oracle primitives"""
    V = 41
    table, a = rnd(V, d, seed=1), rnd(B, t, d, dtype=dtype, seed=2)
    ids = torch.from_numpy(np.random.default_rng(3).integers(0, V, size=(B, t)))
    pe, alpha = rnd(200, d, seed=4), torch.tensor([0.7])
    gm, bt = 1 + 0.1 * rnd(d, seed=5), 0.1 * rnd(d, seed=6)
    dy, ds = rnd(B, t, d, dtype=dtype, seed=7), rnd(B, t, d, seed=8)
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda x_: x_.cuda()) if dev == "cuda" else (lambda x_: x_.clone())
        rng = o.Rng(11, dev)
        z = lambda *s_: torch.zeros(*s_, dtype=torch.float32, device=dev)
        if emb:
            x, y, mean, rstd = o.pe_add_ln_fwd(mv(table), mv(pe), mv(alpha), mv(gm), mv(bt), dtype, p, rng, 17, ids=mv(ids))
        else:
            x, y, mean, rstd = o.pe_add_ln_fwd(mv(a), mv(pe), mv(alpha), mv(gm), mv(bt), dtype, p, rng, 17)
        dg, db, dal, dcs = z(d) + 0.5, z(d) - 0.25, z(1) + 2.0, z(d) + 1.0
        da = o.ln_pe_add_bwd(mv(dy), x, mv(gm), mean, rstd, mv(ds), mv(pe), torch.float32 if emb else dtype, dg, db, dal, p, rng, 17,
                             dcolsum=None if emb else dcs)
        da_nods = o.ln_pe_add_bwd(mv(dy), x, mv(gm), mean, rstd, None, mv(pe), dtype, z(d), z(d), z(1), p, rng, 17)
        out[dev] = (x, y, mean, rstd, da, dg, db, dal, dcs, da_nods)
    names = ("x", "y", "mean", "rstd", "da", "dgamma", "dbeta", "dalpha", "dcolsum", "da without ds")
    for i, (u, v) in enumerate(zip(out["cuda"], out["cpu"])):
        red = i in (5, 6, 7, 8)
        close(u, v, f"stack head {names[i]}", rtol=2e-3 if red else tol(u.dtype)["rtol"], atol=(2e-2 * (B * t) ** 0.5 if red else tol(u.dtype, k=2)["atol"]))
    assert out["cuda"][0].dtype == torch.float32 and out["cuda"][1].dtype == dtype


@pytest.mark.parametrize("B,L,T,d", [(48, 128, 925, 256), (3, 1, 7, 8), (2, 2100, 300, 16), (5, 70, 33, 64), (2, 64, 32, 256)])
def test_length_regulator_one_launch_and_two_launch_forms(ops, B, L, T, d):
    """the scan + gather launch (L <= 2048: the utterance's prefix sums live in LDS) and the scan / gather pair behind it: frames, padding
    beyond the utterance, cropping at T and the prefix sums kept for the backward pass, bit for bit against the oracle"""
    g = np.random.default_rng(B * L + T)
    dur = torch.from_numpy(g.integers(0, max(2, 2 * T // L + 2), size=(B, L)))
    dur[0] = 0
    dur[0, L // 2] = 3
    x = rnd(B, L, d, dtype=torch.bfloat16, seed=1)
    y, starts = ops.length_regulate_fwd(x.cuda(), dur.cuda(), T)
    yo, so = P.length_regulate_fwd(x, dur, T)
    assert torch.equal(starts.cpu(), so) and torch.equal(y.cpu(), yo)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,ld", [(1, 4, 4), (7, 80, 80), (44400, 80, 80), (1001, 128, 160), (999, 20, 24), (3000, 132, 132), (513, 64, 64)])
def test_colsum_narrow_and_wide(ops, dtype, M, N, ld):
    """column sums of narrow matrices (N <= 128: several rows per wave; the 80 mel channels of the bias gradients) and of wider
    ones, contiguous and as a column slice of a wider tensor, accumulated onto a non-zero vector"""
    x = rnd(M, ld, dtype=dtype, seed=M + N)
    init = rnd(N, seed=5)
    a = ops.colsum(x.cuda()[:, :N], init.clone().cuda())
    b = P.colsum(x[:, :N], init.clone())
    close(a, b, f"colsum {M}x{N}", rtol=1e-4, atol=2e-4 * M ** 0.5)


def test_copy_batched(ops):
    """several device-to-device copies in one launch: mixed dtypes and sizes, unaligned views, empty tensors, more than 8 pairs"""
    g = np.random.default_rng(0)
    srcs = [torch.from_numpy(g.integers(0, 100, size=n)).cuda() for n in (1, 7, 4096, 48 * 925)] + \
           [rnd(48 * 925 * 80, seed=1).cuda(), rnd(33, seed=2).cuda()[1:], torch.zeros(0).cuda(), rnd(5, 7, dtype=torch.bfloat16, seed=3).cuda()] + \
           [rnd(1000 + i, seed=10 + i).cuda() for i in range(3)]
    dsts = [torch.empty_like(s_) if s_.is_contiguous() else torch.empty(s_.shape, dtype=s_.dtype, device="cuda") for s_ in srcs]
    ops.copy_batched(dsts, [s_.contiguous() if not s_.is_contiguous() else s_ for s_ in srcs])
    for d_, s_ in zip(dsts, srcs):
        assert torch.equal(d_, s_)


@pytest.mark.parametrize("n", [1, 3, 4, 1027, 44400 * 80])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_add_cast(ops, n, dtype):
    """(a + b).to(dtype) in one pass (the post-net's two gradient terms of mel_pred), tails that are not a multiple of four"""
    a, b = rnd(n, seed=1), rnd(n, seed=2)
    out = ops.add_cast(a.cuda(), b.cuda(), dtype)
    assert out.dtype == dtype and torch.equal(out.cpu(), P.add_cast(a, b, dtype))


def test_l1_multi(ops):
    """the trainer's L1 terms in one launch each way against the oracle (fp32 and bf16 predictions, an int64 log1p target, odd
    sizes, unaligned tails) and against nn.L1Loss"""
    shapes = [(3, 37, 80), (3, 37, 80), (3, 11), (3, 37), (1, 5)]
    preds = [rnd(*sh, seed=i) for i, sh in enumerate(shapes)]
    preds[3] = preds[3].to(torch.bfloat16)
    targets = [rnd(*sh, seed=10 + i) for i, sh in enumerate(shapes)]
    targets[2] = torch.from_numpy(np.random.default_rng(0).integers(0, 9, size=shapes[2]))
    modes = (False, False, True, False, False)
    res = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        pr, tg = [mv(t) for t in preds], [mv(t) for t in targets]
        losses = torch.full((len(pr) + 1,), 7.0, device=dev)       # the terms and, behind them, their sum: stored over whatever was there
        o.l1_multi_fwd(pr, tg, modes, losses)
        if dev == "cuda":           # no float atomics: the same bits on every launch (and the ticket word is back at zero)
            again = torch.empty_like(losses)
            o.l1_multi_fwd(pr, tg, modes, again)
            assert torch.equal(again, losses)
        d = o.l1_multi_bwd(pr, tg, modes, torch.tensor([0.7], device=dev), [torch.float32, torch.bfloat16, torch.float32, torch.bfloat16, torch.float32])
        res[dev] = [losses] + d
    for i, (a, b) in enumerate(zip(res["cuda"], res["cpu"])):
        close(a, b, f"l1_multi output #{i}", rtol=1e-5 if a.dtype == torch.float32 else 1e-2, atol=1e-6)
    ref = torch.nn.L1Loss()(preds[0], targets[0])
    close(res["cuda"][0][0], ref, "vs nn.L1Loss", rtol=1e-5, atol=1e-6)
    close(res["cuda"][0][len(shapes)], res["cuda"][0][:len(shapes)].sum(), "sum slot", rtol=1e-5, atol=1e-6)


def test_adam_matches_torch_optimizer(ops):
    n = 4099
    p0, g1, g2 = rnd(n, seed=1), 3 * rnd(n, seed=2), 0.01 * rnd(n, seed=3)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    p, m, v = p0.cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step, (g, lr) in enumerate(((g1, 2e-3), (g2, 1e-3)), start=1):
        for grp in opt.param_groups:
            grp["lr"] = lr
        ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gsq = torch.zeros(1).cuda()
        ops.sqnorm(g.cuda(), gsq)
        hyper = torch.tensor([lr, 1 - 0.9 ** step, 1 - 0.98 ** step, 1.0]).cuda()
        ops.adam_step(p, g.cuda(), m, v, hyper, gsq, 0.9, 0.98, 1e-9, 1.0)
        close(p, ref.data, f"Adam step {step}", rtol=1e-5, atol=1e-7)


def test_adam_with_kernel_layout_conv_gradients(ops):
    """fs2_adam_step_perm: Conv1d-weight ranges of the arena keep their gradient as [o][j][i] (what the weight-gradient GEMM
    writes); the update must equal plain Adam on the (O,I,k) gradient, bit for bit, whatever the segment sizes (odd O*I*k, a
    float4 group running across an o boundary, padding after a segment) -- and optim.ParamArena must expose param.grad in the
    reference's layout"""
    shapes = [(7,), (5, 3, 9), (16, 8), (3, 7, 5), (2, 2, 3), (64, 32, 9), (11,)]
    off, offs = 0, []
    for sh in shapes:
        offs.append(off)
        off += (int(np.prod(sh)) + 3) // 4 * 4
    n = off
    p0, m0, v0 = rnd(n, seed=1), 0.1 * rnd(n, seed=2), (0.1 * rnd(n, seed=3)) ** 2
    g_ref = torch.zeros(n)
    g_arena = torch.zeros(n)
    segs = []
    for sh, o in zip(shapes, offs):
        gt = rnd(*sh, seed=10 + o)
        num = gt.numel()
        g_ref[o:o + num] = gt.reshape(-1)
        if len(sh) == 3:
            g_arena[o:o + num] = gt.permute(0, 2, 1).reshape(-1)        # [o][j][i]
            segs.append([o, o + num, sh[0], sh[1], sh[2]])
        else:
            g_arena[o:o + num] = gt.reshape(-1)
    perm = torch.tensor(segs, dtype=torch.int64)
    hyper = torch.tensor([1e-3, 1 - 0.9, 1 - 0.98, 1.0])
    res = []
    # ONE squared norm for both runs: the sum's rounding depends on the element order (and on the arrival order of the blocks'
    # atomics), and the clip factor derived from it scales every element -- the comparison below is about the gather alone
    gsq = torch.zeros(1).cuda()
    ops.sqnorm(g_ref.cuda(), gsq)
    for g, pm in ((g_ref, None), (g_arena, perm)):
        p, m, v = p0.cuda(), m0.cuda(), v0.cuda()
        ops.adam_step(p, g.cuda(), m, v, hyper.cuda(), gsq, 0.9, 0.98, 1e-9, 1.0, perm=pm.cuda() if pm is not None else None)
        res.append((p.cpu(), m.cpu(), v.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    po, mo, vo = p0.clone(), m0.clone(), v0.clone()
    P.adam_step(po, g_arena, mo, vo, hyper, torch.tensor([float((g_ref ** 2).sum())]), 0.9, 0.98, 1e-9, 1.0, perm=perm)
    close(res[1][0], po, "oracle adam with permuted segments", rtol=1e-4, atol=1e-6)
    # the arena's views
    from transformer_tts_amd.optim import ParamArena
    conv = torch.nn.Conv1d(6, 10, 5).cuda()
    arena = ParamArena(list(conv.parameters()))
    w = conv.weight
    assert w.grad.shape == w.shape and arena.perm.tolist() == [[0, 300, 10, 6, 5]]
    w._fs2_grad_raw.copy_(torch.arange(300, dtype=torch.float32).view(10, 30))
    assert torch.equal(w.grad.cpu(), torch.arange(300, dtype=torch.float32).view(10, 5, 6).permute(0, 2, 1))


@pytest.mark.parametrize("dtype", DT)
def test_fused_bias_gradients_onehot_and_batched_shadows(ops, dtype):
    """dcolsum outputs of the backward row kernels, the sums-only GEMM epilogue, the one-hot embedding-gradient
    operand and the one-launch weight-shadow refresh"""
    M, d = 203, 64
    x32, a, h = rnd(M, d, seed=1), rnd(M, d, dtype=dtype, seed=3), rnd(M, d, dtype=dtype, seed=4)
    gm, bt = 1 + 0.1 * rnd(d, seed=5), 0.1 * rnd(d, seed=6)
    dy, dsd = rnd(M, d, dtype=dtype, seed=7), rnd(M, d, seed=8)
    w = rnd(48, d, dtype=dtype, seed=9, scale=0.2)
    idx = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=M).astype(np.int32))
    idx[:60] = 0
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        rng = o.Rng(99, dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        r = []
        p = 0.25
        y, mu, rs = o.layernorm_fwd(mv(a), mv(gm), mv(bt), dtype, 1e-5, p, rng, 5)
        cs = z(d)
        dx = o.layernorm_bwd(mv(dy), mv(a), mv(gm), mu, rs, z(d), z(d), p, rng, 5, relu_mask=True, dcolsum=cs)
        r += [cs, dx]
        s, y, mu, rs = o.add_ln_fwd(mv(x32), mv(a), mv(gm), mv(bt), 1e-5, p, rng, 7)
        cs = z(d)
        dr, da = o.add_ln_bwd(mv(dsd), mv(dy), s, mv(gm), mu, rs, z(d), z(d), p, rng, 7, dcolsum=cs)
        r += [cs, da]
        y, mu, rs = o.ffn_ln_fwd(mv(a), mv(h), mv(gm), mv(bt), 1e-5, p, rng, 8)
        cs = z(d)
        g = o.ffn_ln_bwd(mv(dy), mv(a), mv(h), mv(gm), mu, rs, z(d), z(d), p, rng, 8, dcolsum=cs)
        r += [cs, g]
        cs, dal = z(d), z(1)
        da0 = o.pe_add_bwd(mv(dsd).view(7, 29, d), mv(rnd(64, d, seed=11)), dtype, dal, p, rng, 9, dcolsum=cs)
        r += [cs, da0]
        cs = z(48)
        yy = o.linear(mv(a), mv(w), relu_mask=mv(rnd(M, 48, dtype=dtype, seed=12)), colsum=cs)
        r += [cs, yy]
        oh = o.onehot(mv(idx), 256, dtype)
        dE = z(256, d)
        o.wgrad(oh, mv(dy), dE)
        r += [oh, dE]
        # batched shadows
        w3, w2, b1 = mv(rnd(24, 16, 5, seed=13)), mv(rnd(32, 16, seed=14)), mv(rnd(32, seed=15))
        f3, d3 = torch.zeros(24, 80, dtype=dtype, device=dev), torch.zeros(16, 120, dtype=dtype, device=dev)
        fused, bias = torch.zeros(64, 16, dtype=dtype, device=dev), torch.zeros(64, device=dev)
        table = o.make_cast_table([(w3, f3, 0), (w3, d3, 1), (w2, fused[32:64], 0), (b1, bias[32:64], 2)], dev)
        o.cast_permute_batched(table, 4, dtype)
        r += [f3, d3, fused, bias]
        out[dev] = r
    for i, (a_, b_) in enumerate(zip(out["cuda"], out["cpu"])):
        k = 30 if a_.dim() == 1 else 2
        close(a_, b_, f"fused output #{i}", **tol(dtype, k=k))
    assert torch.equal(out["cuda"][10].cpu(), out["cpu"][10]), "one-hot is exact"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("O,I,k", [(70, 130, 9), (33, 65, 3), (100, 72, 1), (256, 1024, 9), (512, 80, 5)])
def test_cast_permute_batched_tile_edges(ops, dtype, O, I, k):
    """the one-launch shadow refresh at shapes that do not divide its 16 x 64 / 64 x 16 tiles, odd leading dimensions included
    (a fused q/v/k shadow writes into column / row blocks of a wider tensor): exact against the oracle's permutes"""
    w = rnd(O, I, k, seed=3) if k > 1 else rnd(O, I, seed=3)
    res = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        wd = w.to(dev)
        f = torch.zeros(O, k * I + 3, dtype=dtype, device=dev)[:, :k * I]       # row stride k*I + 3: unaligned pairs
        g = torch.zeros(I, k * O, dtype=dtype, device=dev)
        table = o.make_cast_table([(wd, f, 0), (wd, g, 1)], dev)
        o.cast_permute_batched(table, 2, dtype)
        res[dev] = (f.float().cpu(), g.float().cpu())
    assert torch.equal(res["cuda"][0], res["cpu"][0]) and torch.equal(res["cuda"][1], res["cpu"][1])
    w3 = w.reshape(O, I, k)
    assert torch.equal(res["cuda"][0], w3.permute(0, 2, 1).reshape(O, k * I).to(dtype).float())
    assert torch.equal(res["cuda"][1], w3.flip(2).permute(1, 2, 0).reshape(I, k * O).to(dtype).float())


# ------------------------------------------------------------------------------------------------ 16-wave row-major GEMM (gemm_ring.hip)
@pytest.fixture
def big_gemm(monkeypatch):
    """route every eligible product to the 256-wide 16-wave LDS-DMA kernel (FS2_GEMM_RING=2), whatever its size"""
    def set_bm(bm):
        monkeypatch.setenv("FS2_GEMM_RING", "2")
        monkeypatch.setenv("FS2_GEMM_WS", "0")              # (K = 256 shapes would otherwise go to gemm_ws.hip)
        monkeypatch.setenv("FS2_GEMM_BIG_BM", str(bm))
    yield set_bm
    monkeypatch.delenv("FS2_GEMM_RING", raising=False)
    monkeypatch.delenv("FS2_GEMM_WS", raising=False)
    monkeypatch.delenv("FS2_GEMM_BIG_BM", raising=False)
    monkeypatch.delenv("FS2_RING_S", raising=False)


@pytest.mark.parametrize("bm", [128, 192, 256])
@pytest.mark.parametrize("M,N,K", [(300, 80, 72), (129, 256, 256), (1000, 1024, 8), (577, 264, 200), (2048, 512, 64)])
def test_big_gemm_linear_epilogues(ops, big_gemm, bm, M, N, K):
    """every fused epilogue of the large-tile kernel (bias, ReLU, ReLU mask, fp32 / bf16 residual, fp32 output, column
    sums, BatchNorm statistics) on ragged shapes (M, N, K not multiples of the tile) against the oracle primitive"""
    big_gemm(bm)
    dtype = torch.bfloat16
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias, res, mask = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, dtype=dtype, seed=5)
    for kw in (dict(), dict(bias=True, relu=True), dict(residual="f32", out_f32=True), dict(residual="bf16", bias=True),
               dict(relu_mask=True), dict(relu_mask=True, colsum=True), dict(bias=True, colstats=True), dict(colsum=True),
               dict(relu_mask=True, residual="f32", out_f32=True), dict(alpha=0.25)):
        def call(o, dev):
            mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
            cs = torch.zeros(2 * N, dtype=torch.float32, device=dev) if kw.get("colstats") else None
            cl = torch.zeros(N, dtype=torch.float32, device=dev) if kw.get("colsum") else None
            r = None
            if kw.get("residual"):
                r = mv(res if kw["residual"] == "f32" else res.to(dtype))
            out = o.linear(mv(x), mv(w), bias=mv(bias) if kw.get("bias") else None, relu=kw.get("relu", False), residual=r,
                           relu_mask=mv(mask) if kw.get("relu_mask") else None, colstats=cs, colsum=cl,
                           out_dtype=torch.float32 if kw.get("out_f32") else None, alpha=kw.get("alpha", 1.0))
            return out, cs, cl
        (a, acs, acl), (b, bcs, bcl) = call(ops, "cuda"), call(P, "cpu")
        rerouted = bm == 256 and (kw.get("colstats") or (kw.get("relu_mask") and (kw.get("residual") == "f32" or kw.get("colsum"))))   # not compiled for 256 rows (spills)
        assert ops.lib().fs2_gemm_last_tile() == (130 if bm == 128 else 192 if rerouted else bm)
        close(a, b, f"big linear bm{bm} {kw}", **tol(a.dtype))
        if acs is not None:
            close(acs, bcs, "colstats", rtol=2e-3, atol=2e-2 * M ** 0.5)
        if acl is not None:
            close(acl, bcl, "colsum", rtol=2e-3, atol=2e-2 * M ** 0.5)


@pytest.mark.parametrize("bm", [128, 192, 256])
@pytest.mark.parametrize("B,t,C,N,taps,pad", [(3, 37, 32, 128, 3, 1), (2, 50, 80, 256, 5, 4), (3, 37, 64, 64, 9, 4),
                                               (2, 131, 256, 80, 5, 0), (4, 20, 32, 32, 1, 0), (5, 301, 72, 264, 9, 4)])
def test_big_gemm_conv_geometry(ops, big_gemm, bm, B, t, C, N, taps, pad):
    """implicit-GEMM Conv1d on the large-tile kernel: halo rows, sequence boundaries inside a row slab, K tails"""
    big_gemm(bm)
    dtype = torch.bfloat16
    x, w = rnd(B, t, C, dtype=dtype, seed=1), rnd(N, taps * C, dtype=dtype, seed=2, scale=(taps * C) ** -0.5)
    bias = rnd(N, seed=3)
    a = ops.conv(x.cuda(), w.cuda(), taps, pad, bias=bias.cuda(), relu=True)
    b = P.conv(x, w, taps, pad, bias=bias, relu=True)
    close(a, b, "big conv", **tol(dtype))
    res = rnd(B, t, N, seed=4)
    a = ops.conv(x.cuda(), w.cuda(), taps, pad, residual=res.cuda(), out_dtype=torch.float32)
    b = P.conv(x, w, taps, pad, residual=res, out_dtype=torch.float32)
    close(a, b, "big conv+res f32 out", **tol(dtype))


def test_big_gemm_is_bit_identical_to_the_128_tile_kernel(ops, monkeypatch):
    """same k order inside every accumulator: the two kernels must agree bit for bit (44400-row decoder product); so must the
    2- and 3-slot rings of the 128-row tile (the counted vmcnt waits are the only difference)"""
    x, w, bias = rnd(44400, 256, dtype=torch.bfloat16, seed=1).cuda(), rnd(512, 256, dtype=torch.bfloat16, seed=2).cuda(), rnd(512, seed=3).cuda()
    monkeypatch.setenv("FS2_GEMM_RING", "0")
    monkeypatch.setenv("FS2_GEMM_WS", "0")
    ref = ops.linear(x, w, bias=bias, relu=True)
    assert ops.lib().fs2_gemm_last_tile() == 128
    for bm, ring in (("128", "3"), ("128", "2"), ("192", "2"), ("256", "2")):
        monkeypatch.setenv("FS2_GEMM_RING", "2")
        monkeypatch.setenv("FS2_GEMM_BIG_BM", bm)
        monkeypatch.setenv("FS2_RING_S", ring)
        out = ops.linear(x, w, bias=bias, relu=True)
        assert ops.lib().fs2_gemm_last_tile() == (130 if bm == "128" else int(bm))
        assert torch.equal(out, ref), f"bm {bm}: {(out.float() - ref.float()).abs().max().item()}"
    monkeypatch.delenv("FS2_RING_S", raising=False)
    monkeypatch.delenv("FS2_GEMM_WS", raising=False)


@pytest.mark.parametrize("M,N", [(300, 80), (129, 256), (1000, 1024), (577, 264), (4100, 768)])
def test_weights_stationary_gemm_epilogues(ops, monkeypatch, M, N):
    """gemm_ws.hip (K = 256: the weight tile lives in registers, activations stream through an LDS ring, two row halves ping-pong
    between MFMA and epilogue): every fused epilogue it is compiled for, on ragged shapes (rows not a multiple of the 64-row slab,
    columns not a multiple of the 256-column tile), against the oracle primitive"""
    monkeypatch.setenv("FS2_GEMM_WS", "2")
    dtype, K = torch.bfloat16, 256
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=K ** -0.5)
    bias, res, mask = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, dtype=dtype, seed=5)
    for kw in (dict(), dict(bias=True, relu=True), dict(residual="f32", out_f32=True), dict(residual="bf16", bias=True), dict(relu_mask=True),
               dict(relu_mask=True, colsum=True), dict(colsum=True), dict(alpha=0.25), dict(bias=True, out_f32=True)):
        def call(o, dev):
            mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
            cl = torch.zeros(N, dtype=torch.float32, device=dev) if kw.get("colsum") else None
            r = None
            if kw.get("residual"):
                r = mv(res if kw["residual"] == "f32" else res.to(dtype))
            out = o.linear(mv(x), mv(w), bias=mv(bias) if kw.get("bias") else None, relu=kw.get("relu", False), residual=r,
                           relu_mask=mv(mask) if kw.get("relu_mask") else None, colsum=cl,
                           out_dtype=torch.float32 if kw.get("out_f32") else None, alpha=kw.get("alpha", 1.0))
            return out, cl
        (a, acl), (b, bcl) = call(ops, "cuda"), call(P, "cpu")
        assert ops.lib().fs2_gemm_last_tile() == 131, kw
        close(a, b, f"ws linear {kw}", **tol(a.dtype))
        if acl is not None:
            close(acl, bcl, "colsum", rtol=2e-3, atol=2e-2 * M ** 0.5)
    monkeypatch.delenv("FS2_GEMM_WS", raising=False)


def test_weights_stationary_gemm_is_bit_identical_to_the_tiled_kernels(ops, monkeypatch):
    """same MFMA, same k order inside every accumulator: the weights-stationary stream, the 16-wave tiled kernel and the 4-wave
    kernel agree bit for bit on a 44400-row decoder product (and the default routing picks the stream for it)"""
    x, w, bias = rnd(44400, 256, dtype=torch.bfloat16, seed=1).cuda(), rnd(768, 256, dtype=torch.bfloat16, seed=2).cuda(), rnd(768, seed=3).cuda()
    outs = {}
    for name, env in (("4-wave", ("0", "0")), ("tiled", ("0", "2")), ("stream", ("2", "0")), ("default", (None, None))):
        for k, v in zip(("FS2_GEMM_WS", "FS2_GEMM_RING"), env):
            monkeypatch.delenv(k, raising=False) if v is None else monkeypatch.setenv(k, v)
        outs[name] = (ops.linear(x, w, bias=bias, relu=True), ops.lib().fs2_gemm_last_tile())
    assert outs["4-wave"][1] == 128 and outs["tiled"][1] in (130, 192, 256) and outs["stream"][1] == 131 and outs["default"][1] == 131
    for name in ("tiled", "stream", "default"):
        assert torch.equal(outs[name][0], outs["4-wave"][0]), name


@pytest.mark.parametrize("bm", [128, 192, 256])
@pytest.mark.parametrize("B,t,C,N,taps,pad,split", [(4, 128, 256, 256, 9, 4, 5), (3, 37, 64, 64, 9, 4, 3), (2, 50, 80, 264, 5, 4, 16), (5, 100, 1024, 256, 1, 0, 4)])
def test_sliced_split_k(ops, monkeypatch, bm, B, t, C, N, taps, pad, split):
    """FS2Gemm.accumulate = 2 + fs2_splitk_reduce (gemm_ring.hip): every split stores its fp32 partial tile into its own workspace
    slice with plain stores (no atomics: the result is bitwise reproducible), the reduce pass adds bias / ReLU / residual; splits that
    do not divide the slot count, more splits asked for than slots, k tails and conv halos inside a split"""
    monkeypatch.setenv("FS2_SPLITK_N", str(split))
    monkeypatch.setenv("FS2_GEMM_BIG_BM", str(bm))
    dtype = torch.bfloat16
    x, w = rnd(B, t, C, dtype=dtype, seed=1), rnd(N, taps * C, dtype=dtype, seed=2, scale=(taps * C) ** -0.5)
    bias, res = rnd(N, seed=3), rnd(B, t, N, seed=4)
    outs = []
    for _ in range(2):
        # (_splitk_plan only splits long reductions onto few tiles: call the runner directly)
        g = ops.FS2Gemm()
        x2 = x.cuda().view(B * t, C)
        wc = w.cuda()
        g.A, g.B, g.lda, g.ldb = x2.data_ptr(), wc.data_ptr(), C, taps * C
        g.M, g.N, g.K, g.dtype = B * t, N, C, ops.BF16
        g.split_k = g.batch1 = g.batch2 = 1
        g.conv, g.taps, g.pad, g.seq_len = 1, taps, pad, t
        out = torch.empty(B * t, N, dtype=torch.float32, device="cuda")
        ops._splitk_run(g, B * t, N, split, out, bias.cuda(), True, res.cuda().view(B * t, N))
        outs.append(out.view(B, t, N))
        nslots = taps * ((C + 63) // 64)
        per = -(-nslots // min(split, nslots))
        assert ops.lib().fs2_gemm_last_splits() == -(-nslots // per)
    assert torch.equal(outs[0], outs[1]), "sliced split-K must be bitwise reproducible"
    b = P.conv(x, w, taps, pad, bias=bias, relu=True, residual=res, out_dtype=torch.float32)
    close(outs[0], b, f"sliced split-K bm{bm}", **tol(dtype))
    monkeypatch.delenv("FS2_SPLITK_N", raising=False)
    monkeypatch.delenv("FS2_GEMM_BIG_BM", raising=False)


# ------------------------------------------------------------------------------------------------ autoregressive decoder kernels
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("tq,tk,causal", [(37, 37, True), (64, 64, True), (45, 13, False), (9, 130, False), (1, 1, True)])
def test_softmax_rect_causal_and_cross(ops, dtype, p, tq, tk, causal):
    """decoder self-attention (no-peak mask of train.py:26-36 on top of the key padding mask) and encoder-decoder
    attention (tq x tk scores) through fs2_softmax_rect_fwd / _bwd, inside a (B, N, H, tq, tkp) buffer"""
    B, H, NL = 3, 2, 2
    tkp = (tk + 7) // 8 * 8
    lens = [tk, max(1, tk // 2), max(1, tk - 5)]
    km = torch.zeros(B, tk, dtype=torch.bool)
    for b, n in enumerate(lens):
        km[b, :n] = True
    out = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda x: x.cuda()) if dev == "cuda" else (lambda x: x.clone())
        rng = o.Rng(5, dev)
        buf = mv(rnd(B, NL, H, tq, tkp, dtype=dtype, seed=1, scale=3.0))
        bufd = torch.zeros_like(buf) if p > 0 else buf
        S, Pd = buf[:, 1], bufd[:, 1]
        o.softmax_rect_fwd(S, Pd, mv(km), tk, causal, p, rng, 11)
        dP = mv(rnd(B, H, tq, tkp, dtype=dtype, seed=2))
        dP[..., tk:] = float("nan")
        o.softmax_rect_bwd(dP, S, tk, p, rng, 11)
        out[dev] = (S.clone(), Pd.clone(), dP)
    for a, b, n in zip(out["cuda"], out["cpu"], ("P", "P_drop", "dS")):
        close(a, b, n, **tol(dtype))
    Pc = out["cuda"][0].float().cpu()
    assert torch.all(Pc[..., tk:] == 0), "pad columns must be written as zero"
    close(Pc[..., :tk].sum(-1), torch.ones(B, H, tq), "rows sum to one", rtol=1e-2, atol=1e-2)
    if causal and tq > 1:
        upper = torch.triu(torch.ones(tq, tk), diagonal=1).bool()
        assert float(Pc[0][..., :tk][:, upper].abs().max()) < 1e-6, "future keys get ~0 probability"


@pytest.mark.parametrize("dtype", DT)
def test_dropout_and_bce(ops, dtype):
    n = 4099
    x, gate = rnd(n, dtype=dtype, seed=1), rnd(n, dtype=dtype, seed=2)
    logits, y = rnd(3, 41, dtype=dtype, seed=3, scale=2.0), torch.from_numpy((np.random.default_rng(0).random((3, 41)) < 0.2).astype(np.float32))
    gs = torch.tensor([0.37])
    res = {}
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        rng = o.Rng(9, dev)
        d0 = o.dropout(mv(x), 0.5, rng, 7)
        d1 = o.dropout(mv(x), 0.5, rng, 7, relu_gate=mv(gate))
        d2 = o.dropout(mv(x), 0.0, None, 7)
        loss = torch.zeros(1, device=dev)
        o.bce_logits_fwd(mv(logits), mv(y), 5.0, loss)
        dx = o.bce_logits_bwd(mv(logits), mv(y), 5.0, mv(gs), torch.float32)
        res[dev] = (d0, d1, d2, loss, dx)
    for a, b, nme in zip(res["cuda"], res["cpu"], ("dropout", "dropout+gate", "p=0", "bce", "dbce")):
        close(a, b, nme, **tol(dtype if nme != "dbce" else torch.float32))
    kept = (res["cuda"][0].float() != 0).float().mean().item()
    assert 0.42 < kept < 0.58
    # against torch itself
    ref = torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), y, pos_weight=torch.tensor(5.0))
    assert abs(res["cuda"][3].item() - ref.item()) <= 2e-5 * abs(ref.item()) + (1e-2 if dtype == torch.bfloat16 else 0)


# ------------------------------------------------------------------------------------------------ fp8 operand mode (configs[4])
@pytest.mark.parametrize("bf8", [False, True])
@pytest.mark.parametrize("src_dtype", DT)
def test_fp8_quantisation_is_bit_exact(ops, bf8, src_dtype):
    """fs2_amax + fs2_quantize_fp8 against torch's own float8 conversion: the power-of-two scale and every code byte"""
    for scale, n in ((1.0, 5000), (3e-5, 4099), (700.0, 64), (0.0, 33)):
        x = rnd(n, dtype=src_dtype, seed=3, scale=scale)
        q, st = ops.quantize_fp8(x.cuda(), bf8)
        qr, sr = P.quantize_fp8(x, bf8)
        assert torch.equal(st.cpu(), sr), (st, sr)
        assert torch.equal(q.cpu(), qr), f"{int((q.cpu() != qr).sum())} of {n} codes differ (bf8={bf8}, scale={scale})"


@pytest.mark.parametrize("backward", [False, True])
def test_fp8_copy_written_by_the_gemm_epilogue_is_current_scaling(ops, backward):
    """FS2Gemm.q8: the fp8 ring kernel writes the fp8 copy of its bf16 output with the scale of LAST step's amax and reduces the new
    amax; fs2_quantize_fp8_repair re-quantises only when the binade moved.  Codes and {amax, 1/scale} must equal fs2_amax +
    fs2_quantize_fp8 of the output in every case: first step (no history), history that matches, history off by 2^7 either way."""
    M, N, K = 3000, 1024, 512
    x = rnd(M, K, dtype=torch.bfloat16, seed=1).cuda()
    w = (rnd(N, K, dtype=torch.bfloat16, seed=2) * K ** -0.5).cuda()
    bias, mask, cs = rnd(N, seed=3).cuda(), rnd(M, N, dtype=torch.bfloat16, seed=4).cuda(), torch.zeros(N, device="cuda")
    ops._FP8_STATES["buf"] = ops._FP8_STATES["prev"] = None
    ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = True, backward
    try:
        for step, factor in enumerate((1.0, 1.0, 1.0, 128.0, 1.0 / 128.0)):
            ops.fp8_begin_step(x.device)
            xs = (x * factor).to(torch.bfloat16)
            if backward:
                out = ops.linear(xs, w, relu_mask=mask, colsum=cs.zero_(), q8_out=True)
            else:
                out = ops.linear(xs, w, bias, relu=True, q8_out=True)
            assert ops.lib().fs2_gemm_last_tile() == 130
            q, st, bf8 = out._fs2_q8
            assert bf8 == backward
            q_ref, st_ref = ops.quantize_fp8(out, backward)
            assert torch.equal(st, st_ref), (step, st, st_ref)
            assert torch.equal(q, q_ref), (step, int((q != q_ref).sum()))
            # the consumer picks the copy up: same result as quantising the activation itself
            w2 = (rnd(256, N, dtype=torch.bfloat16, seed=5) * N ** -0.5).cuda()
            y = ops.linear(out, w2)
            del out._fs2_q8
            y_ref = ops.linear(out, w2)
            assert torch.equal(y, y_ref)
    finally:
        ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = False, False
        ops._FP8_STATES["buf"] = ops._FP8_STATES["prev"] = None


def test_fp8_copy_written_by_the_row_kernels_is_current_scaling(ops):
    """fs2_q8_next: the four LayerNorm-family kernels whose bf16 output is a GEMM operand (add_ln fwd / bwd, FFN tail fwd / bwd) write
    its fp8 copy themselves (e4m3 forward, e5m2 backward); codes and state equal the two-pass quantisation of the stored output for a
    first step, a matching history and a history in another binade"""
    M, d = 3000, 512
    dev = torch.device("cuda", torch.cuda.current_device())
    r, a = rnd(M, d, seed=1).cuda(), rnd(M, d, dtype=torch.bfloat16, seed=2).cuda()
    f2, h = rnd(M, d, dtype=torch.bfloat16, seed=3).cuda(), rnd(M, d, dtype=torch.bfloat16, seed=4).cuda()
    gam, bet = (1 + 0.1 * rnd(d, seed=5)).cuda(), (0.1 * rnd(d, seed=6)).cuda()
    rng = ops.Rng(7, dev)
    ops._FP8_STATES["buf"] = ops._FP8_STATES["prev"] = None
    ops.FP8_MODE["on"] = True
    try:
        for step, factor in enumerate((1.0, 1.0, 64.0, 1.0 / 64.0)):
            ops.fp8_begin_step(dev)
            rs, as_ = r * factor, (a * factor).to(torch.bfloat16)
            outs = []
            ops.FP8_MODE["backward"] = False
            s1, y1, mean, rstd = ops.add_ln_fwd(rs, as_, gam * factor, bet, p=0.1, rng=rng, site=3)
            outs.append((y1, False))
            s2, y2, m1, r1, m2, r2 = ops.ffn_tail_fwd(f2, h, rs, gam, bet, gam * factor, bet, p=0.1, rng=rng, site1=4, site2=5)
            outs.append((y2, False))
            ops.FP8_MODE["backward"] = True
            dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
            dr, da = ops.add_ln_bwd(None, as_, s1, gam, mean, rstd, dg, db, p=0.1, rng=rng, site=3)
            outs.append((da, True))
            dg2, db2, dg1, db1 = (torch.zeros(d, device=dev) for _ in range(4))
            dr2, g = ops.ffn_tail_bwd(None, as_, s2, gam, m2, r2, f2, h, gam, m1, r1, dg2, db2, dg1, db1, p=0.1, rng=rng, site1=4, site2=5)
            outs.append((g, True))
            for y, bf8 in outs:
                q, st, fmt = y._fs2_q8
                assert fmt == bf8
                q_ref, st_ref = ops.quantize_fp8(y, bf8)
                assert torch.equal(st, st_ref), (step, bf8, st, st_ref)
                assert torch.equal(q, q_ref), (step, bf8, int((q != q_ref).sum()))
            y3 = y1.view(3, 1000, d)                  # (a plain view drops the attribute; the models use ops.view2d)
            assert not hasattr(y3, "_fs2_q8")
            y3._fs2_q8 = y1._fs2_q8
            assert ops.view2d(y3, M, d)._fs2_q8[0].shape == (M, d)
    finally:
        ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = False, False
        ops._FP8_STATES["buf"] = ops._FP8_STATES["prev"] = None


def test_fp8_weight_gradients_against_the_dequantised_product(ops, big_km):
    """fp8 weight gradients (gemm_big_km.hip, ES = 1: dY in e5m2 x X in e4m3, ds_read_b64_tr_b8 fragments, v_mfma_f32_16x16x32_bf8_fp8):
    exact products of the fp8 values summed in fp32 and scaled by the two de-quantisation factors -- linear, the batched q/v/k form and
    Conv1d taps, launched on their own and as a deferred group, accumulating onto a non-zero gradient"""
    M, d = 2560, 256
    dy = rnd(M, 3 * d, dtype=torch.bfloat16, seed=1).cuda()
    x = rnd(M, d, dtype=torch.bfloat16, seed=2, scale=3.0).cuda()
    dyc, xc = rnd(5, 448, 256, dtype=torch.bfloat16, seed=3).cuda(), rnd(5, 448, 144, dtype=torch.bfloat16, seed=4).cuda()
    g0 = [rnd(3 * d, d, seed=5), rnd(3 * (d * d + d), seed=6), rnd(256, 3 * 144, seed=7)]

    def attach(t, bf8):
        q, st = ops.quantize_fp8(t, bf8)
        t._fs2_q8 = (q, st, bf8)
        return P.dequantize_fp8(q.cpu(), st.cpu(), bf8)

    ddy, dx, ddyc, dxc = attach(dy, True), attach(x, False), attach(dyc, True), attach(xc, False)
    ref0 = g0[0].double() + ddy.t() @ dx
    ref1 = [g0[1][j * (d * d + d):j * (d * d + d) + d * d].view(d, d).double() + ddy[:, j * d:(j + 1) * d].t() @ dx for j in range(3)]
    ref2 = P.conv_wgrad(ddyc.float(), dxc.float(), 3, 1, g0[2].clone())
    ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = True, True
    try:
        for defer in (False, True):
            outs = [t.cuda() for t in g0]
            kw = dict(defer=True) if defer else {}
            ops.wgrad(dy, x, outs[0], **kw)
            blocks = [outs[1][j * (d * d + d):j * (d * d + d) + d * d].view(d, d) for j in range(3)]
            ops.wgrad_batched(dy, x, blocks, **kw)
            ops.conv_wgrad(dyc, xc, 3, 1, outs[2], **kw)
            if defer:
                descs = [p[0] for p in ops._WG.pending]
                assert len(descs) == 3
                ops.wgrad_flush()
            close(outs[0], ref0.float(), f"fp8 wgrad defer={defer}", rtol=2e-5, atol=2e-5 * M ** 0.5 * 30)
            for b, r in zip(blocks, ref1):
                close(b, r.float(), f"fp8 batched wgrad defer={defer}", rtol=2e-5, atol=2e-5 * M ** 0.5 * 30)
            close(outs[2], ref2, f"fp8 conv wgrad defer={defer}", rtol=2e-3, atol=2e-3 * 2240 ** 0.5)
        # 144 output tiles: balanced-stream decomposition, float-atomic flush with the scales read on the device
        dys, xs = rnd(8, 128, 1024, dtype=torch.bfloat16, seed=8).cuda(), rnd(8, 128, 256, dtype=torch.bfloat16, seed=9).cuda()
        dd, dxs = attach(dys, True), attach(xs, False)
        o2 = torch.zeros(1024, 9 * 256, device="cuda")
        ops.conv_wgrad(dys, xs, 9, 4, o2, defer=True)
        ops.wgrad_flush()
        close(o2, P.conv_wgrad(dd.float(), dxs.float(), 9, 4, torch.zeros(1024, 9 * 256)), "fp8 stream-mode conv wgrad", rtol=2e-3, atol=2e-3 * 1024 ** 0.5)
        assert float((o2.cpu() - P.conv_wgrad(dys.cpu(), xs.cpu(), 9, 4, torch.zeros(1024, 9 * 256))).abs().max()) > 1e-3
        # and it is the fp8 kernel that ran: the result differs from the bf16 product of the unquantised operands
        full = g0[0].double() + dy.cpu().double().t() @ x.cpu().double()
        assert float((outs[0].cpu().double() - full).abs().max()) > 1e-3
    finally:
        ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = False, False


def test_fp8_batched_weight_quantisation_equals_the_per_tensor_path(ops):
    """fs2_quantize_fp8_batched (all weight shadows of a model in two launches) against fs2_amax + fs2_quantize_fp8 per tensor: the same
    codes and the same {amax, 1/scale}, for sizes with ragged tails, a zero tensor and more than one 32768-element chunk"""
    shapes = [(256, 256), (80, 512), (3, 7), (1024, 2304), (17, 33), (512, 1280)]
    ws = [rnd(*sh, dtype=torch.bfloat16, seed=i, scale=10.0 ** (i - 3)).cuda() for i, sh in enumerate(shapes)]
    ws.append(torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda"))
    ops._FP8_W.clear()
    ops.fp8_quantize_shadows(ws)
    for w in ws:
        if w.numel() < 16:
            assert w.data_ptr() not in ops._FP8_W
            continue
        q, st, n = ops._FP8_W[w.data_ptr()]
        q2, st2 = ops.quantize_fp8(w, False)
        assert torch.equal(q, q2) and torch.equal(st, st2), w.shape
    # a second call re-uses the table and re-quantises in place
    ws[0].mul_(3.0)
    ops.fp8_quantize_shadows(ws)
    q, st, n = ops._FP8_W[ws[0].data_ptr()]
    q2, st2 = ops.quantize_fp8(ws[0], False)
    assert torch.equal(q, q2) and torch.equal(st, st2)
    ops._FP8_W.clear()


@pytest.mark.parametrize("M,N,K", [(1500, 256, 256), (2048, 1024, 272), (1111, 264, 1024)])
def test_fp8_gemm_against_the_dequantised_product(ops, M, N, K):
    """fs2_gemm with fp8 operands (e4m3 x e4m3 forward, e5m2 x e4m3 backward): exact products of the fp8 values summed in fp32,
    scaled by the two de-quantisation factors, + bias / ReLU, against the fp64 product of the dequantised operands"""
    x, w, bias = rnd(M, K, dtype=torch.bfloat16, seed=1), rnd(N, K, dtype=torch.bfloat16, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    for backward in (False, True):
        ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = True, backward
        try:
            out = ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), relu=not backward, out_dtype=torch.float32)
            assert ops.lib().fs2_gemm_last_tile() in (130, 192)
        finally:
            ops.FP8_MODE["on"], ops.FP8_MODE["backward"] = False, False
        xq, sx = P.quantize_fp8(x, backward)
        wq, sw = P.quantize_fp8(w, False)
        ref = P.dequantize_fp8(xq, sx, backward) @ P.dequantize_fp8(wq, sw).t() + bias.double()
        if not backward:
            ref = ref.clamp(min=0)
        close(out, ref.float(), f"fp8 gemm backward={backward}", rtol=2e-5, atol=2e-5 * K ** 0.5)
        # and the quantisation error itself stays within the format's resolution (e4m3: 2^-4 relative per operand)
        full = x.double() @ w.double().t() + bias.double()
        if not backward:
            full = full.clamp(min=0)
        rel = float((out.cpu().double() - full).norm() / full.norm())
        assert rel < (0.08 if backward else 0.04), rel


# ------------------------------------------------------------------------------------------------ 16-wave weight-gradient GEMM
@pytest.fixture
def big_km(monkeypatch):
    monkeypatch.setenv("FS2_GEMM_BIG_KM", "2")
    yield
    monkeypatch.delenv("FS2_GEMM_BIG_KM", raising=False)


@pytest.mark.parametrize("M,N,K", [(700, 256, 256), (3000, 1024, 264), (129, 80, 72), (5000, 136, 520), (64, 8, 8)])
def test_big_km_wgrad(ops, big_km, M, N, K):
    """gemm_big_km.hip (in-workgroup k-split, LDS-DMA k-major images, LDS reduction, row-contiguous atomics): dW += dy^T x on
    ragged shapes (reduction tails, column edges), accumulating onto a non-zero gradient"""
    dy, x = rnd(M, N, dtype=torch.bfloat16, seed=1), rnd(M, K, dtype=torch.bfloat16, seed=2)
    g0 = rnd(N, K, seed=3)
    a = ops.wgrad(dy.cuda(), x.cuda(), g0.clone().cuda())
    assert ops.lib().fs2_gemm_last_tile() == 129
    b = P.wgrad(dy, x, g0.clone())
    close(a, b, "big km wgrad", rtol=2e-3, atol=2e-3 * M ** 0.5)


def test_big_km_wgrad_batched_and_conv(ops, big_km):
    M, d = 2500, 256
    dy, x = rnd(M, 3 * d, dtype=torch.bfloat16, seed=1), rnd(M, d, dtype=torch.bfloat16, seed=2)
    for o, dev in ((ops, "cuda"), (P, "cpu")):
        mv = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t.clone())
        arena = torch.zeros(3 * (d * d + d), device=dev)
        outs = [arena[j * (d * d + d):j * (d * d + d) + d * d].view(d, d) for j in range(3)]
        o.wgrad_batched(mv(dy), mv(x), outs)
        if dev == "cuda":
            assert ops.lib().fs2_gemm_last_tile() == 129
            got = [t.clone() for t in outs]
        else:
            for a, b in zip(got, outs):
                close(a, b, "big km batched wgrad", rtol=2e-3, atol=2e-3 * M ** 0.5)
    # (the last two shapes have tile counts no uniform k-split maps onto 256 CUs: 144 and 180 tiles run as a balanced stream,
    #  every workgroup the tail of one tile and -- first -- the head of the next)
    for (B, t, C, N, taps, pad) in ((7, 301, 72, 264, 9, 4), (5, 450, 256, 256, 5, 4), (9, 200, 80, 256, 3, 1), (8, 128, 256, 1024, 9, 4),
                                    (3, 400, 640, 512, 9, 4)):
        dyc, xc = rnd(B, t, N, dtype=torch.bfloat16, seed=3), rnd(B, t, C, dtype=torch.bfloat16, seed=4)
        a = ops.conv_wgrad(dyc.cuda(), xc.cuda(), taps, pad, torch.zeros(N, taps * C).cuda())
        assert ops.lib().fs2_gemm_last_tile() == 129
        b = P.conv_wgrad(dyc, xc, taps, pad, torch.zeros(N, taps * C))
        close(a, b, f"big km conv wgrad taps={taps}", rtol=2e-3, atol=2e-3 * (B * t) ** 0.5)


def test_sliced_weight_gradients(ops, big_km):
    """fs2_wgrad_sliced / fs2_wgrad_reduce: the partial tiles of the k-split go to a workspace with plain stores and ONE reduce launch
    adds the tiles of several deferred products (linear, batched q/v/k, Conv1d taps) onto non-zero gradients; against the oracle and
    against the float-atomics flush of the same kernel"""
    wg = ops._WG
    M, d = 2500, 256
    dy, x = rnd(M, 3 * d, dtype=torch.bfloat16, seed=1), rnd(M, d, dtype=torch.bfloat16, seed=2)
    dyc, xc = rnd(5, 450, 256, dtype=torch.bfloat16, seed=3), rnd(5, 450, 136, dtype=torch.bfloat16, seed=4)
    g0 = [rnd(3 * d, d, seed=5), rnd(3 * (d * d + d), seed=6), rnd(256, 3 * 136, seed=7)]

    def run(o, mv, defer):
        kw = dict(defer=True) if defer else {}
        outs = [mv(t) for t in g0]
        o.wgrad(mv(dy), mv(x), outs[0], **kw)
        blocks = [outs[1][j * (d * d + d):j * (d * d + d) + d * d].view(d, d) for j in range(3)]
        o.wgrad_batched(mv(dy), mv(x), blocks, **kw)
        o.conv_wgrad(mv(dyc), mv(xc), 3, 1, outs[2], **kw)
        return outs

    assert wg.enabled
    a = run(ops, lambda t: t.cuda(), True)
    assert len(wg.pending) == 3 and not wg.parts      # deferred products wait for their group launch: nothing has been added yet
    assert torch.equal(a[0].cpu(), g0[0])
    ops.wgrad_flush()
    assert not wg.parts and not wg.pending and wg.off == 0
    ref = run(P, lambda t: t.clone(), False)
    for u, v in zip(a, ref):
        close(u, v, "sliced wgrad", rtol=2e-3, atol=2e-3 * M ** 0.5)
    # immediate mode, and the atomics flush of the same kernel (different summation order only)
    b = run(ops, lambda t: t.cuda(), False)
    assert not wg.parts
    wg.enabled = False
    try:
        c = run(ops, lambda t: t.cuda(), False)
    finally:
        wg.enabled = True
    for u, v, w in zip(a, b, c):
        close(u, v, "grouped vs immediate", rtol=1e-4, atol=1e-3)
        close(u, w, "sliced vs atomics", rtol=1e-4, atol=1e-3)
    # two deferred products into ONE gradient (the mel Linear of the model receives two gradient terms): never in one reduce launch
    acc = g0[0].cuda()
    ops.wgrad(dy.cuda(), x.cuda(), acc, defer=True)
    ops.wgrad(dy.cuda(), x.cuda(), acc, defer=True)
    assert len(wg.pending) == 1 and not wg.parts      # the first term was launched and added before the second was queued
    ops.wgrad_flush()
    close(acc, 2 * a[0] - g0[0].cuda(), "two terms, one gradient", rtol=1e-5, atol=1e-3)
    # one by one (no grouping): the same partial sums per product, other split counts -> equal up to the summation order
    wg.group = False
    try:
        f = run(ops, lambda t: t.cuda(), True)
        assert len(wg.parts) == 3 and not wg.pending
        ops.wgrad_flush()
    finally:
        wg.group = True
    for u, v in zip(a, f):
        close(u, v, "grouped vs single launches", rtol=1e-4, atol=1e-3)
    # a small workspace (the group still fits: its k-splits are sized for ~256 items in all)
    keep = wg.FLOATS
    try:
        wg.FLOATS = 300 * 128 * 128
        wg.ws.clear()
        e = run(ops, lambda t: t.cuda(), True)
        ops.wgrad_flush()
    finally:
        wg.FLOATS = keep
        wg.ws.clear()
    for u, v in zip(a, e):
        assert torch.equal(u, v)
    # a group with a product the grouped launch does not take (144 output tiles: balanced-stream decomposition, float-atomic flush):
    # every product of the group is then launched on its own
    dys, xs = rnd(8, 128, 1024, dtype=torch.bfloat16, seed=8), rnd(8, 128, 256, dtype=torch.bfloat16, seed=9)
    o1, o2 = g0[0].cuda(), torch.zeros(1024, 9 * 256, device="cuda")
    ops.wgrad(dy.cuda(), x.cuda(), o1, defer=True)
    ops.conv_wgrad(dys.cuda(), xs.cuda(), 9, 4, o2, defer=True)
    assert len(wg.pending) == 2
    ops.wgrad_flush()
    close(o1, a[0], "fallback: linear", rtol=1e-4, atol=1e-3)
    close(o2, P.conv_wgrad(dys, xs, 9, 4, torch.zeros(1024, 9 * 256)), "fallback: stream-mode conv", rtol=2e-3, atol=2e-3 * 1024 ** 0.5)
    # balanced-stream products leave partial tiles too (slice = workgroup + tile; FS2WgradPart.splits < 0): a plain product with ragged
    # edges (8 x 18 tiles of 1000 x 2296) and the nine taps of a Conv1d, onto non-zero gradients, one reduce for both; against the oracle
    # and against the float-atomic flush of the same decomposition
    dyp, xp = rnd(2048, 1000, dtype=torch.bfloat16, seed=10), rnd(2048, 2296, dtype=torch.bfloat16, seed=11)
    g1, g2 = rnd(1000, 2296, seed=12), rnd(1024, 9 * 256, seed=13)
    o3, o4 = g1.cuda(), g2.cuda()
    ops.wgrad(dyp.cuda(), xp.cuda(), o3, defer=True)
    ops.conv_wgrad(dys.cuda(), xs.cuda(), 9, 4, o4, defer=True)
    ops.wgrad_launch()
    assert [q.splits < 0 for q in wg.parts] == [True, True], [q.splits for q in wg.parts]
    assert torch.equal(o3.cpu(), g1) and torch.equal(o4.cpu(), g2)       # nothing added before the reduce
    ops.wgrad_flush()
    r3, r4 = g1.clone(), g2.clone()
    P.wgrad(dyp, xp, r3)
    P.conv_wgrad(dys, xs, 9, 4, r4)
    close(o3, r3, "stream-mode product, partial tiles", rtol=2e-3, atol=2e-3 * 2048 ** 0.5)
    close(o4, r4, "stream-mode conv, partial tiles", rtol=2e-3, atol=2e-3 * 1024 ** 0.5)
    wg.enabled = False
    try:
        o5, o6 = g1.cuda(), g2.cuda()
        ops.wgrad(dyp.cuda(), xp.cuda(), o5)
        ops.conv_wgrad(dys.cuda(), xs.cuda(), 9, 4, o6)
    finally:
        wg.enabled = True
    close(o3, o5, "stream: partial tiles vs atomics", rtol=1e-4, atol=1e-3)
    close(o4, o6, "stream conv: partial tiles vs atomics", rtol=1e-4, atol=1e-3)


def test_torch_library_ops(ops):
    """the fs2:: custom operators (transformer_tts_amd/torch_ops.py, torch.library): forward and autograd through the dispatcher on the
    GPU against plain PyTorch fp32 -- nn.Linear + ReLU, channels-last Conv1d, attention() without dropout (causal and padded keys)."""
    import transformer_tts_amd.torch_ops  # noqa: F401
    dtype = torch.bfloat16
    x = rnd(5, 300, 256, dtype=dtype, seed=1).cuda().requires_grad_(True)
    w = (rnd(512, 256, dtype=dtype, seed=2) * 256 ** -0.5).cuda().requires_grad_(True)
    b = rnd(512, seed=3).cuda().requires_grad_(True)
    y = torch.ops.fs2.linear(x, w, b, True)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    yr = torch.relu(xr @ wr.t() + br)
    close(y.float(), yr, "fs2::linear", **tol(dtype))
    g = rnd(5, 300, 512, dtype=dtype, seed=4).cuda()
    y.backward(g)
    yr.backward(g.float())
    for a, r, n in ((x.grad, xr.grad, "dx"), (w.grad, wr.grad, "dw"), (b.grad, br.grad, "db")):
        err = float((a.float() - r).abs().max() / r.abs().max())
        assert err < 3e-2, (n, err)
    # Conv1d over time, channels-last
    xc, wc = rnd(3, 50, 64, dtype=dtype, seed=5).cuda(), (rnd(128, 64, 5, dtype=dtype, seed=6) * 320 ** -0.5).cuda()
    yc = torch.ops.fs2.conv1d_cl(xc, wc, None, 2, False)
    ref = torch.nn.functional.conv1d(xc.float().transpose(1, 2), wc.float(), padding=2).transpose(1, 2)
    close(yc.float(), ref, "fs2::conv1d_cl", **tol(dtype))
    # ... and its autograd: data, weight and bias gradients against torch's conv1d (ReLU on)
    xc2 = xc.detach().clone().requires_grad_(True)
    wc2 = wc.detach().clone().requires_grad_(True)
    bc2 = rnd(128, seed=9).cuda().requires_grad_(True)
    yc2 = torch.ops.fs2.conv1d_cl(xc2, wc2, bc2, 2, True)
    xr2, wr2, br2 = (t_.detach().float().requires_grad_(True) for t_ in (xc2, wc2, bc2))
    yr2 = torch.relu(torch.nn.functional.conv1d(xr2.transpose(1, 2), wr2, br2, padding=2).transpose(1, 2))
    close(yc2.float(), yr2, "fs2::conv1d_cl + bias + relu", **tol(dtype))
    gc = rnd(3, 50, 128, dtype=dtype, seed=10).cuda()
    yc2.backward(gc)
    yr2.backward(gc.float())
    for a, r, n in ((xc2.grad, xr2.grad, "conv dx"), (wc2.grad, wr2.grad, "conv dw"), (bc2.grad, br2.grad, "conv db")):
        err = float((a.float() - r).abs().max() / r.abs().max())
        assert err < 4e-2, (n, err)
    # attention
    B, H, t, dk = 2, 2, 130, 64
    # contiguous (B, H, t, dk) inputs -- how ordinary PyTorch code holds q / k / v -- go through as well
    qc, kc, vc = (rnd(B, H, t, dk, dtype=dtype, seed=20 + j, scale=1.5).cuda().requires_grad_(True) for j in range(3))
    kmc = torch.ones(B, t, dtype=torch.bool, device="cuda")
    kmc[0, 77:] = False
    oc, _ = torch.ops.fs2.flash_attention(qc, kc, vc, kmc, False)
    qf, kf, vf = (t_.detach().float().requires_grad_(True) for t_ in (qc, kc, vc))
    oc_ref = torch.softmax(((qf @ kf.transpose(-1, -2)) * dk ** -0.5).masked_fill(~kmc[:, None, None, :], -1e4), -1) @ vf
    assert float((oc.detach().float() - oc_ref.detach()).abs().max() / oc_ref.detach().abs().max()) < 3e-2
    goc = rnd(B, H, t, dk, dtype=dtype, seed=24).cuda()
    oc.backward(goc)
    oc_ref.backward(goc.float())
    for a, r, n in ((qc.grad, qf.grad, "dq"), (kc.grad, kf.grad, "dk"), (vc.grad, vf.grad, "dv")):
        err = float((a.float() - r).abs().max() / r.abs().max())
        assert err < 4e-2, ("contiguous " + n, err)
    qkv = rnd(B, t, 3, H, dk, dtype=dtype, seed=7, scale=1.5).cuda().requires_grad_(True)
    q, v, k = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    km = torch.ones(B, t, dtype=torch.bool, device="cuda")
    km[1, 100:] = False
    o, _ = torch.ops.fs2.flash_attention(q, k, v, km, True)
    qr = qkv.detach().float().requires_grad_(True)
    q2, v2, k2 = (qr[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    s = (q2 @ k2.transpose(-1, -2)) * dk ** -0.5
    vis = km[:, None, None, :] & torch.tril(torch.ones(t, t, dtype=torch.bool, device="cuda"))
    o_ref = torch.softmax(s.masked_fill(~vis, -1e4), -1) @ v2
    err = float((o.detach().float() - o_ref.detach()).abs().max() / o_ref.detach().abs().max())
    assert err < 3e-2, err
    go = rnd(B, H, t, dk, dtype=dtype, seed=8).cuda()
    o.backward(go)
    o_ref.backward(go.float())
    err = float((qkv.grad.float() - qr.grad).abs().max() / qr.grad.abs().max())
    assert err < 4e-2, err


def test_zero(ops):
    """fs2_zero (optimizer.zero_grad on the gradient arena): every byte of the range, nothing beyond it"""
    for n in (4, 16384 + 4, (1 << 22) + 12):
        buf = torch.full((n + 8,), 3.0, device="cuda")
        ops.zero(buf[4:4 + n])
        assert float(buf[4:4 + n].abs().max()) == 0.0 and buf[:4].tolist() == [3.0] * 4 and buf[4 + n:].tolist() == [3.0] * 4
    with pytest.raises(RuntimeError):
        ops.zero(torch.ones(6, device="cuda")[1:5])       # not 16-byte aligned

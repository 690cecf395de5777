"""End-to-end parity of the HIP-backed FastSpeech2 training path on the MI355X.

Exact-fp32 mode (hp.amp=False; f32-input MFMA) is the parity mode of BASELINE.json's north star:
mel L1 distance to the reference forward <= 1e-4 (measured here against golden vectors produced by the
imported reference, tests/golden/*.npz) -- the tolerance is written in the asserts below.  bf16 mode
(hp.amp=True) is the throughput mode; its stated tolerance is mean |mel - ref| <= 1.6e-2 = twice the largest
measured value (the reference's own CPU bf16 autocast is 2e-3..7e-3 off its fp64 forward; here
additionally the residual-free activations, attention probabilities and their gradients are stored in bf16)."""
import numpy as np
import pytest
import torch

from helpers import CONFIGS, batch_to, check_digest, is_null_gradient_param, oracle_model, product_model, record_measure

pytestmark = pytest.mark.gpu
OUT_NAMES = ["mel_before", "mel_after", "log_d", "p_pred", "e_pred", "va_out", "text_dur", "attn_enc", "attn_dec"]
MEL_L1_TOL_FP32 = 1e-4     # north star: "mel L1 within 1e-4 of reference"
# bf16 mode: 2 x the largest error measured on the MI355X over every fixture (mel_before 3.5e-3 .. 4.7e-3, mel_after 3.2e-3 .. 8.0e-3 mean
# |mel - reference|: tiny, small, the option fixtures, the inference fixtures, the configs[1] digests; gpurun_out/measured.jsonl of round
# 4, recorded by record_measure below) -- the reference's own CPU bf16 autocast sits at 2.1e-3 / 7.1e-3 against its fp64 forward
MEL_L1_TOL_BF16 = 1.6e-2
OPTIONS = ["opt_concat", "opt_nopitch", "opt_noenergy", "opt_ss1", "opt_ss_half"]      # golden_configs.OPTION_CONFIGS


def _seed(name):
    """scheduled sampling draws torch.rand(B) on the CPU generator inside the forward: seeded as the fixture recipe seeds it"""
    if "forward_seed" in CONFIGS[name]:
        torch.manual_seed(CONFIGS[name]["forward_seed"])


def fwd_bwd(model, hp, batch):
    from transformer_tts_amd.train_fastspeech2 import compute_losses, create_masks
    text, mel, pos_text, pos_mel, _, _, _, _, f0, energy, align = batch[:11]
    src_mask, mel_mask = create_masks(pos_text, pos_mel, task="fastspeech2")
    out = model(text, src_mask, mel_mask, align, f0, energy)
    total, parts = compute_losses(hp, out, mel, align, f0, energy)
    for p in model.parameters():
        p.grad = None
    total.backward()
    torch.cuda.synchronize()
    return out, total, parts


def knife_edge_l1_terms(name, g, batch, eps=5e-6):
    """L1 terms of a fixture whose REFERENCE prediction sits within eps of its target: d|pred - target|/dpred = sign(pred - target) of
    such an element follows the last bit of the forward, and every gradient moves with it by O(1 / #elements).  `opt_ss1` holds one
    (mel_after[2, 17, 44], 8e-7 from its target): a change of forward rounding that flips it fails the gradient checks of that fixture
    without being wrong (round 4: the fused stack-head forward in fp32).  Reported in the failure message, not hidden."""
    mel = batch[1].cpu().numpy()
    return {n: int((np.abs(g[f"out.{n}"] - mel) < eps).sum()) for n in ("mel_before", "mel_after")}


@pytest.mark.parametrize("name", ["tiny", "small", *OPTIONS])
def test_fp32_forward_backward_vs_reference_golden(name):
    model, hp, g = product_model(name, amp=False, device="cuda")
    _seed(name)
    out, total, parts = fwd_bwd(model, hp, batch_to(CONFIGS[name]["batch"](), "cuda"))
    assert len(out) == 14 and all(o is None for o in out[9:])
    assert (out[3] is None) == (not hp.pitch_pred) and (out[4] is None) == (not hp.energy_pred)
    for n, o in zip(OUT_NAMES, out[:9]):
        if o is None:
            continue
        ref = g[f"out.{n}"]
        got = o.detach().float().cpu().numpy()
        assert got.shape == ref.shape, n
        l1 = float(np.abs(got - ref).mean())
        assert l1 <= MEL_L1_TOL_FP32, f"{n}: mean |diff| {l1:.3e} > {MEL_L1_TOL_FP32}"
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-4, err_msg=n)
    golden_parts = dict(frame_before="mel", frame_after="post_mel", duration="duration", f0="f0", energy="energy")
    for k, v in parts.items():
        ref = float(g[f"loss.{golden_parts[k]}"])
        assert abs(v.item() - ref) <= 2e-5 * max(1.0, abs(ref)), (k, v.item(), ref)
    assert abs(total.item() - float(g["loss.total"])) <= 2e-5 * float(g["loss.total"])
    gsq = 0.0
    knife = knife_edge_l1_terms(name, g, CONFIGS[name]["batch"]())
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        check_digest(p.grad, g[f"graddig.{k}"], rtol=2e-3, atol=2e-5,
                     what=f"grad {k} (L1 terms of this fixture within 5e-6 of their target -- their gradient sign follows the forward's last bit: {knife})")
        gsq += float((p.grad.double() ** 2).sum())
    assert abs(gsq ** 0.5 - float(g["grad_global_norm"])) <= 1e-3 * float(g["grad_global_norm"])
    assert float(model.encoder.embed.weight.grad[0].abs().sum()) == 0.0


@pytest.mark.parametrize("name", ["tiny", "small", *OPTIONS])
def test_fp32_three_train_steps_vs_reference_train_loop(name):
    """FusedAdam (arena, fused clip) + trainer step against the reference's own train_loop digests."""
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import train_step
    model, hp, g = product_model(name, amp=False, device="cuda")
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0)
    batch = CONFIGS[name]["batch"]()
    step = int(g["train.start_step"])
    losses = []
    n_steps = CONFIGS[name]["train_steps"]
    for s in range(n_steps):
        _seed(name)
        loss, _, _ = train_step(model, opt, step, batch, hp)
        step += 1
        losses.append(loss.item())
        if s in (0, n_steps - 1):
            for k, v in model.state_dict().items():
                if is_null_gradient_param(k):
                    continue
                tol = dict(rtol=2e-4, atol=5e-6) if s == 0 else dict(rtol=2e-3, atol=1e-4)
                check_digest(v.float(), g[f"step{s + 1}.pdig.{k}"], what=f"step{s + 1} {k}", **tol)
    np.testing.assert_allclose(losses, g["train.loss_total"], rtol=5e-5)
    sd = opt.state_dict()
    assert int(sd["state"][0]["step"]) == n_steps and sd["state"][0]["exp_avg"].shape == model.encoder.embed.weight.shape


def test_fp32_vs_oracle_on_unseen_batch_with_dropout_off():
    """A batch no fixture covers: the oracle (CPU) and the HIP path must agree output by output."""
    from oracle import train as otrain
    from transformer_tts_amd import synthetic
    model, hp, g = product_model("small", amp=False, device="cuda")
    omodel, _, _ = oracle_model("small")
    batch = synthetic.make_batch(321, 5, l_range=(3, 30), dur_range=(0, 7), vocab=60)
    out, total, parts = fwd_bwd(model, hp, batch_to(batch, "cuda"))
    ototal, oparts, oout = otrain.forward_backward(omodel, batch)
    for n, a, b in zip(OUT_NAMES, out[:9], oout[:9]):
        torch.testing.assert_close(a.detach().float().cpu(), b.detach(), rtol=1e-3, atol=2e-4, msg=lambda m: f"{n}: {m}")
    assert abs(total.item() - ototal.item()) <= 2e-5 * abs(ototal.item())
    ograds = dict(omodel.named_parameters())
    for k, p in model.named_parameters():
        og = ograds[k].grad if ograds[k].grad is not None else torch.zeros_like(ograds[k])
        torch.testing.assert_close(p.grad.cpu(), og, rtol=5e-3, atol=5e-5, msg=lambda m: f"grad {k}: {m}")


@pytest.mark.parametrize("name", ["tiny", "small", "opt_concat", "opt_nopitch"])
def test_bf16_mode_within_stated_tolerance(name):
    model, hp, g = product_model(name, amp=True, device="cuda")
    out, total, parts = fwd_bwd(model, hp, batch_to(CONFIGS[name]["batch"](), "cuda"))
    for n in ("mel_before", "mel_after"):
        got = out[OUT_NAMES.index(n)].detach().float().cpu().numpy()
        l1 = float(np.abs(got - g[f"out.{n}"]).mean())
        record_measure(f"bf16.{name}.{n}.mean_abs_err", l1)
        assert l1 <= MEL_L1_TOL_BF16, f"{n}: mean |diff| {l1:.3e}"
    assert abs(total.item() - float(g["loss.total"])) <= 2e-2 * float(g["loss.total"])
    # gradients: direction agrees with the fp32 reference gradients on every sizeable tensor
    ref, _, _ = product_model(name, amp=False, device="cuda")
    fwd_bwd(ref, hp, batch_to(CONFIGS[name]["batch"](), "cuda"))
    rg = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        if p.numel() >= 1024 and not is_null_gradient_param(k):
            a, b = p.grad.double().flatten(), rg[k].grad.double().flatten()
            if float(b.norm()) > 1e-6:
                cos = float(a @ b / (a.norm() * b.norm() + 1e-30))
                assert cos > 0.98, (k, cos)


def test_benchmark_config_fp32_anchors():
    """BASELINE.json configs[1] (B=48, L_pad=128, T_pad=925) at full size in exact-fp32 mode against the
    reference's digests: outputs, the five losses, and the gradient norm."""
    model, hp, g = product_model("bench", amp=False, device="cuda", return_attn=False)
    batch = CONFIGS["bench"]["batch"]()
    assert int(batch[5].sum()) == 32172
    out, total, parts = fwd_bwd(model, hp, batch_to(batch, "cuda"))
    for n in OUT_NAMES[:7]:
        check_digest(out[OUT_NAMES.index(n)].float(), g[f"outdig.{n}"], rtol=2e-3, atol=3e-4, what=n)
    golden_parts = dict(frame_before="mel", frame_after="post_mel", duration="duration", f0="f0", energy="energy")
    for k, v in parts.items():
        ref = float(g[f"loss.{golden_parts[k]}"])
        assert abs(v.item() - ref) <= 5e-5 * max(1.0, abs(ref)), (k, v.item(), ref)
    gsq = sum(float((p.grad.double() ** 2).sum()) for p in model.parameters())
    assert abs(gsq ** 0.5 - float(g["grad_global_norm"])) <= 2e-3 * float(g["grad_global_norm"])
    for k, p in model.named_parameters():
        check_digest(p.grad, g[f"graddig.{k}"], rtol=1e-2, atol=2e-4, what=f"grad {k}")


@pytest.mark.parametrize("return_attn", [True, False])
def test_benchmark_config_bf16_within_stated_tolerance(return_attn):
    """The TIMED configuration (BASELINE.json configs[1] at full size, hp.amp=True: bf16 MFMA, LDS-strip / flash attention,
    split-K forward, XCD-aware tile walk) against the reference's digests.  Stated bf16 tolerance: mean |mel - ref| <= 1.6e-2
    (here over the digest's 64 evenly spaced samples and through the l2 norms), losses within 2 %, gradient norm within
    2 %, per-tensor gradient norms within 10 % for every sizeable non-null-gradient tensor."""
    model, hp, g = product_model("bench", amp=True, device="cuda", return_attn=return_attn)
    batch = CONFIGS["bench"]["batch"]()
    out, total, parts = fwd_bwd(model, hp, batch_to(batch, "cuda"))
    from golden_configs import sample_index
    for n in OUT_NAMES[:7]:
        dig = g[f"outdig.{n}"]
        x = out[OUT_NAMES.index(n)].detach().double().cpu().reshape(-1)
        assert x.numel() == int(dig[3]), n
        samples = x[torch.from_numpy(sample_index(x.numel()))].numpy()
        scale = max(1.0, float(np.abs(dig[4:]).mean()))
        l1 = float(np.abs(samples - dig[4:]).mean())
        if n in ("mel_before", "mel_after"):
            record_measure(f"bf16.bench.return_attn={return_attn}.{n}.mean_abs_err_over_samples", l1)
        assert l1 <= MEL_L1_TOL_BF16 * scale, f"{n}: mean |diff| over the samples {l1:.3e} (scale {scale:.2f})"
        l2 = float((x * x).sum().sqrt())
        assert abs(l2 - dig[2]) <= 2e-2 * dig[2], f"{n}: l2 {l2} vs {dig[2]}"
    golden_parts = dict(frame_before="mel", frame_after="post_mel", duration="duration", f0="f0", energy="energy")
    for k, v in parts.items():
        ref = float(g[f"loss.{golden_parts[k]}"])
        assert abs(v.item() - ref) <= 2e-2 * max(1.0, abs(ref)), (k, v.item(), ref)
    assert abs(total.item() - float(g["loss.total"])) <= 2e-2 * float(g["loss.total"])
    gsq = sum(float((p.grad.double() ** 2).sum()) for p in model.parameters())
    assert abs(gsq ** 0.5 - float(g["grad_global_norm"])) <= 2e-2 * float(g["grad_global_norm"])
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        dig = g[f"graddig.{k}"]
        if p.numel() >= 1024 and not is_null_gradient_param(k) and dig[2] > 1e-6:
            l2 = float(p.grad.double().norm())
            assert abs(l2 - dig[2]) <= 0.10 * dig[2], f"grad {k}: l2 {l2} vs {dig[2]}"


def test_flash_and_strip_attention_draw_the_same_masks(monkeypatch):
    """Three routes of the attention on the benchmark configuration with dropout 0.2: the LDS-strip kernels with the probabilities in
    HBM (hp.return_attn True, FS2_FLASH_MAPS=0: rounds 1-3), the flash kernels + the maps written after the fact (hp.return_attn True:
    the default) and the flash kernels alone (hp.return_attn False).  The Philox counters are those of the (B,N,H,t,tp) layout in every
    route, so all three see the same masks and differ by bf16 rounding only -- also in the returned maps, whose dropped positions
    coincide."""
    res = {}
    model, hp, g = product_model("bench", amp=True, dropout=0.2, device="cuda", return_attn=True)
    batch = batch_to(CONFIGS["bench"]["batch"](), "cuda")
    for route, ra, maps in (("strip", True, "0"), ("flash+maps", True, "1"), ("flash", False, "1")):
        monkeypatch.setenv("FS2_FLASH_MAPS", maps)
        model.rt.return_attn = ra           # same model: same call-site ids, same rng offset
        out, total, parts = fwd_bwd(model, hp, batch)
        assert (out[7] is None) == (not ra) and (out[8] is None) == (not ra)
        res[route] = (total.item(), {k: p.grad.double().flatten().cpu() for k, p in model.named_parameters()}, out[1].double().cpu(),
                      (out[7][:8].float().cpu(), out[8][:4, :2].float().cpu()) if ra else None)
    for other in ("flash+maps", "flash"):
        assert abs(res["strip"][0] - res[other][0]) <= 5e-3 * abs(res["strip"][0]), (other, res["strip"][0], res[other][0])
        a, b = res["strip"][2], res[other][2]
        assert float((a - b).abs().mean()) <= 2e-2 * max(1.0, float(a.abs().mean())), other
        for k, ga in res["strip"][1].items():
            gb = res[other][1][k]
            if ga.numel() >= 1024 and not is_null_gradient_param(k) and float(ga.norm()) > 1e-6:
                cos = float(ga @ gb / (ga.norm() * gb.norm() + 1e-30))
                assert cos > 0.97, (other, k, cos)
    # the returned maps (encoder: 8 utterances, decoder: 4 utterances x 2 layers): values to bf16 rounding, dropped positions identical
    for ms, mf, what in zip(res["strip"][3], res["flash+maps"][3], ("attn_enc", "attn_dec")):
        assert ms.shape == mf.shape
        assert float((ms - mf).abs().max()) <= 3e-2 * float(ms.max()), (what, float((ms - mf).abs().max()), float(ms.max()))
        sure = (ms > 1e-3) | (mf > 1e-3)            # (a position that holds a sizeable probability in one route ...)
        assert bool(((ms == 0) == (mf == 0))[sure].all()), f"{what}: the two routes dropped different positions"
        rows = ms.sum(-1)
        assert float((mf.sum(-1) - rows).abs().max()) <= 3e-2 * max(1.0, float(rows.max())), what


def test_dropout_statistics_and_replay():
    """p > 0 cannot be bit-matched with the reference's RNG: check the keep rate / scaling of the always-on
    attention dropout (Models/modules.py:19) and that two steps draw different masks while backward replays
    the forward's mask (finite loss decrease under training)."""
    model, hp, g = product_model("small", amp=False, dropout=0.2, device="cuda")
    batch = batch_to(CONFIGS["small"]["batch"](), "cuda")
    out, _, _ = fwd_bwd(model, hp, batch)
    attn = out[8].float()          # decoder maps, post-dropout
    pos_mel = batch[3]
    valid = (pos_mel != 0)
    b = 0
    n = int(valid[b].sum())
    rows = attn[b, :, :, :n, :n]
    zero_frac = float((rows == 0).float().mean())
    assert abs(zero_frac - 0.2) < 0.03, zero_frac
    assert abs(float(rows.sum(-1).mean()) - 1.0) < 0.05          # E[dropout(P)] row sums = 1
    model.rt.rng.advance()
    out2, _, _ = fwd_bwd(model, hp, batch)
    assert not torch.equal(out2[8], out[8])
    out3, _, _ = fwd_bwd(model, hp, batch)
    assert torch.equal(out3[8], out2[8]), "same rng offset -> same masks (deterministic replay)"


def test_hipgraph_replay_equals_eager_training():
    """GraphedTrainStep (one captured hipGraph per batch shape, weight gradients on a second stream) must follow the
    eager trainer: same Philox dropout streams, same kernels -> same losses and the same parameter update at every
    step, up to the order of the float atomics in the weight-gradient / reduction kernels.  That order changes a
    gradient by ~1e-7 relative, which the training dynamics amplify (an L1 loss has sign() gradients, Adam turns a
    near-zero gradient into a +-lr step), so the two trainers run in lockstep and the graphed one is re-synchronised to
    the eager one after every step: each comparison then covers exactly one update from identical state."""
    from transformer_tts_amd import synthetic
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep, train_step
    from transformer_tts_amd.Models import functional
    batches = [batch_to(synthetic.make_batch(100 + (i % 2), 4, l_range=(9, 20), dur_range=(1, 9), vocab=60), "cuda")
               for i in range(6)]
    pair = []
    for _ in range(2):
        functional._site_counter[0] = 1000      # same dropout call-site ids (= Philox streams) for both models
        model, hp, _ = product_model("small", amp=False, dropout=0.1, device="cuda")
        pair.append((model, hp, FusedAdam(model)))
    (m_e, hp_e, opt_e), (m_g, hp_g, opt_g) = pair
    stepper = GraphedTrainStep(m_g, opt_g, hp_g)
    for i, b in enumerate(batches):
        before = opt_e.arena.p.clone()
        out_e = train_step(m_e, opt_e, 4000 + i, b, hp_e)
        out_g = stepper(4000 + i, b)
        np.testing.assert_allclose(out_g[0].item(), out_e[0].item(), rtol=2e-6, err_msg=f"loss at step {i}")
        update = float((opt_e.arena.p - before).abs().max())            # the largest Adam step of this update
        diff = (opt_e.arena.p - opt_g.arena.p).abs()
        loose = diff > 1e-7 + 1e-6 * opt_e.arena.p.abs()
        assert float(diff.max()) <= 2.05 * update, (i, float(diff.max()), update)          # at worst one flipped step
        gerr = float((opt_e.arena.g - opt_g.arena.g).norm() / opt_e.arena.g.norm())
        assert gerr < 1e-4, (i, gerr)       # a flipped sign(pred - target) in an L1 term moves the gradient by ~1/N
        # Float-atomic order alone (gerr ~1e-7) leaves all but the near-zero gradients' updates in place.  When an L1 term sits within
        # that noise of its target (the post-net's BatchNorm statistics are summed with atomics, so mel_after's last bit differs between
        # two runs), its sign -- -1, 0 or +1 -- differs between the two trainers, every post-net gradient moves by ~1e-5 .. 1e-4 and
        # Adam turns that into > 1e-7 on every post-net parameter: seen once in ~25 runs of this file in round 4 (15 % of the
        # parameters "loose" at one step, gerr inside its bound).  The bound on gerr above is the check in that case.
        if gerr < 2e-6:
            assert float(loose.float().mean()) < 0.02, (i, int(loose.sum()), gerr)            # near-zero gradients only
        for dst, src in ((opt_g.arena.p, opt_e.arena.p), (opt_g.m, opt_e.m), (opt_g.v, opt_e.v)):
            dst.copy_(src)
        for bg, be in zip(m_g.buffers(), m_e.buffers()):
            bg.copy_(be)
    assert len(stepper.graphs) == 2, "two batch shapes -> two captured graphs"


def test_long_sequences_and_wide_model_vs_oracle():
    """t > 1024 (8-group softmax rows, many key tiles), L_pad > 128, d_model 128 with 4 heads of 32: the oracle and
    the HIP path on a batch no fixture covers (exact-fp32 mode)."""
    from types import SimpleNamespace
    from golden_configs import _BASE
    from oracle import train as otrain
    from oracle.model import FastSpeech2 as OracleFS2
    from transformer_tts_amd import synthetic
    from transformer_tts_amd.train_fastspeech2 import build_model
    from transformer_tts_amd.utils.utils import fill_variables
    d = dict(_BASE)
    d.update(vocab_size=50, batch_size=2, d_model_encoder=128, n_layer_encoder=1, n_head_encoder=4,
             ff_conv_kernel_size_encoder=5, d_model_decoder=128, n_layer_decoder=1, n_head_decoder=4,
             ff_conv_kernel_size_decoder=3, dropout=0.0, dropout_variance_adaptor=0.0)
    hp = SimpleNamespace(**d)
    fill_variables(hp, verbose=False)
    torch.manual_seed(3)
    omodel = OracleFS2.from_hp(hp, dropout=0.0, dropout_postnet=0.0, dropout_variance_adaptor=0.0)
    omodel.train()
    model = build_model(hp)
    model.postnet.dropout = 0.0
    model.load_state_dict(omodel.state_dict())
    model = model.cuda().train()
    batch = synthetic.make_batch(77, 2, l_range=(150, 171), dur_range=(5, 11), vocab=50)
    assert batch[1].shape[1] > 1024 and batch[0].shape[1] > 128
    out, total, parts = fwd_bwd(model, hp, batch_to(batch, "cuda"))
    ototal, _, oout = otrain.forward_backward(omodel, batch)
    for n, a, b in zip(OUT_NAMES[:7], out[:7], oout[:7]):
        torch.testing.assert_close(a.detach().float().cpu(), b.detach(), rtol=2e-3, atol=5e-4, msg=lambda m: f"{n}: {m}")
    for n, a, b in zip(OUT_NAMES[7:9], out[7:9], oout[7:9]):
        torch.testing.assert_close(a.detach().float().cpu(), b.detach(), rtol=2e-3, atol=2e-5, msg=lambda m: f"{n}: {m}")
    assert abs(total.item() - ototal.item()) <= 5e-5 * abs(ototal.item())
    og = dict(omodel.named_parameters())
    for k, p in model.named_parameters():
        g = og[k].grad if og[k].grad is not None else torch.zeros_like(og[k])
        torch.testing.assert_close(p.grad.cpu(), g, rtol=1e-2, atol=2e-4, msg=lambda m: f"grad {k}: {m}")


@pytest.mark.parametrize("name", ["tiny", "small", "inf_concat", "inf_nopitch", "inf_noenergy"])
def test_inference_branch_vs_reference_golden(name):
    """eval()-mode forward with predicted durations / pitch / energy (BatchNorm running statistics,
    nn.Dropout off) against fixtures from the reference's own inference forward (tests/golden/infer_*.npz).
    Exact-fp32 mode: rounded durations identical, mel L1 <= 1e-4; bf16 mode: same durations on these fixtures
    (their closest rounding margin is 1e-3) and the stated bf16 tolerance."""
    import os
    from helpers import GOLDEN
    g = np.load(os.path.join(GOLDEN, f"infer_{name}.npz"), allow_pickle=False)
    for amp in (False, True):
        model = product_model(name, amp=amp, device="cuda")[0]
        model.eval()
        for b in range(int(g["n_utt"])):
            text = torch.from_numpy(g[f"u{b}.text"]).cuda()
            pos = torch.arange(1, text.shape[1] + 1, device="cuda").unsqueeze(0)
            with torch.no_grad():
                out = model(text, (pos != 0).unsqueeze(-2))
            dur = torch.clamp(torch.round(torch.exp(out[2]) - 1), min=0).cpu().numpy()
            if not amp:
                assert np.array_equal(dur, g[f"u{b}.duration_rounded"]), (name, b)
            if not np.array_equal(dur, g[f"u{b}.duration_rounded"]):
                continue        # bf16: a duration that rounds the other way changes T; nothing to compare frame by frame
            for i, k in enumerate(OUT_NAMES[:7]):
                if out[i] is None:      # p_pred / e_pred with hp.pitch_pred / hp.energy_pred False (the inf_* option fixtures)
                    assert f"u{b}.{k}" not in g.files
                    continue
                ref = g[f"u{b}.{k}"]
                got = out[i].float().cpu().numpy()
                assert got.shape == ref.shape, (k, got.shape, ref.shape)
                l1 = float(np.abs(got - ref).mean())
                if amp and k in ("mel_before", "mel_after"):
                    record_measure(f"bf16.infer_{name}.utt{b}.{k}.mean_abs_err", l1)
                assert l1 <= (MEL_L1_TOL_BF16 if amp else MEL_L1_TOL_FP32), f"amp={amp} utt {b} {k}: mean |diff| {l1:.3e}"
        assert int(model.postnet.pre_batchnorm.num_batches_tracked) == 0


def test_inference_batch_of_utterances_bf16():
    """inference on a padded batch at benchmark-like sizes (the reference itself only supports B = 1 there: its frame
    mask broadcasts against the heads): every utterance of the batch must equal its own single-utterance run"""
    from transformer_tts_amd import synthetic
    model = product_model("small", amp=True, device="cuda")[0]
    model.eval()
    batch = synthetic.make_batch(31, 3, l_range=(20, 40), dur_range=(1, 9), vocab=60)
    text, pos_text, text_len = batch[0].cuda(), batch[2].cuda(), batch[4]
    with torch.no_grad():
        full = model(text, (pos_text != 0).unsqueeze(-2))
        dur = torch.clamp(torch.round(torch.exp(full[2]) - 1), min=0) * (pos_text != 0)
        lens = dur.sum(1).long().cpu()
        for b in range(text.shape[0]):
            n = int(text_len[b])
            one = model(text[b:b + 1, :n], (pos_text[b:b + 1, :n] != 0).unsqueeze(-2))
            T = one[0].shape[1]
            if T != int(lens[b]):
                continue        # bf16 noise moved a duration across a .5 boundary between the two batch shapes
            diff = (full[1][b, :T].float() - one[1][0].float()).abs().mean()
            record_measure(f"bf16.infer_batched_vs_single.utt{b}.mel_after.mean_abs_diff", float(diff))
            assert float(diff) < 2 * MEL_L1_TOL_BF16, (b, float(diff))      # (two bf16 results against each other: measured 1.7e-2)


@pytest.mark.parametrize("heads", [2, 4])
def test_d_model_512_config_vs_oracle(heads):
    """SURVEY 8(f) N4 shape family: d_model 512 (two float4 groups per LayerNorm row, 2048-wide FFN), 2 heads of 256
    (attention on the GEMM + softmax path) or 4 heads of 128 (LDS-strip attention kernels in bf16 mode): exact-fp32
    mode against the oracle, bf16 mode within the stated tolerance; forward outputs, loss and parameter gradients."""
    from types import SimpleNamespace
    from golden_configs import _BASE
    from oracle import train as otrain
    from oracle.model import FastSpeech2 as OracleFS2
    from transformer_tts_amd import synthetic
    from transformer_tts_amd.train_fastspeech2 import build_model
    from transformer_tts_amd.utils.utils import fill_variables
    d = dict(_BASE)
    d.update(vocab_size=50, batch_size=3, d_model_encoder=512, n_layer_encoder=1, n_head_encoder=heads,
             ff_conv_kernel_size_encoder=9, d_model_decoder=512, n_layer_decoder=1, n_head_decoder=heads,
             ff_conv_kernel_size_decoder=1, dropout=0.0, dropout_variance_adaptor=0.0)
    batch = synthetic.make_batch(91, 3, l_range=(20, 41), dur_range=(2, 9), vocab=50)
    results = {}
    for amp in (False, True):
        hp = SimpleNamespace(**dict(d, amp=amp))
        fill_variables(hp, verbose=False)
        torch.manual_seed(5)
        omodel = OracleFS2.from_hp(hp, dropout=0.0, dropout_postnet=0.0, dropout_variance_adaptor=0.0)
        omodel.train()
        model = build_model(hp)
        model.postnet.dropout = 0.0
        model.load_state_dict(omodel.state_dict())
        model = model.cuda().train()
        out, total, parts = fwd_bwd(model, hp, batch_to(batch, "cuda"))
        ototal, _, oout = otrain.forward_backward(omodel, batch)
        for n, a, b in zip(OUT_NAMES[:2], out[:2], oout[:2]):
            l1 = float((a.detach().float().cpu() - b.detach()).abs().mean())
            if amp:
                record_measure(f"bf16.flash_vs_oracle.{n}.mean_abs_err", l1)
            assert l1 <= (MEL_L1_TOL_BF16 if amp else MEL_L1_TOL_FP32), f"amp={amp} {n}: mean |diff| {l1:.3e}"
        assert abs(total.item() - ototal.item()) <= (2e-2 if amp else 5e-5) * abs(ototal.item())
        og = dict(omodel.named_parameters())
        num = den = 0.0
        for k, p in model.named_parameters():
            g = og[k].grad if og[k].grad is not None else torch.zeros_like(og[k])
            num += float((p.grad.float().cpu() - g).pow(2).sum())
            den += float(g.pow(2).sum())
            if not amp:
                torch.testing.assert_close(p.grad.cpu(), g, rtol=1e-2, atol=2e-4, msg=lambda m: f"grad {k}: {m}")
        rel = (num / den) ** 0.5
        assert rel < (6e-2 if amp else 1e-3), f"amp={amp}: relative gradient error {rel:.3e}"
        results[amp] = rel


# fp8 operand mode (BASELINE.json configs[4]): e4m3 activations / weights, e5m2 gradients, per-tensor power-of-two scaling.
# e4m3 carries 3 mantissa bits (relative step 2^-4 .. 2^-3 per operand element, errors average out over K >= 256 products):
# fp8 operand mode, 6+6 layers / d_model 512 against the fp32 oracle: 2 x the measured values (round 4, MI355X: mel 2.27e-2, whole-gradient
# relative L2 error 3.2e-2, loss 1.2e-5 relative; bf16 mode on the same model: 1.8e-3, 7.4e-3, 2e-7)
MEL_L1_TOL_FP8 = 5e-2          # mean |mel - oracle| with every eligible forward product in fp8
GRAD_REL_TOL_FP8 = 7e-2        # relative L2 error of the whole parameter gradient
GRAD_REL_TOL_BF16 = 1.5e-2
LOSS_REL_TOL_FP8 = 1e-3


def _config4_hp(layers, batch, fp8, amp=True):
    from types import SimpleNamespace
    from golden_configs import _BASE
    from transformer_tts_amd.utils.utils import fill_variables
    d = dict(_BASE)
    d.update(vocab_size=152, batch_size=batch, d_model_encoder=512, n_layer_encoder=layers, n_head_encoder=4,
             ff_conv_kernel_size_encoder=9, d_model_decoder=512, n_layer_decoder=layers, n_head_decoder=4,
             ff_conv_kernel_size_decoder=1, dropout=0.0, dropout_variance_adaptor=0.0, amp=amp, fp8=fp8)
    hp = SimpleNamespace(**d)
    fill_variables(hp, verbose=False)
    return hp


def test_config4_model_fp8_vs_oracle():
    """d_model 512, 6+6 FFT layers (the configs[4] architecture) on a batch the CPU oracle finishes in seconds: bf16 and fp8
    operand modes against the fp32 oracle, forward mel, loss and the full parameter gradient, fp8 within its stated tolerance
    and not better-than-plausible (the fp8 path must really have run: it differs from the bf16 result)"""
    from oracle import train as otrain
    from oracle.model import FastSpeech2 as OracleFS2
    from transformer_tts_amd import ops, synthetic
    from transformer_tts_amd.train_fastspeech2 import build_model
    batch = synthetic.make_batch(92, 12, l_range=(40, 81), dur_range=(2, 7), vocab=152)      # M = 12 x ~330 frames >= 1024 rows
    hp = _config4_hp(6, 12, False)
    torch.manual_seed(6)
    omodel = OracleFS2.from_hp(hp, dropout=0.0, dropout_postnet=0.0, dropout_variance_adaptor=0.0)
    omodel.train()
    ototal, _, oout = otrain.forward_backward(omodel, batch)
    og = dict(omodel.named_parameters())
    res = {}
    for fp8 in (False, True):
        hp = _config4_hp(6, 12, fp8)
        model = build_model(hp)
        model.postnet.dropout = 0.0
        model.load_state_dict(omodel.state_dict())
        model = model.cuda().train()
        assert model.rt.fp8 == fp8
        out, total, parts = fwd_bwd(model, hp, batch_to(batch, "cuda"))
        ops.FP8_MODE["on"] = False
        l1 = float((out[0].detach().float().cpu() - oout[0].detach()).abs().mean())
        num = den = 0.0
        for k, p in model.named_parameters():
            g = og[k].grad if og[k].grad is not None else torch.zeros_like(og[k])
            num += float((p.grad.float().cpu() - g).pow(2).sum())
            den += float(g.pow(2).sum())
        res[fp8] = (l1, abs(total.item() - ototal.item()) / abs(ototal.item()), (num / den) ** 0.5, out[0].detach().float().cpu())
    for mode, r in (("bf16", res[False]), ("fp8", res[True])):
        record_measure(f"{mode}.config4_6+6.mel_before.mean_abs_err", r[0])
        record_measure(f"{mode}.config4_6+6.loss_rel_err", r[1])
        record_measure(f"{mode}.config4_6+6.grad_rel_l2_err", r[2])
    assert res[False][0] <= MEL_L1_TOL_BF16 and res[False][2] < GRAD_REL_TOL_BF16, res[False][:3]
    assert res[True][0] <= MEL_L1_TOL_FP8, f"fp8 mel L1 {res[True][0]:.3e}"
    assert res[True][1] <= LOSS_REL_TOL_FP8, f"fp8 loss error {res[True][1]:.3e}"
    assert res[True][2] <= GRAD_REL_TOL_FP8, f"fp8 gradient error {res[True][2]:.3e}"
    assert float((res[True][3] - res[False][3]).abs().mean()) > 1e-4, "fp8 mode produced the bf16 result: it did not run"


def test_fp8_copies_from_the_epilogues_change_nothing():
    """the FFN activations whose fp8 copy is written by the producing GEMM's epilogue (FS2Gemm.q8, speculative scale + repair) against
    the two-pass quantisation of the same tensors over three steps (history: none, then set).  The codes are bit-identical
    (tests/test_kernels_gpu.py checks that on the kernels); here the whole step agrees to the noise of its float-atomic reductions
    (BatchNorm statistics, bias sums), orders of magnitude below the fp8 rounding itself"""
    from transformer_tts_amd import ops, synthetic
    from transformer_tts_amd.train_fastspeech2 import build_model
    batch = batch_to(synthetic.make_batch(92, 12, l_range=(40, 81), dur_range=(2, 7), vocab=152), "cuda")
    res = {}
    for fused in (True, False):
        ops.FP8_FUSED_OUT = fused
        ops._FP8_STATES["buf"] = ops._FP8_STATES["prev"] = None
        try:
            hp = _config4_hp(2, 12, True)
            torch.manual_seed(7)
            model = build_model(hp)
            model.postnet.dropout = 0.0
            model = model.cuda().train()
            runs = []
            for step in range(3):
                for p in model.parameters():
                    p.grad = None
                out, total, parts = fwd_bwd(model, hp, batch)
                runs.append((total.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
            res[fused] = runs
        finally:
            ops.FP8_MODE["on"] = False
            ops.FP8_FUSED_OUT = True
    for (la, ga), (lb, gb) in zip(res[True], res[False]):
        assert abs(la - lb) <= 1e-5 * abs(lb), (la, lb)
        num = sum(float((ga[k].double() - gb[k].double()).pow(2).sum()) for k in ga)
        den = sum(float(gb[k].double().pow(2).sum()) for k in ga)
        assert (num / den) ** 0.5 <= 1e-3, (num / den) ** 0.5


def test_config4_full_size_fp8_step_properties():
    """BASELINE.json configs[4] at full size (d_model 512, 6+6 layers, batch 64, reference dropout): one fp8 train step and
    one bf16 train step from the same weights -- finite losses that agree within the fp8 tolerance, parameters that moved"""
    from transformer_tts_amd import ops, synthetic
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import build_model, train_step
    batch = batch_to(synthetic.make_batch(2025, 64), "cuda")
    losses = {}
    sd = None
    for fp8 in (False, True):
        hp = _config4_hp(6, 64, fp8)
        hp.dropout, hp.dropout_variance_adaptor = 0.1, 0.5
        torch.manual_seed(7)
        model = build_model(hp)
        if sd is None:
            sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.load_state_dict(sd)
        model = model.cuda().train()
        opt = FusedAdam(model)
        before = opt.arena.p.clone()
        loss, parts, _ = train_step(model, opt, 4000, batch, hp)
        torch.cuda.synchronize()
        ops.FP8_MODE["on"] = False
        losses[fp8] = loss.item()
        assert np.isfinite(losses[fp8]) and float((opt.arena.p - before).abs().max()) > 0
    assert abs(losses[True] - losses[False]) <= 5e-2 * abs(losses[False]), losses

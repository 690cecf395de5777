#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (this container only).

Run:  python tests/golden/make_golden.py [tiny|small|bench|all]

The reference lives read-only at /root/reference and never travels to the GPU box; what is
committed is this script plus the small ``.npz`` fixtures it writes next to itself.  Nothing of
the reference's source is copied: the script calls the reference's own
``Models.fastspeech2.FastSpeech2`` and ``train_fastspeech2.train_loop`` /
``create_masks`` with harness-side import shims that touch no arithmetic:
a stub ``turtle`` (stray import at Models/modules.py:1), stub ``librosa`` / ``torchmetrics``
(imported, unused on this path) and a ``datasets`` namespace shim (the installed HuggingFace
``datasets`` package shadows the reference's directory).

Weights come from ``transformer_tts_amd.synthetic.recipe_state_dict`` (numpy RNG, sorted keys)
loaded with ``load_state_dict`` so tests can rebuild the identical model without the reference.
All dropouts are 0 (RNG streams cannot be matched); the model stays in ``train()`` mode because
BatchNorm batch statistics are part of the training arithmetic.
"""
import contextlib
import io
import os
import sys
import types
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(1, "/root/reference")

_t = types.ModuleType("turtle"); _t.distance = None; sys.modules["turtle"] = _t
sys.modules["librosa"] = types.ModuleType("librosa")
_tm = types.ModuleType("torchmetrics"); _tm.StructuralSimilarityIndexMeasure = lambda *a, **k: None
sys.modules["torchmetrics"] = _tm
_pkg = types.ModuleType("datasets"); _pkg.__path__ = ["/root/reference/datasets"]; sys.modules["datasets"] = _pkg

import torch  # noqa: E402

from transformer_tts_amd import synthetic  # noqa: E402
from golden_configs import CONFIGS, hp_namespace, digest  # noqa: E402


def build_reference(cfg):
    from Models.fastspeech2 import FastSpeech2
    from utils.utils import fill_variables
    hp = hp_namespace(cfg)
    with contextlib.redirect_stdout(io.StringIO()):
        fill_variables(hp)
    model = FastSpeech2(hp=hp, src_vocab=hp.vocab_size, trg_vocab=hp.mel_dim, d_model_encoder=hp.d_model_encoder,
                        N_e=hp.n_layer_encoder, n_head_encoder=hp.n_head_encoder,
                        ff_conv_kernel_size_encoder=hp.ff_conv_kernel_size_encoder,
                        concat_after_encoder=hp.concat_after_encoder, d_model_decoder=hp.d_model_decoder,
                        N_d=hp.n_layer_decoder, n_head_decoder=hp.n_head_decoder,
                        ff_conv_kernel_size_decoder=hp.ff_conv_kernel_size_decoder,
                        concat_after_decoder=hp.concat_after_decoder, dropout_variance_adaptor=0.0,
                        reduction_rate=hp.reduction_rate, dropout=0.0, dropout_postnet=0.0,
                        n_bins=hp.nbins, f0_min=hp.f0_min, f0_max=hp.f0_max, energy_min=hp.energy_min,
                        energy_max=hp.energy_max, pitch_pred=hp.pitch_pred, energy_pred=hp.energy_pred,
                        accent_emb=hp.accent_emb, output_type=hp.output_type, num_group=hp.num_group,
                        multi_speaker=hp.is_multi_speaker, spk_emb_dim=hp.spk_emb_dim,
                        spk_emb_architecture=hp.spk_emb_architecture)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synthetic.recipe_state_dict(shapes, cfg["weight_seed"]))
    model.train()
    return model, hp, shapes


def run(name):
    import train_fastspeech2 as T  # the reference trainer module (guarded by __main__)
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, hp, shapes = build_reference(cfg)
    batch = cfg["batch"]()
    text, mel, pos_text, pos_mel, text_len, mel_len, stop, _, f0, energy, align = batch[:11]
    full = name != "bench"
    out = {}
    out["shape_keys"] = np.array(sorted(shapes), dtype=object)
    out["shape_vals"] = np.array([str(shapes[k]) for k in sorted(shapes)], dtype=object)
    if full:
        for k, v in zip(synthetic.FIELDS[:11], batch[:11]):
            if v is not None:
                out[f"in.{k}"] = v.numpy()

    # ---- forward + losses + backward exactly as train_fastspeech2.py:153-167,212-259,312 (non-amp)
    src_mask, trg_mask = T.create_masks(pos_text, pos_mel, task=hp.model)
    res = model(text, src_mask, trg_mask, align, f0, energy, None, spkr_emb=None, fix_mask=None,
                temperature=None, hop_size=None)
    names = ["mel_before", "mel_after", "log_d", "p_pred", "e_pred", "va_out", "text_dur", "attn_enc", "attn_dec"]
    assert all(r is None for r in res[9:]), "default branch returns None in slots 9..13"
    L1 = torch.nn.L1Loss()
    losses = {
        "mel": L1(res[0], mel), "post_mel": L1(res[1], mel),
        "duration": L1(res[2], torch.log(align.float() + 1)),
        "f0": L1(res[3], f0), "energy": L1(res[4], energy)}
    total = losses["mel"] + losses["post_mel"] + losses["f0"] + losses["energy"] + losses["duration"]
    for p in model.parameters():
        p.grad = None
    total.backward()
    for n, r in zip(names, res[:9]):
        r = r.detach()
        if full:
            out[f"out.{n}"] = r.numpy()
        out[f"outdig.{n}"] = digest(r)
    for k, v in losses.items():
        out[f"loss.{k}"] = np.float64(v.item())
    out["loss.total"] = np.float64(total.item())
    gsq = 0.0
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"graddig.{k}"] = digest(g)
        if full and g.numel() <= 4096:
            out[f"grad.{k}"] = g.numpy().copy()
        gsq += float((g.double() ** 2).sum())
    out["grad_global_norm"] = np.float64(gsq ** 0.5)

    # ---- three optimizer steps through the reference's own train_loop (train_fastspeech2.py:100-315)
    model, hp, _ = build_reference(cfg)   # fresh weights + BN buffers
    hp.amp = False
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)  # :416
    nsteps = cfg["train_steps"]
    step = cfg["start_step"]
    log = io.StringIO()
    for s in range(nsteps):
        with contextlib.redirect_stdout(log):
            step = T.train_loop(model, opt, step, 0, SimpleNamespace(n_gpus=0), hp, 1, [batch])
        if s in (0, nsteps - 1):
            tag = f"step{s + 1}"
            for k, v in model.state_dict().items():
                out[f"{tag}.pdig.{k}"] = digest(v.float())
                if full and v.numel() <= 1024:
                    out[f"{tag}.p.{k}"] = v.numpy().copy()
    tl = [float(l.split("=")[1]) for l in log.getvalue().splitlines() if l.startswith("loss_total")]
    out["train.loss_total"] = np.asarray(tl, np.float64)
    out["train.start_step"] = np.int64(cfg["start_step"])
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB", "loss", out["loss.total"], "train", tl)


def run_infer(name):
    """Inference branch: the reference's own eval-mode forward with predicted durations, exactly as
    test_fastspeech2.py:159-171 calls it (one utterance per call, un-padded text, no targets).  -> infer_<name>.npz"""
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    model, hp, shapes = build_reference(cfg)
    model.eval()
    batch = cfg["batch"]()
    text, text_len = batch[0], batch[4]
    out = {"n_utt": np.int64(text.shape[0])}
    names = ["mel_before", "mel_after", "log_d", "p_pred", "e_pred", "va_out", "text_dur"]
    for b in range(text.shape[0]):
        n = int(text_len[b])
        tx = text[b:b + 1, :n]
        pos = torch.arange(1, n + 1).unsqueeze(0)
        src_mask = (pos != 0).unsqueeze(-2)
        with torch.no_grad():
            res = model(tx, src_mask, mel_mask=None, d_target=None, p_target=None, e_target=None, accent=None,
                        spkr_emb=None, fix_mask=None, pitch_perturbation=False, duration_perturbation=False, hop_size=None)
        dur = torch.clamp(torch.round(torch.exp(res[2]) - 1), min=0)          # test_fastspeech2.py:198
        frac = (torch.exp(res[2]) - 1) - torch.floor(torch.exp(res[2]) - 1)
        out[f"u{b}.text"] = tx.numpy()
        out[f"u{b}.duration_rounded"] = dur.numpy()
        out[f"u{b}.round_margin"] = np.float64((frac - 0.5).abs().min())       # distance of the closest call to a .5 tie
        for k, r in zip(names, res[:7]):
            out[f"u{b}.{k}"] = r.numpy()
        out[f"u{b}.attn_dec_dig"] = digest(res[8])
        print(name, "utt", b, "L", n, "T", int(dur.sum()), "round margin", float(out[f"u{b}.round_margin"]))
    path = os.path.join(HERE, f"infer_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which.startswith("infer"):
        for n in ("tiny", "small"):
            run_infer(n)
    else:
        for n in (list(CONFIGS) if which == "all" else [which]):
            run(n)

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
REF = "/root/reference"
OUT = HERE                         # --out DIR redirects the fixtures (tests/test_golden_recipe.py regenerates into a temp dir)
sys.dont_write_bytecode = True
# /root/reference FIRST: the repo root holds CLI shims named like the reference's scripts (train_fastspeech2.py,
# test_fastspeech2.py); packages of the build are imported by their qualified name (transformer_tts_amd.*) only.
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.path.insert(2, HERE)

_t = types.ModuleType("turtle"); _t.distance = None; sys.modules["turtle"] = _t
sys.modules["librosa"] = types.ModuleType("librosa")
_tm = types.ModuleType("torchmetrics"); _tm.StructuralSimilarityIndexMeasure = lambda *a, **k: None
sys.modules["torchmetrics"] = _tm
_pkg = types.ModuleType("datasets"); _pkg.__path__ = ["/root/reference/datasets"]; sys.modules["datasets"] = _pkg
# train.py (the autoregressive trainer) imports tensorboard's SummaryWriter at module level and never uses it
_tb = types.ModuleType("torch.utils.tensorboard"); _tb.SummaryWriter = lambda *a, **k: None; sys.modules["torch.utils.tensorboard"] = _tb

import importlib.util  # noqa: E402

import torch  # noqa: E402

from transformer_tts_amd import synthetic  # noqa: E402
from golden_configs import AR_CONFIGS, CONFIGS, hp_namespace, digest  # noqa: E402


def reference_module(name, relpath):
    """import one file of the reference BY PATH under a private module name: nothing on sys.path (the repo root has
    scripts of the same names) can shadow it"""
    key = "_ref_" + name
    if key in sys.modules:
        return sys.modules[key]
    spec = importlib.util.spec_from_file_location(key, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[key] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_trainer():
    return reference_module("train_fastspeech2", "train_fastspeech2.py")      # guarded by __main__


def shape_arrays(shapes):
    """{key: shape} -> pickle-free arrays: unicode keys, (n, 4) int64 dims padded with -1"""
    keys = sorted(shapes)
    dims = np.full((len(keys), 4), -1, np.int64)
    for i, k in enumerate(keys):
        dims[i, :len(shapes[k])] = shapes[k]
    return np.array(keys, dtype=np.str_), dims


def build_reference_raw(cfg, dropouts=(0.0, 0.0, 0.0)):
    """the reference model as its constructor leaves it (PyTorch default initialisation under the caller's seed);
    dropouts = (dropout, dropout_postnet, dropout_variance_adaptor): 0 for the fixtures, the trainer's own 0.1 / 0.5 / 0.5
    (train_fastspeech2.py:381-389) for the CPU timing of tools/cpu_ref_vs_oracle.py"""
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
                        concat_after_decoder=hp.concat_after_decoder, dropout_variance_adaptor=dropouts[2],
                        reduction_rate=hp.reduction_rate, dropout=dropouts[0], dropout_postnet=dropouts[1],
                        n_bins=hp.nbins, f0_min=hp.f0_min, f0_max=hp.f0_max, energy_min=hp.energy_min,
                        energy_max=hp.energy_max, pitch_pred=hp.pitch_pred, energy_pred=hp.energy_pred,
                        accent_emb=hp.accent_emb, output_type=hp.output_type, num_group=hp.num_group,
                        multi_speaker=hp.is_multi_speaker, spk_emb_dim=hp.spk_emb_dim,
                        spk_emb_architecture=hp.spk_emb_architecture)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    return model, hp, shapes


def build_reference(cfg, dropouts=(0.0, 0.0, 0.0)):
    model, hp, shapes = build_reference_raw(cfg, dropouts)
    model.load_state_dict(synthetic.recipe_state_dict(shapes, cfg["weight_seed"]))
    model.train()
    return model, hp, shapes


def run(name):
    T = ref_trainer()
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, hp, shapes = build_reference(cfg)
    batch = cfg["batch"]()
    text, mel, pos_text, pos_mel, text_len, mel_len, stop, _, f0, energy, align = batch[:11]
    full = name != "bench"
    out = {}
    out["shape_keys"], out["shape_dims"] = shape_arrays(shapes)
    if full:
        for k, v in zip(synthetic.FIELDS[:11], batch[:11]):
            if v is not None:
                out[f"in.{k}"] = v.numpy()

    # ---- forward + losses + backward exactly as train_fastspeech2.py:153-167,212-259,312 (non-amp)
    src_mask, trg_mask = T.create_masks(pos_text, pos_mel, task=hp.model)
    if "forward_seed" in cfg:       # scheduled sampling draws torch.rand(B) on the CPU generator inside the forward
        torch.manual_seed(cfg["forward_seed"])
    res = model(text, src_mask, trg_mask, align, f0, energy, None, spkr_emb=None, fix_mask=None,
                temperature=None, hop_size=None)
    names = ["mel_before", "mel_after", "log_d", "p_pred", "e_pred", "va_out", "text_dur", "attn_enc", "attn_dec"]
    assert all(r is None for r in res[9:]), "default branch returns None in slots 9..13"
    L1 = torch.nn.L1Loss()
    losses = {"mel": L1(res[0], mel), "post_mel": L1(res[1], mel), "duration": L1(res[2], torch.log(align.float() + 1))}
    total = losses["mel"] + losses["post_mel"]
    if hp.pitch_pred:               # :249-252
        losses["f0"] = L1(res[3], f0)
        total = total + losses["f0"]
    if hp.energy_pred:              # :254-257
        losses["energy"] = L1(res[4], energy)
        total = total + losses["energy"]
    total = total + losses["duration"]
    for p in model.parameters():
        p.grad = None
    total.backward()
    for n, r in zip(names, res[:9]):
        if r is None:               # p_pred / e_pred with hp.pitch_pred / hp.energy_pred False
            continue
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
        if "forward_seed" in cfg:
            torch.manual_seed(cfg["forward_seed"])
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
    path = os.path.join(OUT, f"{name}.npz")
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
            if r is not None:           # (p_pred / e_pred are None with hp.pitch_pred / hp.energy_pred False)
                out[f"u{b}.{k}"] = r.numpy()
        out[f"u{b}.attn_dec_dig"] = digest(res[8])
        print(name, "utt", b, "L", n, "T", int(dur.sum()), "round margin", float(out[f"u{b}.round_margin"]))
    path = os.path.join(OUT, f"infer_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def build_reference_ar(cfg):
    """the reference's Models.transformer.Transformer exactly as train.py:83-90 constructs it, fixture weights loaded"""
    from Models.transformer import Transformer
    hp = hp_namespace(cfg)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Transformer(hp=hp, src_vocab=hp.vocab_size, trg_vocab=hp.mel_dim,
                            d_model_encoder=hp.d_model_encoder, N_e=hp.n_layer_encoder, n_head_encoder=hp.n_head_encoder,
                            ff_conv_kernel_size_encoder=hp.ff_conv_kernel_size_encoder, concat_after_encoder=hp.concat_after_encoder,
                            d_model_decoder=hp.d_model_decoder, N_d=hp.n_layer_decoder, n_head_decoder=hp.n_head_decoder,
                            ff_conv_kernel_size_decoder=hp.ff_conv_kernel_size_decoder, concat_after_decoder=hp.concat_after_decoder,
                            reduction_rate=hp.reduction_rate, dropout=hp.dropout, dropout_prenet=hp.dropout_prenet,
                            dropout_postnet=hp.dropout_postnet, multi_speaker=hp.is_multi_speaker, spk_emb_dim=hp.spk_emb_dim,
                            spk_emb_architecture=hp.spk_emb_architecture)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synthetic.recipe_state_dict(shapes, cfg["weight_seed"]))
    model.train()
    return model, hp, shapes


def ar_iteration(model, optimizer, step, d, hp, R):
    """ONE iteration of the loop body of the reference's train.py:156-262.  train.py is a script (everything lives under
    `if __name__ == '__main__'`), so there is no function to call: this harness issues the same calls in the same order on
    the REFERENCE model -- R.create_masks is the reference's own (train.py:38-58), the model, losses and optimizer are
    torch / the reference's; non-amp branch (:251-257)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from utils.utils import get_learning_rate
    lr = get_learning_rate(step, d_model=hp.d_model_decoder, warmup_factor=hp.warmup_factor, warmup_step=hp.warmup_step)   # :160
    for g in optimizer.param_groups:
        g["lr"] = lr
    text, mel, pos_text, pos_mel, text_lengths, mel_lengths, stop_token, spk_emb = d                                    # :164
    if hp.reduction_rate > 1:                                                                                             # :183-189
        mel_input = mel[:, :-hp.reduction_rate:hp.reduction_rate, :]
        pos_mel = pos_mel[:, :-hp.reduction_rate:hp.reduction_rate]
    else:
        mel_input = mel[:, :-1, :]
        pos_mel = pos_mel[:, :-1]
    src_mask, trg_mask = R.create_masks(pos_text, pos_mel)                                                               # :195
    outputs_prenet, outputs_postnet, outputs_stop_token, attn_enc, attn_dec_dec, attn_dec_enc = \
        model(text, mel_input, src_mask, trg_mask, spk_emb)                                                              # :201
    optimizer.zero_grad()                                                                                                 # :205
    raw = (outputs_prenet, outputs_postnet, outputs_stop_token, attn_enc, attn_dec_dec, attn_dec_enc)
    if hp.reduction_rate > 1:                                                                                             # :207-211
        b, t, c = outputs_prenet.shape
        outputs_prenet = outputs_prenet.reshape(b, t * hp.reduction_rate, int(c // hp.reduction_rate))
        outputs_postnet = outputs_postnet.reshape(b, t * hp.reduction_rate, int(c // hp.reduction_rate))
        outputs_stop_token = outputs_stop_token.reshape(b, t * hp.reduction_rate)
    mel_loss = nn.L1Loss()(outputs_prenet, mel[:, hp.reduction_rate:, :])                                                # :214
    post_mel_loss = nn.L1Loss()(outputs_postnet, mel[:, hp.reduction_rate:, :])                                          # :215
    loss_token = F.binary_cross_entropy_with_logits(outputs_stop_token, stop_token[:, hp.reduction_rate:], reduction="mean",
                                                    pos_weight=torch.tensor(hp.positive_weight))                          # :217
    loss = mel_loss + post_mel_loss
    loss += loss_token
    parts = dict(mel=mel_loss.item(), post_mel=post_mel_loss.item(), token=loss_token.item(), total=loss.item())
    step += 1                                                                                                             # :247
    loss /= hp.accum_grad                                                                                                 # :259
    loss.backward()
    if step % hp.accum_grad == 0:                                                                                         # :262-264
        torch.nn.utils.clip_grad_norm_(model.parameters(), hp.clip)
        optimizer.step()
    return step, parts, raw


def run_ar(name):
    """the reference's autoregressive Transformer-TTS -- outputs, losses, gradients of one iteration and
    the parameters after the first / last of `train_steps` iterations.  -> <name>.npz"""
    R = reference_module("train_ar", "train.py")           # everything but the imports is guarded by __main__
    R.DEVICE = torch.device("cpu")
    cfg = AR_CONFIGS[name]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, hp, shapes = build_reference_ar(cfg)
    batch = cfg["batch"]()
    out = {}
    out["shape_keys"], out["shape_dims"] = shape_arrays(shapes)
    for k, v in zip(synthetic.FIELDS[:8], batch):
        if v is not None:
            out[f"in.{k}"] = v.numpy()
    # ---- one iteration: outputs, losses, gradients (the optimizer step it may take is on a throw-away optimizer)
    hp1 = SimpleNamespace(**dict(vars(hp), accum_grad=1, clip=1e30))      # (no clipping: the recorded gradients are the raw ones)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    _, parts, raw = ar_iteration(model, opt, cfg["start_step"], batch, hp1, R)
    names = ["outputs_prenet", "outputs_postnet", "stop_token", "attn_enc", "attn_dec_dec", "attn_dec_enc"]
    for n, r in zip(names, raw):
        out[f"out.{n}"] = r.detach().numpy()
        out[f"outdig.{n}"] = digest(r.detach())
    for k, v in parts.items():
        out[f"loss.{k}"] = np.float64(v)
    gsq = 0.0
    none = []
    for k, p in model.named_parameters():
        if p.grad is None:
            none.append(k)
            continue
        g = p.grad
        out[f"graddig.{k}"] = digest(g)
        if g.numel() <= 4096:
            out[f"grad.{k}"] = g.numpy().copy()
        gsq += float((g.double() ** 2).sum())
    out["grad_global_norm"] = np.float64(gsq ** 0.5)
    out["grad_none"] = np.array(none, dtype=np.str_)       # parameters the reference leaves without a gradient (its post-net)

    # ---- train_steps iterations on fresh weights (Adam as train.py:119)
    model, hp, _ = build_reference_ar(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    step = cfg["start_step"]
    tl = []
    for s in range(cfg["train_steps"]):
        step, parts, _ = ar_iteration(model, opt, step, batch, hp, R)
        tl.append(parts["total"])
        if s in (0, cfg["train_steps"] - 1):
            tag = f"step{s + 1}"
            for k, v in model.state_dict().items():
                out[f"{tag}.pdig.{k}"] = digest(v.float())
                if v.numel() <= 1024:
                    out[f"{tag}.p.{k}"] = v.numpy().copy()
    out["train.loss_total"] = np.asarray(tl, np.float64)
    out["train.start_step"] = np.int64(cfg["start_step"])
    out["train.end_step"] = np.int64(step)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB", "loss", out["loss.total"], "train", tl, "no-grad params", len(none))


def data_hp(root):
    """hparams of the BASELINE configs[0] plumbing run on the synthetic corpus in `root`"""
    from golden_configs import _BASE
    d = dict(_BASE)
    d.update(CONFIGS["tiny"]["hp"])
    d.update(train_script=os.path.join(root, "train.txt"), lengths_file=os.path.join(root, "lengths.npy"), vocab_size=152)
    return SimpleNamespace(**d)


DATA_BATCHES = ([0, 1], [2, 3, 4], [15], [5, 9, 13, 7])      # index lists fed to collate_fn (ragged, singleton, unordered)
DATA_MAX_SEQLEN = 300


def run_data():
    """the reference's OWN TrainDatasets.__getitem__ + collate_fn 16-tuples
    (datasets/datasets_fastspeech2.py:69-174,521-616), LengthsBatchSampler / NumBatchSampler batch lists (:749-845) and
    DistributedSamplerWrapper shards (:847-919) on the synthetic corpus of configs[0].  -> data.npz"""
    import tempfile
    from utils.utils import fill_variables
    from transformer_tts_amd.datasets.datasets_fastspeech2 import write_synthetic_corpus
    D = reference_module("datasets_fastspeech2", "datasets/datasets_fastspeech2.py")
    out = {}
    with tempfile.TemporaryDirectory() as root:
        write_synthetic_corpus(root)
        hp = data_hp(root)
        with contextlib.redirect_stdout(io.StringIO()):
            fill_variables(hp)
        ds = D.TrainDatasets(hp.train_script, hp, alignment_pred=True, pitch_pred=True, energy_pred=True, accent_emb=False)
        out["n_utt"] = np.int64(len(ds))
        for bi, idx in enumerate(DATA_BATCHES):
            tup = D.collate_fn([ds[i] for i in idx])
            assert len(tup) == 16
            out[f"b{bi}.index"] = np.asarray(idx, np.int64)
            out[f"b{bi}.is_none"] = np.asarray([t is None for t in tup], np.bool_)
            for k, t in zip(synthetic.FIELDS, tup):
                if torch.is_tensor(t):
                    out[f"b{bi}.{k}"] = t.numpy()
                    out[f"b{bi}.{k}.dtype"] = np.array(str(t.dtype))
            out[f"b{bi}.mel_name"] = np.array([os.path.basename(n) for n in tup[14]], dtype=np.str_)
            out[f"b{bi}.hop_size_none"] = np.asarray([h is None for h in tup[15]], np.bool_)
        with contextlib.redirect_stdout(io.StringIO()):
            lbs = D.LengthsBatchSampler(ds, DATA_MAX_SEQLEN, hp, hp.lengths_file, shuffle=False, shuffle_one_time=False)
        batches = list(lbs)
        out["lbs.flat"] = np.asarray([i for b in batches for i in b], np.int64)
        out["lbs.sizes"] = np.asarray([len(b) for b in batches], np.int64)
        out["lbs.len"] = np.int64(len(lbs))
        with contextlib.redirect_stdout(io.StringIO()):
            rev = list(D.LengthsBatchSampler(ds, DATA_MAX_SEQLEN, hp, hp.lengths_file, shuffle=False, reverse=True))
        out["lbs_rev.flat"] = np.asarray([i for b in rev for i in b], np.int64)
        np.random.seed(5)
        nbs = D.NumBatchSampler(ds, 3)          # 16 = 5 x 3 + 1: the ragged tail batch; order shuffled in ctor and per epoch
        for ep in range(2):
            bl = list(nbs)
            out[f"nbs.ep{ep}.flat"] = np.asarray([i for b in bl for i in b], np.int64)
            out[f"nbs.ep{ep}.sizes"] = np.asarray([len(b) for b in bl], np.int64)
        for world in (2, 3):
            for r in range(world):
                w = D.DistributedSamplerWrapper(lbs, num_replicas=world, rank=r)
                for ep in range(2):             # the reference never calls set_epoch: every epoch deals the same shard
                    bl = list(w)
                    out[f"dsw.w{world}.r{r}.ep{ep}.flat"] = np.asarray([i for b in bl for i in b], np.int64)
                    out[f"dsw.w{world}.r{r}.ep{ep}.sizes"] = np.asarray([len(b) for b in bl], np.int64)
                out[f"dsw.w{world}.r{r}.len"] = np.int64(len(w))
    path = os.path.join(OUT, "data.npz")
    np.savez_compressed(path, **out)
    print("data ->", path, f"{os.path.getsize(path) / 1024:.0f} KiB", "lbs", [len(b) for b in batches])


AR_DATA_CASES = (("r1", 1, False), ("r2", 2, False), ("r3n", 3, True))       # (tag, reduction rate, mean/variance normalised)
AR_DATA_BATCHES = ([0, 1], [2, 3, 4], [15], [5, 9, 13, 7], [10, 11, 12, 14, 6, 8])
AR_DATA_MAX_SEQLEN = 320


def run_ardata():
    """SURVEY 8(f) N2 input side: the reference's OWN autoregressive data path (datasets/datasets_transformer.py: TrainDatasets
    :18-103 with the all-zero go frame and the round-up of mel_length to the reduction rate, collate_fn :335-383 with the sort by mel
    length, the -5.0 / -0.5 mel pad and the 1.0 stop-token pad, LengthsBatchSampler :431-490, NumBatchSampler :492-522) on the
    synthetic corpus of configs[0], for reduction rates 1, 2, 3 with and without mean/variance files.  -> ardata.npz"""
    import tempfile
    from transformer_tts_amd.datasets.datasets_fastspeech2 import write_synthetic_corpus
    D = reference_module("datasets_transformer", "datasets/datasets_transformer.py")
    rhp = D.hp                      # the reference's module-level hparams singleton (utils/__init__.py): attributes set directly
    names = ("text", "mel", "pos_text", "pos_mel", "text_lengths", "mel_lengths", "stop_token")
    out = {}
    with tempfile.TemporaryDirectory() as root:
        write_synthetic_corpus(root)
        rng = np.random.default_rng(7)
        np.save(os.path.join(root, "mean.npy"), rng.standard_normal(80).astype(np.float32))
        np.save(os.path.join(root, "var.npy"), rng.uniform(0.5, 2.0, 80).astype(np.float32))
        for tag, r, norm in AR_DATA_CASES:
            lf = os.path.join(root, f"lengths_{tag}.npy")
            for k, v in dict(mel_dim=80, reduction_rate=r, spm_model=None, is_multi_speaker=False, lengths_file=lf,
                             mean_file=os.path.join(root, "mean.npy") if norm else None,
                             var_file=os.path.join(root, "var.npy") if norm else None).items():
                setattr(rhp, k, v)
            ds = D.TrainDatasets(os.path.join(root, "train.txt"), rhp)
            out[f"{tag}.n_utt"] = np.int64(len(ds))
            for bi, idx in enumerate(AR_DATA_BATCHES):
                tup = D.collate_fn([ds[i] for i in idx])
                assert len(tup) == 8 and tup[7] is None
                out[f"{tag}.b{bi}.index"] = np.asarray(idx, np.int64)
                for k, t in zip(names, tup):
                    out[f"{tag}.b{bi}.{k}"] = t.numpy()
                    out[f"{tag}.b{bi}.{k}.dtype"] = np.array(str(t.dtype))
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                lbs = D.LengthsBatchSampler(ds, AR_DATA_MAX_SEQLEN, lf, shuffle=False, shuffle_one_time=False)      # writes the lengths file
            out[f"{tag}.lengths"] = np.load(lf)
            bl = list(lbs)
            out[f"{tag}.lbs.flat"] = np.asarray([i for b in bl for i in b], np.int64)
            out[f"{tag}.lbs.sizes"] = np.asarray([len(b) for b in bl], np.int64)
            np.random.seed(5)
            nbs = D.NumBatchSampler(ds, 3)
            for ep in range(2):
                bl = list(nbs)
                out[f"{tag}.nbs.ep{ep}.flat"] = np.asarray([i for b in bl for i in b], np.int64)
                out[f"{tag}.nbs.ep{ep}.sizes"] = np.asarray([len(b) for b in bl], np.int64)
    path = os.path.join(OUT, "ardata.npz")
    np.savez_compressed(path, **out)
    print("ardata ->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def run_init():
    """the reference's ``init_weight`` (utils/utils.py:153-177) applied to the reference model built under
    ``torch.manual_seed(0)`` (the construction draws PyTorch's default initialisations in module order, ``apply`` then
    re-draws the Conv1d weights): digests of every state_dict entry.  -> init.npz"""
    from utils.utils import init_weight
    out = {}
    for name in ("tiny", "bench"):
        cfg = CONFIGS[name]
        torch.manual_seed(0)
        model, hp, shapes = build_reference_raw(cfg)
        model.apply(init_weight)
        keys, dims = shape_arrays({k: tuple(v.shape) for k, v in model.state_dict().items()})
        out[f"{name}.shape_keys"], out[f"{name}.shape_dims"] = keys, dims
        for k, v in model.state_dict().items():
            out[f"{name}.dig.{k}"] = digest(v.float())
        print("init", name, len(keys), "entries")
    path = os.path.join(OUT, "init.npz")
    np.savez_compressed(path, **out)
    print("init ->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--out" in argv:
        i = argv.index("--out")
        OUT = argv[i + 1]
        os.makedirs(OUT, exist_ok=True)
        del argv[i:i + 2]
    which = argv[0] if argv else "all"
    if which.startswith("infer"):
        for n in (("inf_concat", "inf_nopitch", "inf_noenergy") if which == "infer_options" else ("tiny", "small")):
            run_infer(n)
    elif which == "ar":
        for n in AR_CONFIGS:
            run_ar(n)
    elif which in AR_CONFIGS:
        run_ar(which)
    elif which == "data":
        run_data()
    elif which == "ardata":
        run_ardata()
    elif which == "init":
        run_init()
    elif which == "options":
        from golden_configs import OPTION_CONFIGS
        for n in OPTION_CONFIGS:
            if n.startswith("opt_"):
                run(n)
    else:
        for n in (list(CONFIGS) if which == "all" else [which]):
            run(n)

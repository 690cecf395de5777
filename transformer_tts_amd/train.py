"""Trainer of the autoregressive Transformer-TTS on the MI355X kernels -- the module surface of the reference's train.py
(``nopeak_mask``, ``create_masks``, the loop body of :156-262 as ``train_step`` / ``train_loop``, the model and optimizer
wiring of :83-119 as ``build_model`` / ``run_training``).  SURVEY.md section 8(f) N2, BASELINE.json configs[3].

The reference's train.py is a script (everything under ``__main__``); the loop body is exposed here as functions so that it
can be tested step by step against fixtures produced by the reference.  Quirks kept on purpose:
  * ``optimizer.zero_grad()`` runs in EVERY iteration (:205), so ``hp.accum_grad`` does not accumulate: it only thins the
    optimizer steps (one every accum_grad iterations, with the loss divided by accum_grad);
  * the post-net built with prev_version=False returns its input (Models/postnets.py:76-79): outputs_postnet is
    outputs_prenet, both L1 terms are equal and the post-net's parameters never receive a gradient (only its BatchNorm
    running statistics move);
  * the decoder sees frames 0 .. T-2 (reduction rate r: every r-th frame) and is trained against frames r .. T-1; frame 0 is the
    all-zero go frame ``datasets_transformer.TrainDatasets`` prepends, and T is a multiple of r (its collate_fn pads to one), so
    the (T - r) / r decoder steps produce exactly the T - r target frames.
"""
import argparse
import os
import sys
import time

import torch
from torch.utils.data import DataLoader

from .Models.functional import l1_loss, l1_loss_multi  # noqa: F401  (l1_loss: part of the module surface the tests import)
from .Models.functional_ar import bce_with_logits
from .Models.transformer import Transformer
from .optim import FusedAdam
from .utils import hparams as hp
from .utils.utils import fill_variables, get_learning_rate, init_weight, load_model, log_config
from .datasets import datasets_transformer as datasets

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


_NOPEAK = {}


def nopeak_mask(size, device=None):
    """reference train.py:26-36: (1, size, size) boolean lower-triangular (diagonal included) mask; built once per (size, device)
    (callers only read it: create_masks combines it with the padding mask into a new tensor)."""
    key = (int(size), str(device or DEVICE))
    m = _NOPEAK.get(key)
    if m is None:
        if len(_NOPEAK) >= 64:
            _NOPEAK.clear()
        m = _NOPEAK[key] = torch.tril(torch.ones((1, size, size), dtype=torch.bool, device=device or DEVICE))
    return m


def create_masks(src_pos, trg_pos, src_pad=0, trg_pad=0):
    """reference train.py:38-58."""
    from .train_fastspeech2 import _pad_mask      # (pos != pad).unsqueeze(-2); on the GPU one launch that also leaves the attention
    src_mask = _pad_mask(src_pos, src_pad)         #  kernels' row bounds / ranking with the mask (`_fs2_kinfo`)
    if trg_pos is not None:
        pad_mask = _pad_mask(trg_pos, trg_pad)
        trg_mask = pad_mask & nopeak_mask(trg_pos.size(1), trg_pos.device)
        if getattr(pad_mask, "_fs2_kinfo", None) is not None:
            trg_mask._fs2_kinfo = pad_mask._fs2_kinfo       # (of the frame padding; the kernels apply the no-peak part themselves)
    else:
        trg_mask = None
    return src_mask, trg_mask


def decoder_inputs(mel, pos_mel, r):
    """reference train.py:183-193 (transformer decoder)."""
    if r > 1:
        return mel[:, :-r:r, :], pos_mel[:, :-r:r]
    return mel[:, :-1, :], pos_mel[:, :-1]


def compute_losses(hp, outputs, mel, stop_token):
    """reference train.py:207-220: outputs regrouped to frame rate, two L1 terms against mel[:, r:], stop-token BCE."""
    r = hp.reduction_rate
    outputs_prenet, outputs_postnet, outputs_stop_token = outputs[:3]
    if r > 1:
        b, t, c = outputs_prenet.shape
        outputs_prenet = outputs_prenet.reshape(b, t * r, c // r)
        outputs_postnet = outputs_postnet.reshape(b, t * r, c // r)
        outputs_stop_token = outputs_stop_token.reshape(b, t * r)
    target = mel[:, r:, :].contiguous()
    if outputs_prenet.shape != target.shape or outputs_stop_token.shape != stop_token[:, r:].shape:
        raise ValueError(f"decoder output {tuple(outputs_prenet.shape)} does not cover the target frames {tuple(target.shape)}: the "
                         f"padded mel length ({mel.shape[1]}) must be a multiple of hp.reduction_rate ({r}) -- batches must come from "
                         "datasets_transformer.collate_fn (go frame + round-up), not from the FastSpeech2 reader")
    # the two L1 terms and their sum from one loss launch (+ its finishing block) each way; the stop-token term is added as the
    # reference adds it: (mel + post_mel) + token
    (l_mel, l_post), l_both = l1_loss_multi([(outputs_prenet, target, False), (outputs_postnet, target, False)])
    parts = {"mel": l_mel, "post_mel": l_post,
             "token": bce_with_logits(outputs_stop_token, stop_token[:, r:].contiguous(), float(hp.positive_weight))}
    loss = l_both + parts["token"]
    return loss, parts


def _set_lr(optimizer, step, hp):
    if hp.optimizer.lower() != "radam":
        lr = get_learning_rate(step, hp.d_model_decoder, hp.warmup_factor, hp.warmup_step)
        for param_group in optimizer.param_groups:
            param_group["lr"] = lr


def step_body(model, optimizer, hp, text, mel, pos_text, pos_mel, stop_token):
    """device work of one iteration with the fused optimizer and accum_grad = 1 (no host synchronisation: capturable in a hipGraph)"""
    mel_input, pos_in = decoder_inputs(mel, pos_mel, hp.reduction_rate)
    src_mask, trg_mask = create_masks(pos_text, pos_in)
    outputs = model(text, mel_input.contiguous(), src_mask, trg_mask, None)
    optimizer.zero_grad()
    loss, parts = compute_losses(hp, outputs, mel, stop_token)
    loss.backward()
    optimizer.launch()
    model.rt.get_rng(text.device).advance()
    return loss, parts


STEP_INPUTS = (0, 1, 2, 3, 6)       # entries of the 8-tuple a step reads: text, mel, pos_text, pos_mel, stop_token


def graphed_train_step(model, optimizer, hp, **kw):
    """train_step with one hipGraph per batch shape (train_fastspeech2.GraphedTrainStep over this module's step): accum_grad = 1 and
    the fused optimizer only; called as stepper(step, batch) -> (loss, parts, new step)"""
    from .train_fastspeech2 import GraphedTrainStep
    assert int(getattr(hp, "accum_grad", 1)) == 1, "the graph stepper applies the optimizer in every captured step"
    def body(model_, optimizer_, hp_, *static):
        # rt.check_masks compares the mask tensors on the host (a synchronisation: not capturable); the first occurrence of a batch
        # shape runs through the eager train_step, where the check is live
        keep = getattr(model_.rt, "check_masks", False)
        model_.rt.check_masks = False
        try:
            return step_body(model_, optimizer_, hp_, *static)
        finally:
            model_.rt.check_masks = keep

    g = GraphedTrainStep(model, optimizer, hp, body=body, inputs=STEP_INPUTS, set_lr=_set_lr,
                         eager=lambda m, o, st, d, h: train_step(m, o, st, d, h)[:2] + (None,), **kw)

    def stepper(step, d):
        loss, parts, _ = g(step, d)
        return loss, parts, step + 1
    stepper.graphs = g
    return stepper


def train_step(model, optimizer, step, d, hp):
    """One iteration of the reference loop body (train.py:156-262, non-amp arithmetic; hp.amp selects the bf16 kernels).
    Returns (loss tensor, parts, new step)."""
    _set_lr(optimizer, step, hp)
    text, mel, pos_text, pos_mel, text_lengths, mel_lengths, stop_token = d[:7]
    mv = lambda x: x.to(DEVICE, non_blocking=True)
    text, mel, pos_text, pos_mel, stop_token = (mv(x) for x in (text, mel, pos_text, pos_mel, stop_token))
    mel_input, pos_in = decoder_inputs(mel, pos_mel, hp.reduction_rate)
    src_mask, trg_mask = create_masks(pos_text, pos_in)
    fused = isinstance(optimizer, FusedAdam)
    outputs = model(text, mel_input.contiguous(), src_mask, trg_mask, None)
    optimizer.zero_grad()
    loss, parts = compute_losses(hp, outputs, mel, stop_token)
    step += 1
    accum = int(getattr(hp, "accum_grad", 1))
    (loss / accum if accum != 1 else loss).backward()
    if step % accum == 0:
        if fused:
            optimizer.host_update()                             # Adam's step count advances only when it steps
            optimizer.launch()                                  # global-norm clip fused into the Adam kernel
        else:
            torch.nn.utils.clip_grad_norm_(model.parameters(), hp.clip)
            optimizer.step()
            model.rt.invalidate()
    model.rt.get_rng(text.device).advance()
    return loss, parts, step


def train_loop(model, optimizer, step, epoch, hp, dataloader, log_every=1, stepper=None):
    """stepper: graphed_train_step(...) (one hipGraph per batch shape) or None = eager train_step"""
    n_run = 0
    for d in dataloader:
        loss, parts, step = stepper(step, d) if stepper is not None else train_step(model, optimizer, step, d, hp)
        n_run += 1
        if n_run == 3:              # (train_fastspeech2.settle_gc: the run's long-lived objects leave the garbage collector's generations)
            from .train_fastspeech2 import settle_gc
            settle_gc()
        assert not torch.isnan(loss), "loss is nan"          # every iteration, as the reference does (train.py:236)
        if (step - 1) % log_every == 0:
            print(f"step {step - 1}")
            print(f"loss_token = {parts['token'].item()}")
            print(f"loss_frame_before = {parts['mel'].item()}")
            print(f"loss_frame_after = {parts['post_mel'].item()}")
            print(f"loss_total = {loss.item()}")
            print(f"batch size = {d[1].shape[0]}")
            sys.stdout.flush()
    if (epoch + 1) >= (hp.max_epoch - 10) or (epoch + 1) % hp.save_per_epoch >= (hp.save_per_epoch - 10) or \
            (epoch + 1) % hp.save_per_epoch == 0:
        torch.save(model.state_dict(), hp.save_dir + "/network.epoch{}".format(epoch + 1))
    if (epoch + 1) % hp.save_per_epoch == 0:
        torch.save(optimizer.state_dict(), hp.save_dir + "/network.optimizer.epoch{}".format(epoch + 1))
    return step


def build_model(hp):
    """argument wiring of the reference (train.py:83-90)."""
    return Transformer(hp=hp, src_vocab=hp.vocab_size, trg_vocab=hp.mel_dim, d_model_encoder=hp.d_model_encoder,
                       N_e=hp.n_layer_encoder, n_head_encoder=hp.n_head_encoder,
                       ff_conv_kernel_size_encoder=hp.ff_conv_kernel_size_encoder, concat_after_encoder=hp.concat_after_encoder,
                       d_model_decoder=hp.d_model_decoder, N_d=hp.n_layer_decoder, n_head_decoder=hp.n_head_decoder,
                       ff_conv_kernel_size_decoder=hp.ff_conv_kernel_size_decoder, concat_after_decoder=hp.concat_after_decoder,
                       reduction_rate=hp.reduction_rate, dropout=hp.dropout, dropout_prenet=hp.dropout_prenet,
                       dropout_postnet=hp.dropout_postnet, multi_speaker=hp.is_multi_speaker, spk_emb_dim=hp.spk_emb_dim,
                       spk_emb_architecture=hp.spk_emb_architecture)


def run_training(hp):
    model = build_model(hp).to(DEVICE)
    model.apply(init_weight)
    model.train()
    optimizer = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=hp.clip)
    os.makedirs(hp.save_dir, exist_ok=True)
    dataset_train = datasets.TrainDatasets(hp.train_script, hp)                    # reference train.py:129-137
    collate_fn = datasets.make_collate_fn(hp)
    assert (hp.batch_size is None) != (hp.max_seqlen is None)
    if hp.batch_size is not None:
        sampler = datasets.NumBatchSampler(dataset_train, hp.batch_size)
    else:
        sampler = datasets.LengthsBatchSampler(dataset_train, hp.max_seqlen, hp.lengths_file, shuffle=True, shuffle_one_time=False)
    if hp.loaded_epoch is not None:
        start_epoch = hp.loaded_epoch
        print("epoch {} loaded".format(hp.loaded_epoch))
        model.load_state_dict(load_model(os.path.join(hp.loaded_dir, "network.epoch{}".format(hp.loaded_epoch)), map_location=DEVICE))
        model.rt.invalidate()
        loaded = torch.load(os.path.join(hp.loaded_dir, "network.optimizer.epoch{}".format(hp.loaded_epoch)), map_location=DEVICE,
                            weights_only=True)
        optimizer.load_state_dict(loaded)
        step = int(loaded["state"][0]["step"]) * int(getattr(hp, "accum_grad", 1))
    else:
        start_epoch, step = 0, 1
    # on the GPU the device work of a step is captured once per batch shape and replayed (hp.use_graph = False keeps eager launches)
    stepper = None
    if DEVICE.type == "cuda" and bool(getattr(hp, "use_graph", True)) and int(getattr(hp, "accum_grad", 1)) == 1:
        stepper = graphed_train_step(model, optimizer, hp, eager_fallback=True)
    for epoch in range(start_epoch, hp.max_epoch):
        dataloader = DataLoader(dataset_train, batch_sampler=sampler, num_workers=int(getattr(hp, "num_workers", 4)),
                                collate_fn=collate_fn)
        start_time = time.time()
        step = train_loop(model, optimizer, step, epoch, hp, dataloader, max(1, int(getattr(hp, "log_every", 1))), stepper)
        print("EPOCH {} end".format(epoch + 1))
        print(f"elapsed time {time.time() - start_time}")
    from .train_fastspeech2 import unsettle_gc
    unsettle_gc()           # (train_loop froze the run's long-lived objects out of the garbage collector's generations)
    return step


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--hp_file", type=str, default="hparams.py")
    args = parser.parse_args(argv)
    hp.configure(args.hp_file)
    fill_variables(hp)
    log_config(hp)
    run_training(hp)


if __name__ == "__main__":
    main()

"""Synthesis script of the FastSpeech2 path -- the reference's ``test_fastspeech2.py`` (:87-204) on the gfx950 kernels:
load a checkpoint written by ``train_fastspeech2.py`` (or by the reference), run the inference branch of
``Models.fastspeech2.FastSpeech2`` (predicted durations / pitch / energy) one utterance at a time, undo the optional
mean/variance normalisation and write ``<name>.npy`` (T, mel_dim) and ``<name>_alignment.npy`` (rounded durations).

Same command line as the reference: ``--load_name <checkpoint> [--test_script f] [--save] [--use_prenet]
[--pitch_perturbation] [--duration_perturbation]``; the hyper-parameters come from ``hparams.py`` next to the
checkpoint.  As in the reference the model is built with every dropout rate 0 (:124-131) and put in ``eval()``.
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from .datasets import datasets_fastspeech2 as datasets
from .Models.fastspeech2 import FastSpeech2
from .utils import hparams as hp
from .utils.utils import fill_variables, load_model

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def build_inference_model(hp):
    """Argument wiring of the reference (test_fastspeech2.py:124-131): all dropout rates 0."""
    return FastSpeech2(hp, src_vocab=hp.vocab_size, trg_vocab=hp.mel_dim, d_model_encoder=hp.d_model_encoder,
                       N_e=hp.n_layer_encoder, n_head_encoder=hp.n_head_encoder,
                       ff_conv_kernel_size_encoder=hp.ff_conv_kernel_size_encoder,
                       concat_after_encoder=hp.concat_after_encoder, d_model_decoder=hp.d_model_decoder,
                       N_d=hp.n_layer_decoder, n_head_decoder=hp.n_head_decoder,
                       ff_conv_kernel_size_decoder=hp.ff_conv_kernel_size_decoder,
                       concat_after_decoder=hp.concat_after_decoder, reduction_rate=hp.reduction_rate, dropout=0.0,
                       dropout_postnet=0.0, dropout_variance_adaptor=0.0, n_bins=hp.nbins, f0_min=hp.f0_min,
                       f0_max=hp.f0_max, energy_min=hp.energy_min, energy_max=hp.energy_max, pitch_pred=hp.pitch_pred,
                       energy_pred=hp.energy_pred, accent_emb=hp.accent_emb, output_type=hp.output_type,
                       num_group=hp.num_group, multi_speaker=hp.is_multi_speaker, spk_emb_dim=hp.spk_emb_dim,
                       spk_emb_architecture=hp.spk_emb_architecture)


def synthesize(model, text, pos_text, hp, use_prenet=False, pitch_perturbation=False, duration_perturbation=False,
               mean_value=None, var_value=None):
    """One call of the reference's loop body (:159-186): returns (mel (T, mel_dim) numpy, rounded durations (L,) numpy)."""
    src_mask = (pos_text != 0).unsqueeze(-2)
    with torch.no_grad():
        out = model(text, src_mask, mel_mask=None, d_target=None, p_target=None, e_target=None, accent=None,
                    spkr_emb=None, fix_mask=getattr(hp, "fix_mask", None), pitch_perturbation=pitch_perturbation,
                    duration_perturbation=duration_perturbation, hop_size=None)
    outputs_prenet, outputs_postnet, log_d_prediction = out[0], out[1], out[2]
    mel = (outputs_postnet if (hp.postnet_pred and not use_prenet) else outputs_prenet)[0].float().cpu().numpy()
    if var_value is not None:
        mel *= np.sqrt(var_value)
    if mean_value is not None:
        mel += mean_value
    duration_rounded = torch.clamp(torch.round(torch.exp(log_d_prediction) - 1), min=0)        # :198
    return mel, duration_rounded.cpu().numpy()[0]


def main(argv=None):
    random.seed(77)
    parser = argparse.ArgumentParser()
    parser.add_argument("--load_name", required=True)
    parser.add_argument("--test_script", default=None)
    parser.add_argument("--save", action="store_true")
    parser.add_argument("--use_prenet", action="store_true")
    parser.add_argument("--pitch_perturbation", action="store_true")
    parser.add_argument("--duration_perturbation", action="store_true")
    args = parser.parse_args(argv)
    load_name = args.load_name

    hp_file = os.path.join(os.path.dirname(load_name), "hparams.py")
    assert os.path.exists(hp_file), f"{hp_file}: the hyper-parameters are read from hparams.py next to the checkpoint"
    hp.configure(hp_file)
    fill_variables(hp)
    epoch = os.path.basename(load_name).replace("network.average_", "")
    save_path = os.path.join(os.path.dirname(load_name), "dev.7/" + epoch)
    os.makedirs(save_path, exist_ok=True)
    assert hp.architecture == "text-mel", f"invalid architecture {hp.architecture}"
    if args.test_script is not None:
        hp.test_script = args.test_script
    print(f"use_prenet = {args.use_prenet}")

    model = build_inference_model(hp)
    model.to(DEVICE)
    model.eval()
    state = load_model(load_name, map_location=DEVICE)
    model.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in state.items()})
    model.rt.invalidate()

    dataset_test = datasets.TestDatasets(hp.test_script, hp, accent_emb=hp.accent_emb)
    sampler = datasets.NumBatchSampler(dataset_test, 1, shuffle=False)
    dataloader = DataLoader(dataset_test, batch_sampler=sampler, num_workers=0, collate_fn=datasets.collate_fn_test)
    mean_value = var_value = None
    if hp.mean_file is not None and hp.var_file is not None:
        mean_value = np.load(hp.mean_file).reshape(-1, hp.mel_dim)
        var_value = np.load(hp.var_file).reshape(-1, hp.mel_dim)

    start_time = time.time()
    total_time = 0.0
    for idx, d in enumerate(dataloader):
        text, names, pos_text = d[0].to(DEVICE), d[1], d[2].to(DEVICE)
        local_time = time.time()
        outputs, duration_rounded = synthesize(model, text, pos_text, hp, args.use_prenet, args.pitch_perturbation,
                                               args.duration_perturbation, mean_value, var_value)
        total_time += time.time() - local_time
        if args.save:
            output_name = names[0]
        else:
            output_name = os.path.join(save_path, os.path.splitext(os.path.basename(names[0]))[0] + ".npy")
        print(f"save {output_name} {outputs.shape}")
        np.save(output_name, outputs)
        np.save(output_name.replace(".npy", "_alignment.npy"), duration_rounded)
        sys.stdout.flush()
    print(f"elapsed time = {time.time() - start_time}")
    print(f"total_time = {total_time}")


if __name__ == "__main__":
    main()

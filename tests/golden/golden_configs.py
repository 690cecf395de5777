"""Configurations of the golden fixtures (shared by make_golden.py and the tests).

hparams keys are the working set of SURVEY.md Appendix A (the reference ships no hparams file).
"""
from types import SimpleNamespace

import numpy as np
import torch

from transformer_tts_amd import synthetic

_BASE = dict(
    architecture="text-mel", model="Fastspeech2", comment="", save_dir="/tmp/fs2_golden_ckpt",
    train_script="", test_script="", lengths_file="", mean_file=None, var_file=None, spm_model=None,
    mel_dim=80, amp=False, optimizer="Noam", warmup_step=4000, warmup_factor=1.0, max_seqlen=None,
    max_epoch=100, save_per_epoch=50, clip=1.0, loaded_epoch=None, loaded_dir=None,
    encoder_type="transformer", decoder_type="transformer", concat_after_encoder=False,
    concat_after_decoder=False, postnet_pred=True, reduction_rate=1, dropout=0.0, nbins=256,
    f0_min=71.0, f0_max=799.8, energy_min=0.0, energy_max=403.8, pitch_pred=True, energy_pred=True,
    is_multi_speaker=False, different_spk_emb_samespeaker=False)

CONFIGS = {
    # d=32, 2+2 FFT layers, k 3/1: every kernel family at the smallest aligned sizes; ragged batch
    # with zero durations.
    "tiny": dict(hp=dict(vocab_size=40, batch_size=3, d_model_encoder=32, n_layer_encoder=2, n_head_encoder=2,
                         ff_conv_kernel_size_encoder=3, d_model_decoder=32, n_layer_decoder=2, n_head_decoder=2,
                         ff_conv_kernel_size_decoder=1),
                 weight_seed=11, batch=lambda: synthetic.tiny_batch(seed=7, batch_size=3, vocab=40),
                 train_steps=3, start_step=4000),
    # d=64, 1+1 layers, 4 heads, k 9/5: the wide-kernel conv geometry of the benchmark encoder.
    "small": dict(hp=dict(vocab_size=60, batch_size=4, d_model_encoder=64, n_layer_encoder=1, n_head_encoder=4,
                          ff_conv_kernel_size_encoder=9, d_model_decoder=64, n_layer_decoder=1, n_head_decoder=4,
                          ff_conv_kernel_size_decoder=5),
                  weight_seed=12, batch=lambda: synthetic.make_batch(8, 4, l_range=(9, 20), dur_range=(1, 9), vocab=60),
                  train_steps=3, start_step=4000),
    # BASELINE.json configs[1]: scalar anchors + digests only.
    "bench": dict(hp=dict(vocab_size=152, batch_size=48, d_model_encoder=256, n_layer_encoder=4, n_head_encoder=2,
                          ff_conv_kernel_size_encoder=9, d_model_decoder=256, n_layer_decoder=4, n_head_decoder=2,
                          ff_conv_kernel_size_decoder=1),
                  weight_seed=13, batch=lambda: synthetic.benchmark_batch(2024, 48),
                  train_steps=1, start_step=4000),
}

# In-scope options of the FastSpeech2 path beside the defaults, each on the `tiny` geometry:
#   opt_concat    concat_after_encoder / concat_after_decoder = True   (Models/modules.py:38-46,66-67)
#   opt_nopitch   pitch_pred = False, opt_noenergy energy_pred = False (Models/varianceadaptor.py:93-125)
#   opt_ss1       p_scheduled_sampling = 1.0: every utterance's pitch target replaced by the prediction (:99,261-282)
#   opt_ss_half   p_scheduled_sampling = 0.5: torch.rand(B) on the CPU generator decides per utterance; `forward_seed` is set with
#                 torch.manual_seed right before EVERY forward (recipe and tests alike) so that both sides draw the same numbers
def _opt(seed, **hp):
    return dict(hp=dict(CONFIGS["tiny"]["hp"], **hp), weight_seed=seed, batch=CONFIGS["tiny"]["batch"], train_steps=2, start_step=4000,
                forward_seed=1234)


OPTION_CONFIGS = {
    "opt_concat": _opt(31, concat_after_encoder=True, concat_after_decoder=True),
    "opt_nopitch": _opt(32, pitch_pred=False),
    "opt_noenergy": _opt(33, energy_pred=False),
    # the same three for the INFERENCE fixtures (infer_<name>.npz): weight seeds under which every utterance gets predicted durations > 0
    # (the reference's inference branch dies on an utterance whose durations all round to 0) and no duration sits within 2e-3 of a .5 tie
    "inf_concat": dict(_opt(40, concat_after_encoder=True, concat_after_decoder=True), shapes_from="opt_concat"),
    "inf_nopitch": dict(_opt(41, pitch_pred=False), shapes_from="opt_nopitch"),
    "inf_noenergy": dict(_opt(41, energy_pred=False), shapes_from="opt_noenergy"),
    "opt_ss1": _opt(34, p_scheduled_sampling=1.0),
    "opt_ss_half": _opt(35, p_scheduled_sampling=0.5),
}
CONFIGS.update(OPTION_CONFIGS)

# Autoregressive Transformer-TTS (BASELINE.json configs[3]): hparams the reference's train.py /
# Models.transformer.Transformer read on top of the common ones.  Batches are the 8-tuple of datasets_transformer.collate_fn
# (text, mel, pos_text, pos_mel, text_lengths, mel_lengths, stop_token, spk_emb): the first eight fields of the synthetic batch.
_AR = dict(model="Transformer", gst=False, spk_emb_dim=None, spk_emb_architecture="", dropout_prenet=0.0, dropout_postnet=0.0,
           positive_weight=5.0, accum_grad=1, clip=1.0)

AR_CONFIGS = {
    # d=32, 2+2 layers, 2 heads, k 3/1, ragged batch of 3
    "ar_tiny": dict(hp=dict(_AR, vocab_size=40, batch_size=3, d_model_encoder=32, n_layer_encoder=2, n_head_encoder=2,
                            ff_conv_kernel_size_encoder=3, d_model_decoder=32, n_layer_decoder=2, n_head_decoder=2,
                            ff_conv_kernel_size_decoder=1),
                    weight_seed=21, batch=lambda: synthetic.tiny_batch(seed=17, batch_size=3, vocab=40)[:8],
                    train_steps=3, start_step=4000),
    # d_enc 64 -> d_dec 96 (the Linear between encoder and decoder), 1+2 layers, 4 heads, k 9/5, gradient accumulation 2
    "ar_small": dict(hp=dict(_AR, vocab_size=60, batch_size=4, d_model_encoder=64, n_layer_encoder=1, n_head_encoder=4,
                             ff_conv_kernel_size_encoder=9, d_model_decoder=96, n_layer_decoder=2, n_head_decoder=4,
                             ff_conv_kernel_size_decoder=5, accum_grad=2, positive_weight=3.0),
                     weight_seed=22, batch=lambda: synthetic.make_batch(18, 4, l_range=(9, 20), dur_range=(1, 9), vocab=60)[:8],
                     train_steps=4, start_step=4000),
    # reduction rate 2: strided decoder inputs, (B, T/2, 160) outputs regrouped to frame rate
    "ar_r2": dict(hp=dict(_AR, vocab_size=40, batch_size=3, d_model_encoder=32, n_layer_encoder=1, n_head_encoder=2,
                          ff_conv_kernel_size_encoder=3, d_model_decoder=32, n_layer_decoder=1, n_head_decoder=2,
                          ff_conv_kernel_size_decoder=1, reduction_rate=2),
                  weight_seed=23, batch=lambda: even_frames(synthetic.tiny_batch(seed=19, batch_size=3, vocab=40)[:8]),
                  train_steps=2, start_step=4000),
}


def even_frames(batch):
    """crop the padded frame axis to an even length (train.py's reduction-rate slicing and its loss targets only line up
    when T_pad is a multiple of the reduction rate)"""
    text, mel, pos_text, pos_mel, tl, ml, stop, spk = batch
    T = mel.shape[1] - (mel.shape[1] % 2)
    return (text, mel[:, :T].contiguous(), pos_text, pos_mel[:, :T].contiguous(), tl, torch.clamp(ml, max=T),
            stop[:, :T].contiguous(), spk)


def hp_namespace(cfg):
    d = dict(_BASE)
    d.update(cfg["hp"])
    return SimpleNamespace(**d)


def sample_index(n, k=64):
    return np.unique(np.linspace(0, max(n - 1, 0), num=min(n, k)).astype(np.int64))


def digest(t):
    """[sum, sum|x|, l2, n, samples...] in float64: a compact pin for a large tensor."""
    x = torch.as_tensor(t).detach().double().reshape(-1)
    n = x.numel()
    head = [float(x.sum()), float(x.abs().sum()), float((x * x).sum().sqrt()), float(n)]
    return np.asarray(head + x[torch.from_numpy(sample_index(n))].tolist(), np.float64)

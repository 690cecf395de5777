"""Variance adaptor: duration / pitch / energy predictors, length regulator, bucketised embeddings
(reference: Models/varianceadaptor.py:34-259): teacher-forced training branch and the inference branch."""
import random

import numpy as np
import torch
import torch.nn as nn

from .functional import (BucketEmbedAddFunction, LengthRegulatorFunction, Runtime, VariancePredictorFunction, module_params,
                         next_site)


class VariancePredictor(nn.Module):
    """conv3-ReLU-LN-dropout x2 + Linear(filter -> 1) + masked_fill (reference :186-231)."""

    def __init__(self, encoder_hidden_size, variance_predictor_filter_size=256, variance_predictor_kernel_size=3,
                 variance_predictor_dropout=0.5, runtime=None):
        super().__init__()
        assert variance_predictor_kernel_size == 3
        self.input_size, self.filter_size = encoder_hidden_size, variance_predictor_filter_size
        self.kernel, self.dropout = variance_predictor_kernel_size, variance_predictor_dropout
        self.conv1 = nn.Conv1d(self.input_size, self.filter_size, kernel_size=self.kernel, padding=1)
        self.layer_norm1 = nn.LayerNorm(self.filter_size)
        self.conv2 = nn.Conv1d(self.filter_size, self.filter_size, kernel_size=self.kernel, padding=1)
        self.layer_norm2 = nn.LayerNorm(self.filter_size)
        self.linear_layer = nn.Linear(self.filter_size, 1)
        self.site1, self.site2 = next_site(), next_site()
        self.rt = runtime if runtime is not None else Runtime()

    def forward(self, encoder_output, mask, chain=False):
        """chain=True: returns (prediction, alias of encoder_output) -- the caller hands the alias to the next reader of the sequence
        (VariancePredictorFunction: the gradient fan-in then happens inside this predictor's last data-gradient product)."""
        return VariancePredictorFunction.apply(self, encoder_output, mask, bool(chain), *module_params(self))


class LengthRegulator(nn.Module):
    def forward(self, x, duration, max_length=None):
        """training: max_length = mel_mask.shape[2] (second value unused, as in the reference); inference
        (max_length None): pad to the longest utterance and return the frame positions `mel_pos` (B,T) the
        reference's LR builds (Models/varianceadaptor.py:141-156) -- one host read of the longest length."""
        if max_length is not None:
            return LengthRegulatorFunction.apply(x, duration, int(max_length)), None
        duration = duration.long().clamp(min=0)
        lens = duration.sum(dim=1)
        T = int(lens.max())                # the output SHAPE depends on the predicted durations
        assert T > 0, "every predicted duration rounded to zero: nothing to synthesise"
        pos = torch.arange(1, T + 1, device=x.device).unsqueeze(0).expand(x.shape[0], -1)
        return LengthRegulatorFunction.apply(x, duration, T), pos * (pos <= lens.unsqueeze(1))


def scheduled_sampling(predicted, target, p):
    """reference :261-282: with probability p -- ONE draw per utterance, torch.rand(B) on the CPU generator exactly as the reference
    draws it -- the utterance's pitch target is replaced by the predicted pitch before it is bucketised (no gradient flows through
    torch.bucketize).  Not capturable in a hipGraph (host randomness): train_loop launches eagerly when hp.p_scheduled_sampling != 0."""
    if p == 0.0:
        return target
    assert predicted.shape == target.shape
    take = torch.rand(predicted.shape[0]) < p
    if not bool(take.any()):
        return target
    return torch.where(take.to(target.device)[:, None], predicted.detach().to(target.dtype), target)


class VarianceAdaptor(nn.Module):
    def __init__(self, d_model_encoder, n_bins=256, f0_min=71.0, f0_max=795.8, energy_min=0.0, energy_max=315.0,
                 log_offset=1., pitch_pred=True, energy_pred=True, dropout=0.5, use_rnn_length=False, use_pos=False,
                 runtime=None):
        super().__init__()
        assert not use_rnn_length and not use_pos, "use_rnn_length / use_pos are outside the accelerated path"
        self.rt = runtime if runtime is not None else Runtime()
        self.pitch_pred, self.energy_pred = bool(pitch_pred), bool(energy_pred)
        self.duration_predictor = VariancePredictor(d_model_encoder, variance_predictor_dropout=dropout, runtime=self.rt)
        self.length_regulator = LengthRegulator()
        if self.pitch_pred:             # reference :54-57
            self.pitch_predictor = VariancePredictor(d_model_encoder, variance_predictor_dropout=dropout, runtime=self.rt)
            # identical host expressions as the reference (:56,61) so the fp32 boundaries match bit for bit
            self.pitch_bins = torch.exp(torch.linspace(np.log(f0_min), np.log(f0_max), n_bins - 1))
            self.pitch_embedding = nn.Embedding(n_bins, d_model_encoder)
        if self.energy_pred:            # reference :59-62
            self.energy_predictor = VariancePredictor(d_model_encoder, variance_predictor_dropout=dropout, runtime=self.rt)
            self.energy_bins = torch.linspace(energy_min, energy_max, n_bins - 1)
            self.energy_embedding = nn.Embedding(n_bins, d_model_encoder)
        self.log_offset = 1.
        self._bins_dev = {}

    def _bins(self, name, device):
        key = (name, device)
        if key not in self._bins_dev:
            self._bins_dev[key] = getattr(self, name).to(device=device, dtype=torch.float32).contiguous()
        return self._bins_dev[key]

    def pitch_bins_dev(self, device):
        return self._bins("pitch_bins", device)

    def energy_bins_dev(self, device):
        return self._bins("energy_bins", device)

    def _embed_add(self, x, pitch, energy):
        """x (+ pitch_embedding[bucketize(pitch)]) (+ energy_embedding[bucketize(energy)])   (reference :100,116,122-125)"""
        if not (self.pitch_pred or self.energy_pred):
            return x
        tables = ([self.pitch_embedding.weight] if self.pitch_pred else []) + ([self.energy_embedding.weight] if self.energy_pred else [])
        return BucketEmbedAddFunction.apply(self, x, pitch if self.pitch_pred else None, energy if self.energy_pred else None, *tables)

    def forward(self, x, src_mask, mel_mask=None, duration_target=None, pitch_target=None, energy_target=None,
                max_len=None, p_scheduled_sampling=0.0, pitch_perturbation=False, duration_perturbation=False):
        chain = self.training and torch.is_grad_enabled() and x.requires_grad
        if chain:                   # (the alias of x goes on to the length regulator: one consumer of x, no gradient add pass)
            log_duration_prediction, x = self.duration_predictor(x, src_mask, chain=True)
        else:
            log_duration_prediction = self.duration_predictor(x, src_mask)
        if duration_target is None:
            return self._infer(x, log_duration_prediction, max_len, pitch_perturbation, duration_perturbation)
        assert (pitch_target is not None or not self.pitch_pred) and (energy_target is not None or not self.energy_pred)
        if mel_mask is not None:
            max_len = mel_mask.shape[2]
        x, mel_len = self.length_regulator(x, duration_target, max_len)
        pitch_prediction = energy_prediction = None
        if self.pitch_pred:                                                                         # :93-95 / :110
            if chain:
                pitch_prediction, x = self.pitch_predictor(x, mel_mask, chain=True)
            else:
                pitch_prediction = self.pitch_predictor(x, mel_mask)
        if self.energy_pred:                                                                        # :112-114 / :120
            if chain:
                energy_prediction, x = self.energy_predictor(x, mel_mask, chain=True)
            else:
                energy_prediction = self.energy_predictor(x, mel_mask)
        if self.pitch_pred:
            pitch_target = scheduled_sampling(pitch_prediction, pitch_target, p_scheduled_sampling)    # :99
        text_dur_predicted = x
        x = self._embed_add(x, pitch_target, energy_target)
        return x, log_duration_prediction, pitch_prediction, energy_prediction, mel_len, mel_mask, text_dur_predicted

    def _infer(self, x, log_duration_prediction, max_len, pitch_perturbation, duration_perturbation):
        """Inference branch (Models/varianceadaptor.py:74-84,101-109,117-118): predicted durations
        round(exp(log_d) - 1).clamp(0), frame mask from the regulated lengths, variance embeddings of the PREDICTED
        pitch / energy.  Same kernels as training; the only host read is the longest utterance's length."""
        duration_rounded = torch.clamp(torch.round(torch.exp(log_duration_prediction) - self.log_offset), min=0)
        if duration_perturbation:                                                    # :76-81
            rand_weight = random.sample([0.8, 0.9, 1.0, 1.1, 1.2], 1)[0]
            print('dur', rand_weight)
            duration_rounded = torch.round(duration_rounded * rand_weight)
        x, mel_len = self.length_regulator(x, duration_rounded, max_len)             # :82 (mel_len = positions, as there)
        T = x.shape[1]
        ids = torch.arange(0, T, device=x.device).unsqueeze(0).expand(x.shape[0], -1)
        mel_mask = ids <= mel_len                                                    # get_mask_from_lengths, :251-259
        pitch_prediction = energy_prediction = None
        if self.pitch_pred:
            pitch_prediction = self.pitch_predictor(x, mel_mask)
            if pitch_perturbation:                                                   # :103-107
                pitch_prediction = random.sample([0.8, 0.9, 1.0, 1.1, 1.2], 1)[0] * pitch_prediction
        if self.energy_pred:
            energy_prediction = self.energy_predictor(x, mel_mask)
        text_dur_predicted = x
        x = self._embed_add(x, pitch_prediction.contiguous() if self.pitch_pred else None,
                            energy_prediction.contiguous() if self.energy_pred else None)          # :109,118,122-125
        return x, log_duration_prediction, pitch_prediction, energy_prediction, mel_len, mel_mask, text_dur_predicted

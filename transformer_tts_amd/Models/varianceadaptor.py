"""Variance adaptor: duration / pitch / energy predictors, length regulator, bucketised embeddings
(reference: Models/varianceadaptor.py:34-259), teacher-forced training branch."""
import numpy as np
import torch
import torch.nn as nn

from .functional import (BucketEmbedAddFunction, LengthRegulatorFunction, Runtime, VariancePredictorFunction,
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

    def forward(self, encoder_output, mask):
        return VariancePredictorFunction.apply(self, encoder_output, mask, *self.parameters())


class LengthRegulator(nn.Module):
    def forward(self, x, duration, max_length=None):
        assert max_length is not None, "training path: max_length = mel_mask.shape[2]"
        return LengthRegulatorFunction.apply(x, duration, int(max_length)), None


class VarianceAdaptor(nn.Module):
    def __init__(self, d_model_encoder, n_bins=256, f0_min=71.0, f0_max=795.8, energy_min=0.0, energy_max=315.0,
                 log_offset=1., pitch_pred=True, energy_pred=True, dropout=0.5, use_rnn_length=False, use_pos=False,
                 runtime=None):
        super().__init__()
        assert pitch_pred and energy_pred and not use_rnn_length and not use_pos, \
            "only the default pitch+energy configuration is on the accelerated path"
        self.rt = runtime if runtime is not None else Runtime()
        self.pitch_pred, self.energy_pred = pitch_pred, energy_pred
        self.duration_predictor = VariancePredictor(d_model_encoder, variance_predictor_dropout=dropout, runtime=self.rt)
        self.length_regulator = LengthRegulator()
        self.pitch_predictor = VariancePredictor(d_model_encoder, variance_predictor_dropout=dropout, runtime=self.rt)
        # identical host expressions as the reference (:56,61) so the fp32 boundaries match bit for bit
        self.pitch_bins = torch.exp(torch.linspace(np.log(f0_min), np.log(f0_max), n_bins - 1))
        self.pitch_embedding = nn.Embedding(n_bins, d_model_encoder)
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

    def forward(self, x, src_mask, mel_mask=None, duration_target=None, pitch_target=None, energy_target=None,
                max_len=None, p_scheduled_sampling=0.0, pitch_perturbation=False, duration_perturbation=False):
        assert duration_target is not None and pitch_target is not None and energy_target is not None, \
            "inference branch (predicted durations) is a later row of SURVEY section 8(f)"
        assert p_scheduled_sampling == 0.0
        log_duration_prediction = self.duration_predictor(x, src_mask)
        if mel_mask is not None:
            max_len = mel_mask.shape[2]
        x, mel_len = self.length_regulator(x, duration_target, max_len)
        pitch_prediction = self.pitch_predictor(x, mel_mask)
        energy_prediction = self.energy_predictor(x, mel_mask)
        text_dur_predicted = x
        x = BucketEmbedAddFunction.apply(self, x, pitch_target, energy_target, self.pitch_embedding.weight,
                                         self.energy_embedding.weight)
        return x, log_duration_prediction, pitch_prediction, energy_prediction, mel_len, mel_mask, text_dur_predicted

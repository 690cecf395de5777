"""PostConvNet: Linear(d -> mel) + 5 causal Conv1d with BatchNorm/tanh/dropout (reference: Models/postnets.py:13-79)."""
import copy

import torch.nn as nn

from .functional import PostNetFunction, Runtime, module_params, next_site


class PostConvNet(nn.Module):
    def __init__(self, hp, num_hidden, mel_dim, reduction_rate, dropout=0.5, prev_version=True, runtime=None):
        super().__init__()
        # prev_version=False (Models/transformer.py:92): no `out` Linear; the reference's forward then RETURNS ITS INPUT
        # (postnets.py:76-79) -- see functional_ar.postnet_update_statistics
        assert reduction_rate == 1 or not prev_version
        self.prev_version = prev_version
        self.dropout = dropout
        self.conv1 = nn.Conv1d(mel_dim * reduction_rate, num_hidden, kernel_size=5, padding=4)
        # three COPIES of one freshly initialised module, as the reference's clones() makes them (postnets.py:10-11,32-35):
        # one draw from the RNG stream, identical starting weights
        proto = nn.Conv1d(num_hidden, num_hidden, kernel_size=5, padding=4)
        self.conv_list = nn.ModuleList([copy.deepcopy(proto) for _ in range(3)])
        self.conv2 = nn.Conv1d(num_hidden, mel_dim * reduction_rate, kernel_size=5, padding=4)
        if prev_version:
            self.out = nn.Linear(num_hidden, mel_dim * reduction_rate)
        proto_bn = nn.BatchNorm1d(num_hidden)
        self.batch_norm_list = nn.ModuleList([copy.deepcopy(proto_bn) for _ in range(3)])
        self.pre_batchnorm = nn.BatchNorm1d(num_hidden)
        self.sites = [next_site() for _ in range(4)]
        self.rt = runtime if runtime is not None else Runtime()

    def forward(self, input_, mask=None):
        if not self.prev_version:
            from .functional_ar import postnet_update_statistics
            postnet_update_statistics(self, input_)
            return input_
        return PostNetFunction.apply(self, input_, *module_params(self))

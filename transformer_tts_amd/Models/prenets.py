"""DecoderPreNet parameter container (reference: Models/prenets.py:8-44, output_type None): Linear -> ReLU -> Dropout ->
Linear -> ReLU -> Dropout with the reference's attribute names (``layer.fc1`` / ``layer.fc2`` state_dict keys); the
arithmetic runs in functional_ar.DecoderStackFunction."""
from collections import OrderedDict

import torch.nn as nn


class DecoderPreNet(nn.Module):
    def __init__(self, input_size, output_size, hidden_size=256, p=0.5, output_type=None):
        super().__init__()
        assert not output_type, "the embedding pre-net (output_type) is outside the accelerated path"
        assert input_size % 8 == 0 and hidden_size % 8 == 0 and output_size % 8 == 0
        self.input_size, self.output_size, self.hidden_size, self.p = input_size, output_size, hidden_size, p
        self.layer = nn.Sequential(OrderedDict([
            ("fc1", nn.Linear(input_size, hidden_size)), ("relu1", nn.ReLU()), ("dropout1", nn.Dropout(p)),
            ("fc2", nn.Linear(hidden_size, output_size)), ("relu2", nn.ReLU()), ("dropout2", nn.Dropout(p))]))

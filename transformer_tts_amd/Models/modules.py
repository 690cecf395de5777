"""Parameter containers of the FFT block (reference: Models/modules.py:23-111).

These classes keep the reference's attribute names -- hence its state_dict keys -- and own the
parameters; the arithmetic lives in HIP kernels sequenced by ``functional.EncoderStackFunction``.
"""
import numpy as np
import torch
import torch.nn as nn

from .functional import next_site


def positional_table(d_model, max_seq_len=5000):
    """Table of the reference's PositionalEncoder (Models/modules.py:97-105): for even i,
    pe[pos,i] = sin(pos / 10000^(2i/d)), pe[pos,i+1] = cos(pos / 10000^(2(i+1)/d)); float64 math, fp32 store."""
    pos = np.arange(max_seq_len, dtype=np.float64)[:, None]
    i = np.arange(0, d_model, 2, dtype=np.float64)[None, :]
    pe = np.zeros((max_seq_len, d_model), np.float64)
    pe[:, 0::2] = np.sin(pos / (10000.0 ** ((2.0 * i) / d_model)))
    pe[:, 1::2] = np.cos(pos / (10000.0 ** ((2.0 * (i + 1.0)) / d_model)))
    return torch.from_numpy(pe.astype(np.float32))


class PositionalEncoder(nn.Module):
    def __init__(self, d_model, max_seq_len=5000, dropout=0.1):
        super().__init__()
        self.d_model = d_model
        self.alpha = nn.Parameter(torch.ones(1))
        self.pe = positional_table(d_model, max_seq_len).unsqueeze(0)   # plain attribute, not a buffer
        self.site = next_site()
        self._dev = {}

    def table(self, device):
        t = self._dev.get(device)
        if t is None:
            t = self.pe[0].to(device).contiguous()
            self._dev = {device: t}
        return t


class MultiHeadAttention(nn.Module):
    """q/v/k/out projections registered in the reference's order (Models/modules.py:32-41); concat_after (:38-41): the output
    projection reads cat(query input, attention context) -- Linear(2d -> d)."""

    def __init__(self, heads, q_dim, k_dim, v_dim, d_model, dropout=0.1, concat_after=False):
        super().__init__()
        self.d_model, self.d_k, self.h = d_model, d_model // heads, heads
        self.q_linear = nn.Linear(q_dim, d_model)
        self.v_linear = nn.Linear(k_dim, d_model)
        self.k_linear = nn.Linear(v_dim, d_model)
        self.dropout = dropout
        self.concat_after = bool(concat_after)
        self.out = nn.Linear(2 * d_model if self.concat_after else d_model, d_model)


class FeedForward(nn.Module):
    def __init__(self, d_model, ff_conv_kernel_size, dropout=0.1):
        super().__init__()
        k = ff_conv_kernel_size
        self.f_1 = nn.Conv1d(d_model, d_model * 4, kernel_size=k, padding=int(k / 2))
        self.f_2 = nn.Conv1d(d_model * 4, d_model, kernel_size=k, padding=int(k / 2))
        self.layer_norm = nn.LayerNorm(d_model)

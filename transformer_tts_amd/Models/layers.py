"""EncoderLayer parameter container (reference: Models/layers.py:8-41, single-speaker branch)."""
import torch.nn as nn

from .functional import next_site
from .modules import FeedForward, MultiHeadAttention


class EncoderLayer(nn.Module):
    def __init__(self, d_model, heads, ff_conv_kernel_size, dropout=0.1, concat_after=False, multi_speaker=False,
                 spk_emb_dim=None):
        super().__init__()
        assert not multi_speaker, "multi-speaker conditioning is outside the accelerated path"
        self.norm_1 = nn.LayerNorm(d_model)
        self.norm_2 = nn.LayerNorm(d_model)
        self.attn = MultiHeadAttention(heads, d_model, d_model, d_model, d_model, dropout=dropout,
                                       concat_after=concat_after)
        self.ff = FeedForward(d_model, ff_conv_kernel_size, dropout=dropout)
        self.multi_speaker = multi_speaker
        # dropout call sites: attention probabilities, dropout_1, FeedForward.dropout, dropout_2
        self.site_attn, self.site_res1, self.site_ffn, self.site_res2 = (next_site() for _ in range(4))


class DecoderLayer(nn.Module):
    """reference: Models/layers.py:84-125 (single-speaker branch): three pre-LayerNorms, masked self-attention (attn_1),
    encoder-decoder attention (attn_2), conv FeedForward."""

    def __init__(self, d_model, heads, ff_conv_kernel_size, dropout=0.1, concat_after=False, multi_speaker=False, spk_emb_dim=None):
        super().__init__()
        assert not multi_speaker, "multi-speaker conditioning is outside the accelerated path"
        assert not concat_after, "concat_after_decoder of the autoregressive decoder is outside the accelerated path (the FFT stacks take it)"
        self.norm_1 = nn.LayerNorm(d_model)
        self.norm_2 = nn.LayerNorm(d_model)
        self.norm_3 = nn.LayerNorm(d_model)
        self.attn_1 = MultiHeadAttention(heads, d_model, d_model, d_model, d_model, dropout=dropout, concat_after=concat_after)
        self.attn_2 = MultiHeadAttention(heads, d_model, d_model, d_model, d_model, dropout=dropout, concat_after=concat_after)
        self.ff = FeedForward(d_model, ff_conv_kernel_size, dropout=dropout)
        self.multi_speaker = multi_speaker
        # dropout call sites: attn_1 probabilities, dropout_1, attn_2 probabilities, dropout_2, FeedForward.dropout, dropout_3
        self.site_attn1, self.site_res1, self.site_attn2, self.site_res2, self.site_ffn, self.site_res3 = (next_site() for _ in range(6))

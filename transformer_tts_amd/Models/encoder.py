"""``Encoder`` -- the FFT stack used as FastSpeech2's encoder and decoder (reference: Models/encoder.py:31-112)."""
import torch.nn as nn

from .functional import EncoderStackFunction, Runtime, module_params
from .layers import EncoderLayer
from .modules import PositionalEncoder


class Encoder(nn.Module):
    def __init__(self, vocab_size, d_model, N, heads, ff_conv_kernel_size, concat_after_encoder, dropout,
                 multi_speaker=False, spk_emb_dim=None, embedding=True, accent_emb=False, spk_emb_layer=None,
                 gender_emb=False, intermediate_layers_out=None, runtime=None):
        super().__init__()
        assert not (multi_speaker or accent_emb or gender_emb or intermediate_layers_out or spk_emb_layer), \
            "speaker / accent / gender / intermediate outputs are outside the accelerated path"
        assert d_model % heads == 0 and (d_model // heads) % 8 == 0 and d_model % 8 == 0, \
            "d_model and d_model/heads must be multiples of 8 (16-byte MFMA fragments)"
        self.N, self.heads, self.d_model, self.dropout = N, heads, d_model, dropout
        self.embedding = embedding
        if embedding:
            self.embed = nn.Embedding(vocab_size, d_model, padding_idx=0)
        else:
            assert vocab_size % 8 == 0
            self.embed = nn.Linear(vocab_size, d_model)
        self.pe = PositionalEncoder(d_model, dropout=dropout)
        self.layers = nn.ModuleList([EncoderLayer(d_model, heads, ff_conv_kernel_size, dropout=dropout,
                                                  concat_after=concat_after_encoder) for _ in range(N)])
        self.norm = nn.LayerNorm(d_model)
        self.rt = runtime if runtime is not None else Runtime()

    def forward(self, src, mask, spkr_emb=None, accent=None, gender_id=None, attn_detach=True):
        """src: (B,t) int64 ids (embedding=True) or (B,t,vocab) activations; mask: (B,1,t) bool key mask.
        Returns (LayerNorm(x_N) in the compute dtype, attention maps (B,N,H,t,t) -- post-dropout as in the
        reference, Models/modules.py:19-21)."""
        out, attn = EncoderStackFunction.apply(self, src, mask, *module_params(self))
        return out, (attn if self.rt.return_attn else None)

"""``Decoder`` of the autoregressive Transformer-TTS (reference: Models/decoder.py:29-56): pre-net, positional encoding,
N decoder layers (masked self-attention, encoder-decoder attention, conv FFN), final LayerNorm -- parameter container with
the reference's attribute names; computed by functional_ar.DecoderStackFunction."""
import torch.nn as nn

from .functional import Runtime, next_site
from .functional_ar import DecoderStackFunction
from .layers import DecoderLayer
from .modules import PositionalEncoder
from .prenets import DecoderPreNet


class Decoder(nn.Module):
    def __init__(self, vocab_size, d_model, N, heads, ff_conv_kernel_size, concat_after_decoder, dropout, dropout_prenet=0.5,
                 multi_speaker=False, spk_emb_dim=None, output_type=None, runtime=None):
        super().__init__()
        assert not multi_speaker and not output_type, "speaker conditioning / discrete outputs are outside the accelerated path"
        assert d_model % heads == 0 and (d_model // heads) % 8 == 0 and d_model % 8 == 0
        self.N, self.heads, self.d_model = N, heads, d_model
        self.dropout, self.dropout_prenet = dropout, dropout_prenet
        self.output_type = output_type
        self.decoder_prenet = DecoderPreNet(vocab_size, d_model, p=dropout_prenet, output_type=output_type)
        self.pe = PositionalEncoder(d_model, dropout=dropout)
        # nn.Sequential of the layers, as the reference's repeat() builds it (state_dict keys layers.<i>.*)
        self.layers = nn.Sequential(*[DecoderLayer(d_model, heads, ff_conv_kernel_size, dropout, concat_after_decoder)
                                      for _ in range(N)])
        self.norm = nn.LayerNorm(d_model)
        self.site_pre1, self.site_pre2 = next_site(), next_site()
        self.rt = runtime if runtime is not None else Runtime()

    def forward(self, trg, e_outputs, src_mask, trg_key_mask, spk_emb=None, attn_detach=True):
        """trg (B,T,mel) fp32 teacher-forcing frames, e_outputs (B,L,d), src_mask (B,1,L) bool, trg_key_mask (B,T) bool key
        padding of the decoder frames (the no-peak mask is applied inside the softmax kernel).  Returns LayerNorm(x_N) and the
        attention maps (B,N,H,T,T), (B,N,H,T,L), post-dropout as in the reference."""
        out, a1, a2 = DecoderStackFunction.apply(self, trg, e_outputs, src_mask, trg_key_mask, *self.parameters())
        keep = self.rt.return_attn
        return out, (a1 if keep else None), (a2 if keep else None)

"""``Transformer`` -- drop-in module surface of the reference's autoregressive Transformer-TTS (Models/transformer.py:15-118)
for its default branch (transformer encoder + transformer decoder, single speaker, no GST), computed by hand-written gfx950
kernels.  Same constructor / forward signature, same attribute names (state_dict keys) as the reference."""
import torch
import torch.nn as nn

from .. import ops

from .decoder import Decoder
from .encoder import Encoder
from .functional import Runtime
from .functional_ar import LinearFunction
from .postnets import PostConvNet


class Transformer(nn.Module):
    def __init__(self, hp, src_vocab, trg_vocab, d_model_encoder, N_e, n_head_encoder, ff_conv_kernel_size_encoder,
                 concat_after_encoder, d_model_decoder, N_d, n_head_decoder, ff_conv_kernel_size_decoder, concat_after_decoder,
                 reduction_rate, dropout, dropout_prenet=0.5, dropout_postnet=0.5, multi_speaker=False, spk_emb_dim=None,
                 spk_emb_architecture=None, output_type=None, decoder_type='transformer'):
        super().__init__()
        unsupported = [n for n, v in (("gst", getattr(hp, "gst", False)), ("multi_speaker", multi_speaker),
                                      ("spk_emb_architecture", bool(spk_emb_architecture)), ("output_type", bool(output_type))) if v]
        if unsupported or decoder_type.lower() != 'transformer' or getattr(hp, "encoder_type", "transformer").lower() != "transformer":
            raise NotImplementedError(f"options outside the accelerated Transformer-TTS path: {unsupported} (SURVEY.md section 8f N2)")
        self.gst, self.spk_emb_vers, self.decoder_type = False, 1, decoder_type
        self.reduction_rate = reduction_rate
        amp = bool(getattr(hp, "amp", False))
        self.rt = Runtime(torch.bfloat16 if amp else torch.float32, seed=int(getattr(hp, "seed", 1234)))
        self.rt.return_attn = bool(getattr(hp, "return_attn", True))
        self.rt.overlap_wgrad = bool(getattr(hp, "overlap_wgrad", False))
        self.rt.fp8 = amp and bool(getattr(hp, "fp8", False))     # BASELINE.json configs[4]: fp8 MFMA GEMMs
        self.encoder = Encoder(src_vocab, d_model_encoder, N_e, n_head_encoder, ff_conv_kernel_size_encoder, concat_after_encoder,
                               dropout=dropout, runtime=self.rt)
        self.linear = nn.Linear(d_model_encoder, d_model_decoder) if d_model_encoder != d_model_decoder else None
        self.decoder = Decoder(trg_vocab, d_model_decoder, N_d, n_head_decoder, ff_conv_kernel_size_decoder, concat_after_decoder,
                               dropout=dropout, dropout_prenet=dropout_prenet, output_type=output_type, runtime=self.rt)
        self.out = nn.Linear(d_model_decoder, trg_vocab * reduction_rate)
        self.stop_token = nn.Linear(d_model_decoder, reduction_rate)
        self.postnet = PostConvNet(hp, d_model_decoder, trg_vocab, reduction_rate, dropout_postnet, prev_version=False, runtime=self.rt)

    def forward(self, src, trg, src_mask, trg_mask, spkr_emb=None, training=True, ref_mel=None):
        """src (B,L) ids, trg (B,T,mel) teacher-forcing frames, src_mask (B,1,L), trg_mask (B,T,T) = frame padding & no-peak
        mask as the reference's create_masks builds it (train.py:38-58).  Returns the reference's 6-tuple
        (outputs_prenet, outputs_postnet, stop_token, attn_enc, attn_dec_dec, attn_dec_enc)."""
        assert spkr_emb is None, "speaker embeddings are outside the accelerated path"
        ops.FP8_MODE["on"] = self.rt.fp8
        if self.rt.fp8:
            ops.fp8_begin_step(src.device)
        self.rt.refresh(self)
        e_outputs, attn_enc = self.encoder(src, src_mask)
        if self.linear is not None:
            e_outputs = LinearFunction.apply(self.linear, self.rt, e_outputs, False, *self.linear.parameters())
        # the kernels apply the no-peak part themselves: only the key padding of the frames is passed on.  The last query row
        # of pad & no-peak is the padding mask itself.
        trg_key_mask = trg_mask[:, -1, :] if trg_mask.dim() == 3 else trg_mask
        if getattr(trg_mask, "_fs2_kinfo", None) is not None:       # (train.create_masks: row bounds / ranking of the frame padding mask)
            trg_key_mask._fs2_kinfo = trg_mask._fs2_kinfo
        if getattr(self.rt, "check_masks", False):
            T = trg_mask.shape[-1]
            tri = torch.tril(torch.ones(T, T, dtype=torch.bool, device=trg_mask.device))
            assert torch.equal(trg_mask, trg_key_mask.unsqueeze(1) & tri), "trg_mask must be key padding & no-peak mask"
        d_output, attn_dec_dec, attn_dec_enc = self.decoder(trg, e_outputs, src_mask, trg_key_mask)
        outputs_prenet = LinearFunction.apply(self.out, self.rt, d_output, True, *self.out.parameters())
        outputs_postnet = self.postnet(outputs_prenet)        # prev_version=False: returns its input (Models/postnets.py:76-79)
        stop_token = LinearFunction.apply(self.stop_token, self.rt, d_output, True, *self.stop_token.parameters()).squeeze(2)
        return outputs_prenet, outputs_postnet, stop_token, attn_enc, attn_dec_dec, attn_dec_enc

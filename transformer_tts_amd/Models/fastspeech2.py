"""FastSpeech2 -- drop-in module surface of the reference's Models/fastspeech2.py (:19-241) for the
default training branch and the inference branch (transformer encoder/decoder, postnet_pred, no
speaker/SQ-VAE/hop/fix_mask/debug),
computed by hand-written gfx950 kernels."""
import torch
import torch.nn as nn

from .. import ops

from .encoder import Encoder
from .functional import Runtime
from .postnets import PostConvNet
from .varianceadaptor import VarianceAdaptor


class FastSpeech2(nn.Module):
    def __init__(self, hp, src_vocab, trg_vocab, d_model_encoder, N_e, n_head_encoder, ff_conv_kernel_size_encoder,
                 concat_after_encoder, d_model_decoder, N_d, n_head_decoder, ff_conv_kernel_size_decoder,
                 concat_after_decoder, reduction_rate, dropout, dropout_postnet, dropout_variance_adaptor, n_bins,
                 f0_min, f0_max, energy_min, energy_max, pitch_pred=True, energy_pred=True, accent_emb=False,
                 output_type=None, num_group=None, log_offset=1., multi_speaker=False, spk_emb_dim=None,
                 spk_emb_architecture=None, debug=False):
        super().__init__()
        self.hp = hp
        self.spk_emb_architecture = spk_emb_architecture or ''
        unsupported = [n for n, v in (("multi_speaker", multi_speaker), ("accent_emb", accent_emb), ("debug", debug),
                                      ("use_sq_vae", getattr(hp, "use_sq_vae", False)),
                                      ("use_hop", getattr(hp, "use_hop", False)),
                                      ("use_rnn_length", getattr(hp, "use_rnn_length", False)),
                                      ("use_pos", getattr(hp, "use_pos", False)),
                                      ("spk_emb_architecture", bool(self.spk_emb_architecture))) if v]
        if unsupported or getattr(hp, "encoder_type", "transformer").lower() != "transformer" \
                or getattr(hp, "decoder_type", "transformer").lower() != "transformer" or not hp.postnet_pred:
            raise NotImplementedError(f"options outside the accelerated FastSpeech2 path: {unsupported} "
                                      "(SURVEY.md section 8 scope)")
        assert d_model_encoder == d_model_decoder, "the variance adaptor and decoder are sized by d_model_encoder"
        amp = bool(getattr(hp, "amp", False))
        self.rt = Runtime(torch.bfloat16 if amp else torch.float32, seed=int(getattr(hp, "seed", 1234)))
        self.rt.return_attn = bool(getattr(hp, "return_attn", True))
        self.rt.overlap_wgrad = bool(getattr(hp, "overlap_wgrad", False))
        self.rt.fp8 = amp and bool(getattr(hp, "fp8", False))     # BASELINE.json configs[4]: fp8 MFMA GEMMs
        self.encoder = Encoder(src_vocab, d_model_encoder, N_e, n_head_encoder, ff_conv_kernel_size_encoder,
                               concat_after_encoder, dropout, runtime=self.rt)
        self.use_sq_vae = False
        self.variance_adaptor = VarianceAdaptor(d_model_encoder, n_bins, f0_min, f0_max, energy_min, energy_max,
                                                log_offset, pitch_pred, energy_pred, dropout=dropout_variance_adaptor,
                                                runtime=self.rt)
        self.decoder = Encoder(d_model_encoder, d_model_decoder, N_d, n_head_decoder, ff_conv_kernel_size_decoder,
                               concat_after_decoder, dropout, embedding=False, runtime=self.rt)
        self.postnet = PostConvNet(hp=hp, num_hidden=d_model_decoder, mel_dim=trg_vocab, reduction_rate=reduction_rate,
                                   dropout=dropout_postnet, runtime=self.rt)
        self.debug = False

    def forward(self, src, src_mask, mel_mask=None, d_target=None, p_target=None, e_target=None, accent=None,
                spkr_emb=None, fix_mask=None, spkr_emb_post=None, temperature=None, pitch_perturbation=False,
                duration_perturbation=False, hop_size=None):
        assert (self.training and not pitch_perturbation) or (not self.training)
        if fix_mask is not None:
            raise NotImplementedError("fix_mask is outside the accelerated path (SURVEY section 8)")
        ops.FP8_MODE["on"] = self.rt.fp8
        if self.rt.fp8:
            ops.fp8_begin_step(src.device)
        self.rt.refresh(self)           # all weight shadows in one launch (no-op if the weights did not change)
        e_outputs, attn_enc = self.encoder(src, src_mask)
        if d_target is not None:
            variance_adaptor_output, log_d_prediction, p_prediction, e_prediction, _, _, text_dur_predicted = \
                self.variance_adaptor(e_outputs, src_mask, mel_mask, d_target, p_target, e_target,
                                      p_scheduled_sampling=getattr(self.hp, "p_scheduled_sampling", 0.0))
        else:       # inference (Models/fastspeech2.py:174-176): durations, pitch, energy and the frame mask are predicted
            assert not self.training, "the inference branch (d_target=None) runs in eval() mode, as test_fastspeech2.py does"
            variance_adaptor_output, log_d_prediction, p_prediction, e_prediction, mel_len, mel_mask, text_dur_predicted = \
                self.variance_adaptor(e_outputs, src_mask, mel_mask, None, None, None, p_scheduled_sampling=0.0,
                                      pitch_perturbation=pitch_perturbation, duration_perturbation=duration_perturbation)
        d_output, attn_dec = self.decoder(variance_adaptor_output, mel_mask)
        outputs_prenet, outputs_postnet = self.postnet(d_output)
        return (outputs_prenet, outputs_postnet, log_d_prediction, p_prediction, e_prediction, variance_adaptor_output,
                text_dur_predicted, attn_enc, attn_dec, None, None, None, None, None)

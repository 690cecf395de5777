"""Oracle restatement of the reference FastSpeech2 module tree (default branch only).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Plain eager PyTorch, same state_dict keys and
registration order as the reference, so the fixture weights load into it unchanged.
Citations are file:line of /root/reference (syoamakase/Transformer_TTS).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def positional_table(d_model, max_seq_len=5000):
    """Models/modules.py:97-105: pe[pos,i]=sin(pos/10000^(2i/d)), pe[pos,i+1]=cos(pos/10000^(2(i+1)/d))
    for even i (the exponent uses 2*i, not i), evaluated in Python floats and stored as fp32."""
    pos = np.arange(max_seq_len, dtype=np.float64)[:, None]
    i = np.arange(0, d_model, 2, dtype=np.float64)[None, :]
    pe = np.zeros((max_seq_len, d_model), np.float64)
    pe[:, 0::2] = np.sin(pos / (10000.0 ** ((2.0 * i) / d_model)))
    pe[:, 1::2] = np.cos(pos / (10000.0 ** ((2.0 * (i + 1.0)) / d_model)))
    return torch.from_numpy(pe.astype(np.float32))


class PositionalEncoder(nn.Module):
    """Models/modules.py:90-111: x + alpha * pe[:t], then dropout; pe is not a buffer."""

    def __init__(self, d_model, max_seq_len=5000, dropout=0.1):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1))
        self.drop = dropout
        self.pe = positional_table(d_model, max_seq_len)

    def forward(self, x):
        pe = self.pe[: x.shape[1]].to(x.dtype)
        return F.dropout(x + self.alpha * pe, self.drop, self.training)


def attention(q, k, v, d_k, key_mask, p_drop):
    """Models/modules.py:7-21.  key_mask (B,1,t) bool; -1e4 fill on masked KEYS only; the dropout on
    the probabilities is F.dropout with its default training=True, i.e. active even in eval()."""
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d_k)
    s = s.masked_fill(key_mask.unsqueeze(1) == 0, -1e4)
    p = torch.softmax(s, dim=-1)
    p = F.dropout(p, p_drop, True)
    return torch.matmul(p, v), p


class MultiHeadAttention(nn.Module):
    """Models/modules.py:23-70.  Registration order q, v, k, out (:32-41); concat_after (:38-41,45-46,66-67): the output
    projection is Linear(2d -> d) on cat(query input, attention context)."""

    def __init__(self, heads, d_model, dropout, concat_after=False):
        super().__init__()
        self.h, self.d_k, self.p = heads, d_model // heads, dropout
        self.q_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.concat_after = concat_after
        self.out = nn.Linear(2 * d_model if concat_after else d_model, d_model)

    def forward(self, x, key_mask):
        b, t, d = x.shape
        split = lambda y: y.view(b, t, self.h, self.d_k).transpose(1, 2)
        o, p = attention(split(self.q_linear(x)), split(self.k_linear(x)), split(self.v_linear(x)),
                         self.d_k, key_mask, self.p)
        concat = o.transpose(1, 2).reshape(b, t, d)
        if self.concat_after:
            concat = torch.cat((x, concat), dim=-1)                    # :66-67
        return self.out(concat), p


class FeedForward(nn.Module):
    """Models/modules.py:72-88: LN(dropout(conv2(relu(conv1(x))) + x)) -- its own residual and LN."""

    def __init__(self, d_model, k, dropout):
        super().__init__()
        self.f_1 = nn.Conv1d(d_model, 4 * d_model, k, padding=k // 2)
        self.f_2 = nn.Conv1d(4 * d_model, d_model, k, padding=k // 2)
        self.layer_norm = nn.LayerNorm(d_model)
        self.p = dropout

    def forward(self, x):
        y = self.f_2(F.relu(self.f_1(x.transpose(1, 2)))).transpose(1, 2)
        return self.layer_norm(F.dropout(y + x, self.p, self.training))


class EncoderLayer(nn.Module):
    """Models/layers.py:8-41 (single speaker): pre-LN attention and FFN blocks with outer residuals."""

    def __init__(self, d_model, heads, k, dropout, concat_after=False):
        super().__init__()
        self.norm_1 = nn.LayerNorm(d_model)
        self.norm_2 = nn.LayerNorm(d_model)
        self.attn = MultiHeadAttention(heads, d_model, dropout, concat_after)
        self.ff = FeedForward(d_model, k, dropout)
        self.p = dropout

    def forward(self, x, key_mask):
        a, p = self.attn(self.norm_1(x), key_mask)
        x = x + F.dropout(a, self.p, self.training)
        x = x + F.dropout(self.ff(self.norm_2(x)), self.p, self.training)
        return x, p


class Encoder(nn.Module):
    """Models/encoder.py:31-112: embed (Embedding pad 0 | Linear) -> PE -> N layers -> LayerNorm;
    attention maps of all layers stacked to (B,N,H,t,t) (:97,105)."""

    def __init__(self, vocab, d_model, N, heads, k, dropout, embedding=True, concat_after=False):
        super().__init__()
        self.embed = nn.Embedding(vocab, d_model, padding_idx=0) if embedding else nn.Linear(vocab, d_model)
        self.pe = PositionalEncoder(d_model, dropout=dropout)
        self.layers = nn.ModuleList([EncoderLayer(d_model, heads, k, dropout, concat_after) for _ in range(N)])
        self.norm = nn.LayerNorm(d_model)

    def forward(self, src, key_mask):
        x = self.pe(self.embed(src))
        attns = []
        for layer in self.layers:
            x, p = layer(x, key_mask)
            attns.append(p)
        return self.norm(x), torch.stack(attns, dim=1)


class VariancePredictor(nn.Module):
    """Models/varianceadaptor.py:186-231: conv3-ReLU-LN-drop, conv3-ReLU-LN-drop, Linear(256->1),
    masked_fill(mask==0, 0)."""

    def __init__(self, d_in, filt=256, k=3, dropout=0.5):
        super().__init__()
        self.conv1 = nn.Conv1d(d_in, filt, k, padding=1)
        self.layer_norm1 = nn.LayerNorm(filt)
        self.conv2 = nn.Conv1d(filt, filt, k, padding=1)
        self.layer_norm2 = nn.LayerNorm(filt)
        self.linear_layer = nn.Linear(filt, 1)
        self.p = dropout

    def forward(self, x, mask):
        x = F.relu(self.conv1(x.transpose(1, 2))).transpose(1, 2)
        x = F.dropout(self.layer_norm1(x), self.p, self.training)
        x = F.relu(self.conv2(x.transpose(1, 2))).transpose(1, 2)
        x = F.dropout(self.layer_norm2(x), self.p, self.training)
        out = self.linear_layer(x).squeeze(-1)
        return out.masked_fill(mask.squeeze(1) == 0, 0.0)


def length_regulate(x, dur, max_len):
    """Models/varianceadaptor.py:141-184,233-249: repeat phoneme vector i dur[i] times, then zero-pad
    (or crop: F.pad with a negative amount) each utterance to max_len (None: the longest of the batch)."""
    reps = [torch.repeat_interleave(x[b], dur[b].long().clamp(min=0), dim=0) for b in range(x.shape[0])]
    if max_len is None:
        max_len = max(r.shape[0] for r in reps)
    out = x.new_zeros(x.shape[0], max_len, x.shape[2])
    for b, rep in enumerate(reps):
        rep = rep[:max_len]
        out[b, : rep.shape[0]] = rep
    return out


def mel_positions(dur):
    """the `mel_pos` the inference branch gets back from LengthRegulator.LR (:141-156, max_length=None): frame
    positions 1..len_b, zero-padded to the longest utterance"""
    lens = dur.long().clamp(min=0).sum(dim=1)
    T = int(lens.max())
    pos = torch.arange(1, T + 1, device=dur.device).unsqueeze(0).expand(dur.shape[0], -1)
    return pos * (pos <= lens.unsqueeze(1))


def get_mask_from_lengths(lengths):
    """Models/varianceadaptor.py:251-259 as the inference branch calls it (:84) -- with the (B,T) position tensor, not
    lengths: mask[b][j] = j <= mel_pos[b][j], i.e. True on valid frames (and on j = 0)."""
    max_len = int(torch.max(lengths))
    ids = torch.arange(0, max_len, device=lengths.device).unsqueeze(0).expand(lengths.shape[0], -1)
    return ids <= lengths


def scheduled_sampling(predicted, target, p):
    """Models/varianceadaptor.py:261-282: with probability p (one host draw per utterance, torch.rand(B) on the CPU generator) the
    utterance's pitch target is replaced by the predicted pitch before it is bucketised (no gradient flows: bucketize)."""
    if p == 0.0:
        return target
    assert predicted.shape == target.shape
    result = target.clone()
    rand_vals = torch.rand(predicted.shape[0])
    for i in range(predicted.shape[0]):
        if rand_vals[i] < p:
            result[i] = predicted[i].detach().to(result.dtype)
    return result


class VarianceAdaptor(nn.Module):
    """Models/varianceadaptor.py:34-129, teacher-forced branch (duration/pitch/energy targets given); pitch_pred / energy_pred
    False (:93-125): that predictor, its embedding table and its term of the sum do not exist, its prediction is None."""

    def __init__(self, d_model, n_bins, f0_min, f0_max, energy_min, energy_max, dropout, pitch_pred=True, energy_pred=True):
        super().__init__()
        self.pitch_pred, self.energy_pred = pitch_pred, energy_pred
        self.duration_predictor = VariancePredictor(d_model, dropout=dropout)
        if pitch_pred:
            self.pitch_predictor = VariancePredictor(d_model, dropout=dropout)
            # :56,61 -- fp32 boundaries, plain attributes
            self.pitch_bins = torch.exp(torch.linspace(np.log(f0_min), np.log(f0_max), n_bins - 1))
            self.pitch_embedding = nn.Embedding(n_bins, d_model)
        if energy_pred:
            self.energy_predictor = VariancePredictor(d_model, dropout=dropout)
            self.energy_bins = torch.linspace(energy_min, energy_max, n_bins - 1)
            self.energy_embedding = nn.Embedding(n_bins, d_model)

    def forward(self, x, src_mask, mel_mask, d_target, p_target, e_target, p_scheduled_sampling=0.0):
        log_d = self.duration_predictor(x, src_mask)                       # :69
        if d_target is None:
            return self.infer(x, log_d)
        x = length_regulate(x, d_target, mel_mask.shape[2])                # :71-73
        out, p, e = x, None, None
        if self.pitch_pred:
            p = self.pitch_predictor(x, mel_mask)                          # :95
            p_t = scheduled_sampling(p, p_target, p_scheduled_sampling)    # :99
            out = out + self.pitch_embedding(torch.bucketize(p_t, self.pitch_bins.to(p_t.dtype)))   # :100,123
        if self.energy_pred:
            e = self.energy_predictor(x, mel_mask)                         # :114
            out = out + self.energy_embedding(torch.bucketize(e_target, self.energy_bins.to(e_target.dtype)))  # :116,125
        return out, log_d, p, e, x                                         # :122-129

    def infer(self, x, log_d):
        """Inference branch (:74-84,101-109,117-118; no perturbation): predicted durations, the variance embeddings
        of the PREDICTED pitch / energy.  Returns the training tuple + (mel_pos, mel_mask)."""
        dur = torch.clamp(torch.round(torch.exp(log_d) - 1.0), min=0)      # :75 (log_offset = 1)
        x = length_regulate(x, dur, None)                                  # :82
        mel_pos = mel_positions(dur)
        mel_mask = get_mask_from_lengths(mel_pos)                          # :84
        out, p, e = x, None, None
        if self.pitch_pred:
            p = self.pitch_predictor(x, mel_mask)                          # :95
            out = out + self.pitch_embedding(torch.bucketize(p, self.pitch_bins.to(p.dtype)))      # :109,123
        if self.energy_pred:
            e = self.energy_predictor(x, mel_mask)                         # :114
            out = out + self.energy_embedding(torch.bucketize(e, self.energy_bins.to(e.dtype)))    # :118,125
        return out, log_d, p, e, x, mel_pos, mel_mask


class PostConvNet(nn.Module):
    """Models/postnets.py:13-79 (prev_version=True): Linear(d->mel) then 5 causal convs (k=5, pad 4,
    crop the last 4), BatchNorm(batch stats)+tanh+dropout after the first four, residual to mel_pred."""

    def __init__(self, num_hidden, mel_dim, dropout):
        super().__init__()
        self.conv1 = nn.Conv1d(mel_dim, num_hidden, 5, padding=4)
        self.conv_list = nn.ModuleList([nn.Conv1d(num_hidden, num_hidden, 5, padding=4) for _ in range(3)])
        self.conv2 = nn.Conv1d(num_hidden, mel_dim, 5, padding=4)
        self.out = nn.Linear(num_hidden, mel_dim)
        self.batch_norm_list = nn.ModuleList([nn.BatchNorm1d(num_hidden) for _ in range(3)])
        self.pre_batchnorm = nn.BatchNorm1d(num_hidden)
        self.p = dropout

    def forward(self, x):
        mel_pred = self.out(x).transpose(1, 2)
        h = F.dropout(torch.tanh(self.pre_batchnorm(self.conv1(mel_pred)[:, :, :-4])), self.p, self.training)
        for bn, conv in zip(self.batch_norm_list, self.conv_list):
            h = F.dropout(torch.tanh(bn(conv(h)[:, :, :-4])), self.p, self.training)
        post = self.conv2(h)[:, :, :-4] + mel_pred
        return mel_pred.transpose(1, 2), post.transpose(1, 2)


class FastSpeech2(nn.Module):
    """Models/fastspeech2.py:38-116 (ctor) and :118-241 (forward), default options only:
    transformer encoder/decoder, postnet_pred=True, no speaker / sq-vae / hop / fix_mask / debug."""

    def __init__(self, vocab, mel_dim, d_model, N_e, H_e, k_e, N_d, H_d, k_d, dropout, dropout_postnet,
                 dropout_variance_adaptor, n_bins, f0_min, f0_max, energy_min, energy_max, concat_after_encoder=False,
                 concat_after_decoder=False, pitch_pred=True, energy_pred=True, p_scheduled_sampling=0.0):
        super().__init__()
        self.encoder = Encoder(vocab, d_model, N_e, H_e, k_e, dropout, embedding=True, concat_after=concat_after_encoder)
        self.variance_adaptor = VarianceAdaptor(d_model, n_bins, f0_min, f0_max, energy_min, energy_max,
                                                dropout_variance_adaptor, pitch_pred, energy_pred)
        self.decoder = Encoder(d_model, d_model, N_d, H_d, k_d, dropout, embedding=False, concat_after=concat_after_decoder)
        self.postnet = PostConvNet(d_model, mel_dim, dropout_postnet)
        self.p_scheduled_sampling = p_scheduled_sampling              # hp.p_scheduled_sampling (Models/fastspeech2.py:178)

    @classmethod
    def from_hp(cls, hp, dropout=None, dropout_postnet=0.5, dropout_variance_adaptor=None):
        """Argument wiring of train_fastspeech2.py:381-389."""
        return cls(hp.vocab_size, hp.mel_dim, hp.d_model_encoder, hp.n_layer_encoder, hp.n_head_encoder,
                   hp.ff_conv_kernel_size_encoder, hp.n_layer_decoder, hp.n_head_decoder,
                   hp.ff_conv_kernel_size_decoder, hp.dropout if dropout is None else dropout, dropout_postnet,
                   getattr(hp, "dropout_variance_adaptor", 0.5) if dropout_variance_adaptor is None
                   else dropout_variance_adaptor, hp.nbins, hp.f0_min, hp.f0_max, hp.energy_min, hp.energy_max,
                   bool(getattr(hp, "concat_after_encoder", False)), bool(getattr(hp, "concat_after_decoder", False)),
                   bool(getattr(hp, "pitch_pred", True)), bool(getattr(hp, "energy_pred", True)),
                   float(getattr(hp, "p_scheduled_sampling", 0.0)))

    def double(self):
        """fp64 reference: also cast the plain-attribute tables."""
        super().double()
        va = self.variance_adaptor
        if va.pitch_pred:
            va.pitch_bins = va.pitch_bins.double()
        if va.energy_pred:
            va.energy_bins = va.energy_bins.double()
        return self

    def forward(self, src, src_mask, mel_mask=None, d_target=None, p_target=None, e_target=None):
        e_out, attn_enc = self.encoder(src, src_mask)
        if d_target is None:       # inference (:174-176): the variance adaptor makes the frame mask
            va_out, log_d, p_pred, e_pred, text_dur, _, mel_mask = self.variance_adaptor(e_out, src_mask, None, None,
                                                                                         None, None)
            mel_mask = mel_mask.unsqueeze(1)       # (B,T) -> key mask (B,1,T); the reference relies on B == 1 here
        else:
            va_out, log_d, p_pred, e_pred, text_dur = self.variance_adaptor(e_out, src_mask, mel_mask, d_target,
                                                                             p_target, e_target, self.p_scheduled_sampling)
        d_out, attn_dec = self.decoder(va_out, mel_mask)
        mel_before, mel_after = self.postnet(d_out)
        return (mel_before, mel_after, log_d, p_pred, e_pred, va_out, text_dur, attn_enc, attn_dec,
                None, None, None, None, None)

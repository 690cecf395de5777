"""Oracle restatement of the reference autoregressive Transformer-TTS path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Plain eager PyTorch, same state_dict keys and registration
order as the reference's ``Models.transformer.Transformer`` (default branch: transformer encoder/decoder,
single speaker, no GST, concat_after=False), so the fixture weights load into it unchanged.
Citations are file:line of /root/reference (syoamakase/Transformer_TTS).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import Encoder, FeedForward, PositionalEncoder


def attention_general(q, k, v, d_k, mask, p_drop):
    """Models/modules.py:7-21 with a general mask (B,1,tk) or (B,tq,tk): -1e4 fill where mask == 0, softmax,
    dropout with F.dropout's default training=True (active in eval() too), P V."""
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d_k)
    s = s.masked_fill(mask.unsqueeze(1) == 0, -1e4)
    p = torch.softmax(s, dim=-1)
    p = F.dropout(p, p_drop, True)
    return torch.matmul(p, v), p


class MultiHeadAttentionQKV(nn.Module):
    """Models/modules.py:23-70 (concat_after=False) with separate query / key / value inputs.  Registration order
    q_linear, v_linear, k_linear, out (:32-41); k_linear is applied to the `k` argument, v_linear to `v` (:49-51)."""

    def __init__(self, heads, d_model, dropout):
        super().__init__()
        self.h, self.d_k, self.p = heads, d_model // heads, dropout
        self.q_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.out = nn.Linear(d_model, d_model)

    def forward(self, q, k, v, mask):
        b, tq, d = q.shape
        split = lambda y: y.view(b, -1, self.h, self.d_k).transpose(1, 2)
        o, p = attention_general(split(self.q_linear(q)), split(self.k_linear(k)), split(self.v_linear(v)), self.d_k, mask, self.p)
        return self.out(o.transpose(1, 2).reshape(b, tq, d)), p


class DecoderPreNet(nn.Module):
    """Models/prenets.py:8-44 (output_type None): Linear -> ReLU -> Dropout -> Linear -> ReLU -> Dropout, hidden 256.
    nn.Dropout: active in train() only."""

    def __init__(self, input_size, output_size, hidden_size=256, p=0.5):
        super().__init__()
        self.layer = nn.Sequential(OrderedDict([
            ("fc1", nn.Linear(input_size, hidden_size)), ("relu1", nn.ReLU()), ("dropout1", nn.Dropout(p)),
            ("fc2", nn.Linear(hidden_size, output_size)), ("relu2", nn.ReLU()), ("dropout2", nn.Dropout(p))]))

    def forward(self, x):
        return self.layer(x)


class DecoderLayer(nn.Module):
    """Models/layers.py:84-125 (single speaker): pre-LN masked self-attention, encoder-decoder attention and the
    conv FFN (which carries its own residual + LayerNorm inside, Models/modules.py:72-88), each with an outer residual."""

    def __init__(self, d_model, heads, k, dropout):
        super().__init__()
        self.norm_1 = nn.LayerNorm(d_model)
        self.norm_2 = nn.LayerNorm(d_model)
        self.norm_3 = nn.LayerNorm(d_model)
        self.attn_1 = MultiHeadAttentionQKV(heads, d_model, dropout)
        self.attn_2 = MultiHeadAttentionQKV(heads, d_model, dropout)
        self.ff = FeedForward(d_model, k, dropout)
        self.p = dropout

    def forward(self, x, e_outputs, src_mask, trg_mask):
        h = self.norm_1(x)
        a, p1 = self.attn_1(h, h, h, trg_mask)
        x = x + F.dropout(a, self.p, self.training)
        h = self.norm_2(x)
        a, p2 = self.attn_2(h, e_outputs, e_outputs, src_mask)
        x = x + F.dropout(a, self.p, self.training)
        x = x + F.dropout(self.ff(self.norm_3(x)), self.p, self.training)
        return x, p1, p2


class Decoder(nn.Module):
    """Models/decoder.py:29-56: prenet -> PE -> N decoder layers -> LayerNorm; attention maps of all layers stacked
    to (B,N,H,t1,t1) and (B,N,H,t1,t2)."""

    def __init__(self, vocab, d_model, N, heads, k, dropout, dropout_prenet):
        super().__init__()
        self.decoder_prenet = DecoderPreNet(vocab, d_model, p=dropout_prenet)
        self.pe = PositionalEncoder(d_model, dropout=dropout)
        self.layers = nn.Sequential(*[DecoderLayer(d_model, heads, k, dropout) for _ in range(N)])
        self.norm = nn.LayerNorm(d_model)

    def forward(self, trg, e_outputs, src_mask, trg_mask):
        x = self.pe(self.decoder_prenet(trg))
        a1, a2 = [], []
        for layer in self.layers:
            x, p1, p2 = layer(x, e_outputs, src_mask, trg_mask)
            a1.append(p1)
            a2.append(p2)
        return self.norm(x), torch.stack(a1, dim=1), torch.stack(a2, dim=1)


class PostConvNetV2(nn.Module):
    """Models/postnets.py:13-79 with prev_version=False as Models/transformer.py:92 builds it: no `out` Linear, the five
    causal convolutions run on the mel prediction itself -- and forward RETURNS ITS INPUT (:76-79 return mel_pred in
    this branch): the convolution stack only updates the BatchNorm running statistics and receives no gradient."""

    def __init__(self, num_hidden, mel_dim, dropout):
        super().__init__()
        self.conv1 = nn.Conv1d(mel_dim, num_hidden, 5, padding=4)
        self.conv_list = nn.ModuleList([nn.Conv1d(num_hidden, num_hidden, 5, padding=4) for _ in range(3)])
        self.conv2 = nn.Conv1d(num_hidden, mel_dim, 5, padding=4)
        self.batch_norm_list = nn.ModuleList([nn.BatchNorm1d(num_hidden) for _ in range(3)])
        self.pre_batchnorm = nn.BatchNorm1d(num_hidden)
        self.p = dropout

    def forward(self, x):
        mel_pred = x.transpose(1, 2)
        h = F.dropout(torch.tanh(self.pre_batchnorm(self.conv1(mel_pred)[:, :, :-4])), self.p, self.training)
        for bn, conv in zip(self.batch_norm_list, self.conv_list):
            h = F.dropout(torch.tanh(bn(conv(h)[:, :, :-4])), self.p, self.training)
        _ = self.conv2(h)[:, :, :-4] + mel_pred
        return mel_pred.transpose(1, 2)


class Transformer(nn.Module):
    """Models/transformer.py:15-118 (transformer encoder + transformer decoder, d_model_encoder == d_model_decoder or a
    Linear between them, single speaker, no GST).  forward returns the reference's 6-tuple
    (outputs_prenet, outputs_postnet, stop_token, attn_enc, attn_dec_dec, attn_dec_enc)."""

    def __init__(self, vocab, mel_dim, d_enc, N_e, H_e, k_e, d_dec, N_d, H_d, k_d, reduction_rate, dropout,
                 dropout_prenet=0.5, dropout_postnet=0.5):
        super().__init__()
        self.encoder = Encoder(vocab, d_enc, N_e, H_e, k_e, dropout)
        self.linear = nn.Linear(d_enc, d_dec) if d_enc != d_dec else None
        self.decoder = Decoder(mel_dim, d_dec, N_d, H_d, k_d, dropout, dropout_prenet)
        self.out = nn.Linear(d_dec, mel_dim * reduction_rate)
        self.stop_token = nn.Linear(d_dec, reduction_rate)
        self.postnet = PostConvNetV2(d_dec, mel_dim * reduction_rate, dropout_postnet)
        self.reduction_rate = reduction_rate

    @classmethod
    def from_hp(cls, hp, dropout=None, dropout_prenet=None, dropout_postnet=None):
        return cls(hp.vocab_size, hp.mel_dim, hp.d_model_encoder, hp.n_layer_encoder, hp.n_head_encoder,
                   hp.ff_conv_kernel_size_encoder, hp.d_model_decoder, hp.n_layer_decoder, hp.n_head_decoder,
                   hp.ff_conv_kernel_size_decoder, hp.reduction_rate, hp.dropout if dropout is None else dropout,
                   hp.dropout_prenet if dropout_prenet is None else dropout_prenet,
                   hp.dropout_postnet if dropout_postnet is None else dropout_postnet)

    def forward(self, src, trg, src_mask, trg_mask):
        e, attn_enc = self.encoder(src, src_mask)
        if self.linear is not None:
            e = self.linear(e)
        d, a1, a2 = self.decoder(trg, e, src_mask, trg_mask)
        outputs_prenet = self.out(d)
        outputs_postnet = self.postnet(outputs_prenet)
        stop = self.stop_token(d).squeeze(2)
        return outputs_prenet, outputs_postnet, stop, attn_enc, a1, a2


# ------------------------------------------------------------------------------------------------ training step
def nopeak_mask(size):
    """train.py:26-36: lower-triangular (incl. diagonal) boolean mask (1, size, size)."""
    return torch.from_numpy(np.triu(np.ones((1, size, size)), k=1).astype("uint8") == 0)


def create_masks(src_pos, trg_pos):
    """train.py:38-58: src_mask = (src_pos != 0) (B,1,L); trg_mask = (trg_pos != 0) (B,1,T) & no-peak (1,T,T)."""
    src_mask = (src_pos != 0).unsqueeze(-2)
    trg_mask = (trg_pos != 0).unsqueeze(-2) & nopeak_mask(trg_pos.size(1)).to(trg_pos.device)
    return src_mask, trg_mask


def decoder_inputs(mel, pos_mel, r):
    """train.py:183-193: teacher forcing: the decoder sees frames 0, r, 2r, ... (all but the last group)."""
    if r > 1:
        return mel[:, :-r:r, :], pos_mel[:, :-r:r]
    return mel[:, :-1, :], pos_mel[:, :-1]


def losses(outputs, mel, stop_token, r, positive_weight):
    """train.py:205-220: outputs regrouped to frame rate, L1 of both mel outputs against mel[:, r:] (mean over every
    element), BCE-with-logits of the stop token with pos_weight (mean)."""
    pre, post, stop = outputs[:3]
    if r > 1:
        b, t, c = pre.shape
        pre = pre.reshape(b, t * r, c // r)
        post = post.reshape(b, t * r, c // r)
        stop = stop.reshape(b, t * r)
    parts = {
        "mel": F.l1_loss(pre, mel[:, r:, :]),
        "post_mel": F.l1_loss(post, mel[:, r:, :]),
        "token": F.binary_cross_entropy_with_logits(stop, stop_token[:, r:], reduction="mean",
                                                    pos_weight=torch.tensor(positive_weight, dtype=stop.dtype)),
    }
    return parts["mel"] + parts["post_mel"] + parts["token"], parts


def forward_backward(model, batch, r=1, positive_weight=5.0, accum_grad=1, zero=True):
    """train.py:176-262 on one 8-tuple batch (text, mel, pos_text, pos_mel, text_len, mel_len, stop_token, spk_emb)."""
    text, mel, pos_text, pos_mel, _, _, stop_token = batch[:7]
    dt = next(model.parameters()).dtype
    mel = mel.to(dt)
    mel_input, pos_in = decoder_inputs(mel, pos_mel, r)
    src_mask, trg_mask = create_masks(pos_text, pos_in)
    out = model(text, mel_input, src_mask, trg_mask)
    total, parts = losses(out, mel, stop_token.to(dt), r, positive_weight)
    if zero:
        for p in model.parameters():
            p.grad = None
    (total / accum_grad).backward()
    return total, parts, out


def make_optimizer(model):
    """train.py:112-119 (Adam branch)."""
    return torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)


def train_step(model, optimizer, step, batch, d_model, r=1, positive_weight=5.0, warmup_factor=1.0, warmup_step=4000,
               clip=1.0, accum_grad=1):
    """One iteration of the loop body of train.py:156-262 (non-amp branch): Noam lr from `step`, zero_grad, forward,
    losses, step += 1, backward of loss/accum_grad; when the NEW step is a multiple of accum_grad: clip + Adam."""
    from .train import noam_lr
    lr = noam_lr(step, d_model, warmup_factor, warmup_step)
    for g in optimizer.param_groups:
        g["lr"] = lr
    total, parts, _ = forward_backward(model, batch, r, positive_weight, accum_grad)
    step += 1
    assert not torch.isnan(total), "loss is nan"
    if step % accum_grad == 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        optimizer.step()
    return total.detach(), step

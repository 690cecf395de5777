"""Autograd Functions of the autoregressive Transformer-TTS path: the decoder stack of
Models/decoder.py:29-56 / Models/layers.py:84-125 (pre-net, masked self-attention, encoder-decoder attention, conv FFN),
the output / stop-token projections and the stop-token loss of the reference's train.py:214-217, sequenced on the same
gfx950 kernels as the FastSpeech2 path (functional.py).  Hand-written backward: PyTorch's autograd only carries tensors
between these blocks."""
import math
import os

import torch

from .. import ops
from .functional import _conv_wgrad, _linear_wgrad, _tp, announce_list, fp8_bwd, grad_of


def _heads(x2, B, t, n, H, dk):
    """(B*t, n*H*dk) -> n views (B,H,t,dk) of the fused projection output"""
    x5 = x2.view(B, t, n, H, dk)
    return [x5[:, :, j].permute(0, 2, 1, 3) for j in range(n)]


class DecoderStackFunction(torch.autograd.Function):
    """Decoder.forward (Models/decoder.py:44-56) with N x DecoderLayer.forward (Models/layers.py:108-125): DecoderPreNet
    (Models/prenets.py:8-44), PositionalEncoder, per layer LN -> masked self-attention -> residual, LN -> encoder-decoder
    attention -> residual, LN -> FeedForward -> residual, final LayerNorm.  Masks: `trg_km` (B,T) key padding of the decoder
    frames combined with the no-peak mask (train.py:26-58) inside the softmax kernel; `src_km` (B,L) for the encoder keys."""

    @staticmethod
    def forward(ctx, dec, trg, e, src_km, trg_km, *params):
        rt = dec.rt
        T = rt.dtype
        dev = trg.device
        rng = rt.get_rng(dev)
        p_att = dec.dropout                      # F.dropout(training=True) inside attention(): always on (modules.py:19)
        p = dec.dropout if dec.training else 0.0
        p_pre = dec.dropout_prenet if dec.training else 0.0
        N, H, d = dec.N, dec.heads, dec.d_model
        dk = d // H
        B, t, mel_dim = trg.shape
        L = e.shape[1]
        M, Me = B * t, B * L
        tp, Lp = _tp(t), _tp(L)
        scale = 1.0 / math.sqrt(dk)
        src_km_in, trg_km_in = src_km, trg_km
        src_km = src_km.reshape(B, L).contiguous()
        trg_km = trg_km.reshape(B, t).contiguous()
        e2 = e.reshape(Me, d)
        sv = {}

        # ---- pre-net (prenets.py:30-37) and positional encoding (decoder.py:45-48)
        pre = dec.decoder_prenet.layer
        x0 = trg.reshape(M, mel_dim)
        x0 = x0 if T == torch.float32 else ops.cast(x0.contiguous(), T)
        h1 = ops.linear(x0, rt.w_fwd(pre.fc1.weight), pre.fc1.bias.detach(), relu=True)
        h1d = ops.dropout(h1, p_pre, rng, dec.site_pre1) if p_pre > 0 else h1
        h2 = ops.linear(h1d, rt.w_fwd(pre.fc2.weight), pre.fc2.bias.detach(), relu=True)
        h2d = ops.dropout(h2, p_pre, rng, dec.site_pre2) if p_pre > 0 else h2
        x = ops.pe_add_fwd(h2d.view(B, t, d), dec.pe.table(dev), dec.pe.alpha.detach(), p, rng, dec.pe.site)
        n1 = dec.layers[0].norm_1
        h, mean0, rstd0 = ops.layernorm_fwd(x, n1.weight.detach(), n1.bias.detach(), T)
        sv.update(x0=x0, h1=h1, h1d=h1d, h2=h2, x_pe=x, mean0=mean0, rstd0=rstd0)

        # hp.return_attn = False: the maps are not wanted -> flash kernels in their causal (self-attention) and rectangular
        # (encoder-decoder) modes, no (T x T) / (T x L) tensor in HBM; the Philox counters are those of the (B,N,H,t,tp) layouts
        # either way, so both modes draw the same dropout masks
        # hp.return_attn = True: the same kernels, and the maps of each layer written after the fact (fs2_flash_attention_probs) for the return
        # value; FS2_FLASH_MAPS=0: the unfused scores / softmax / product path with the probabilities in HBM (A/B measurements)
        flash = ops.flash_attn_supported(max(t, L), dk, T) and (not rt.return_attn or os.environ.get("FS2_FLASH_MAPS", "1") != "0")
        if flash:
            attn1 = attn2 = None
            attn1_drop = torch.empty((B, N, H, t, tp), dtype=T, device=dev) if rt.return_attn else None
            attn2_drop = torch.empty((B, N, H, t, Lp), dtype=T, device=dev) if rt.return_attn else None
            stats1 = torch.empty((N, B, H, t, 2), dtype=torch.float32, device=dev)
            stats2 = torch.empty((N, B, H, t, 2), dtype=torch.float32, device=dev)
            keep1 = torch.empty((N, ops.flash_attn_keep_words_rect(B, H, t, t)), dtype=torch.int16, device=dev) if p_att > 0 else None
            keep2 = torch.empty((N, ops.flash_attn_keep_words_rect(B, H, t, L)), dtype=torch.int16, device=dev) if p_att > 0 else None
            # (train.create_masks leaves the row bounds / ranking of both key masks with them: no scan launches here then)
            kinfo_trg, kinfo_src = getattr(trg_km_in, "_fs2_kinfo", None), getattr(src_km_in, "_fs2_kinfo", None)
            if kinfo_trg is None or kinfo_trg.shape != (B, 3) or kinfo_trg.device != dev:
                kinfo_trg = ops.flash_mask_info(trg_km)
            if kinfo_src is None or kinfo_src.shape != (B, 3) or kinfo_src.device != dev:
                kinfo_src = ops.flash_mask_info(src_km)
        else:
            attn1 = torch.empty((B, N, H, t, tp), dtype=T, device=dev)
            attn1_drop = torch.empty_like(attn1) if p_att > 0 else attn1
            attn2 = torch.empty((B, N, H, t, Lp), dtype=T, device=dev)
            attn2_drop = torch.empty_like(attn2) if p_att > 0 else attn2
        layers = []
        for i, layer in enumerate(dec.layers):
            # ---- masked self-attention (layers.py:110-112)
            wf, _, bqkv = rt.qkv(layer.attn_1)
            qkv = ops.linear(h.view(M, d), wf, bqkv)
            q, v, k = _heads(qkv, B, t, 3, H, dk)
            O = torch.empty((B, t, H, dk), dtype=T, device=dev)
            if flash:
                ops.flash_attention_fwd(q, k, v, trg_km, O.permute(0, 2, 1, 3), stats1[i], keep1[i] if keep1 is not None else None, scale,
                                        N * H * t * tp, p_att, rng, layer.site_attn1, causal=True, key_info=kinfo_trg)
                if attn1_drop is not None:
                    ops.flash_attention_probs(q, k, v, trg_km, O.permute(0, 2, 1, 3), stats1[i], keep1[i] if keep1 is not None else None,
                                              attn1_drop[:, i], scale, p_att, causal=True, key_info=kinfo_trg)
            else:
                S, Pd = attn1[:, i], attn1_drop[:, i]
                ops.bmm(q, k, S[..., :t], trans_b=True, alpha=scale)
                ops.softmax_rect_fwd(S, Pd, trg_km, t, True, p_att, rng, layer.site_attn1)
                ops.bmm(Pd, v, O.permute(0, 2, 1, 3), trans_b=False)
            a = ops.linear(O.view(M, d), rt.w_fwd(layer.attn_1.out.weight), layer.attn_1.out.bias.detach())
            n2 = layer.norm_2
            x1, hq, m2, r2 = ops.add_ln_fwd(x, a.view(B, t, d), n2.weight.detach(), n2.bias.detach(), 1e-5, p, rng, layer.site_res1)
            # ---- encoder-decoder attention (layers.py:114-116): queries from the decoder, keys / values from e_outputs
            wf2, _, b2 = rt.qkv(layer.attn_2)            # rows [q | v | k]
            q2 = ops.linear(hq.view(M, d), wf2[:d], b2[:d])
            vk = ops.linear(e2, wf2[d:], b2[d:])          # (B*L, 2d): [v | k]
            (qc,) = _heads(q2, B, t, 1, H, dk)
            vc, kc = _heads(vk, B, L, 2, H, dk)
            O2 = torch.empty((B, t, H, dk), dtype=T, device=dev)
            if flash:
                ops.flash_attention_fwd(qc, kc, vc, src_km, O2.permute(0, 2, 1, 3), stats2[i], keep2[i] if keep2 is not None else None, scale,
                                        N * H * t * Lp, p_att, rng, layer.site_attn2, causal=False, key_info=kinfo_src)
                if attn2_drop is not None:
                    ops.flash_attention_probs(qc, kc, vc, src_km, O2.permute(0, 2, 1, 3), stats2[i], keep2[i] if keep2 is not None else None,
                                              attn2_drop[:, i], scale, p_att, causal=False, key_info=kinfo_src)
            else:
                S2, Pd2 = attn2[:, i], attn2_drop[:, i]
                ops.bmm(qc, kc, S2[..., :L], trans_b=True, alpha=scale)
                ops.softmax_rect_fwd(S2, Pd2, src_km, L, False, p_att, rng, layer.site_attn2)
                ops.bmm(Pd2, vc, O2.permute(0, 2, 1, 3), trans_b=False)
            a2 = ops.linear(O2.view(M, d), rt.w_fwd(layer.attn_2.out.weight), layer.attn_2.out.bias.detach())
            n3 = layer.norm_3
            x2, h3, m3, r3 = ops.add_ln_fwd(x1, a2.view(B, t, d), n3.weight.detach(), n3.bias.detach(), 1e-5, p, rng, layer.site_res2)
            # ---- FeedForward with its inner residual + LayerNorm (modules.py:81-88), outer residual (layers.py:122)
            ff = layer.ff
            kk = ff.f_1.weight.shape[2]
            f1 = ops.conv(h3, rt.w_fwd(ff.f_1.weight), kk, kk // 2, ff.f_1.bias.detach(), relu=True)
            f2 = ops.conv(f1, rt.w_fwd(ff.f_2.weight), kk, kk // 2, ff.f_2.bias.detach())
            lnf = ff.layer_norm
            yff, mf, rf = ops.ffn_ln_fwd(f2, h3, lnf.weight.detach(), lnf.bias.detach(), 1e-5, p, rng, layer.site_ffn)
            nn_ = dec.layers[i + 1].norm_1 if i + 1 < N else dec.norm
            x3, hn, mn, rn = ops.add_ln_fwd(x2, yff, nn_.weight.detach(), nn_.bias.detach(), 1e-5, p, rng, layer.site_res3)
            layers.append(dict(h=h, qkv=qkv, O=O, x1=x1, hq=hq, m2=m2, r2=r2, q2=q2, vk=vk, O2=O2, x2=x2, h3=h3, m3=m3, r3=r3,
                               f1=f1, f2=f2, mf=mf, rf=rf, x3=x3, mn=mn, rn=rn))
            x, h = x3, hn

        ctx.dec, ctx.sv, ctx.layers = dec, sv, layers
        ctx.attn = (None, None, None, None) if flash else (attn1, attn1_drop, attn2, attn2_drop)
        ctx.flash = (stats1, stats2, keep1, keep2, kinfo_trg, kinfo_src, trg_km, src_km) if flash else None
        ctx.e2, ctx.dims = e2, (B, t, L, mel_dim)
        ctx.set_materialize_grads(False)
        if attn1_drop is None:
            a1_out, a2_out = torch.empty((B, N, H, 0, 0), dtype=T, device=dev), torch.empty((B, N, H, 0, 0), dtype=T, device=dev)
        else:
            a1_out, a2_out = attn1_drop[..., :t], attn2_drop[..., :L]
        ctx.mark_non_differentiable(a1_out, a2_out)
        return h, a1_out, a2_out

    @staticmethod
    @fp8_bwd
    def backward(ctx, dh, _da1, _da2):
        dec, sv, layers = ctx.dec, ctx.sv, ctx.layers
        attn1, attn1_drop, attn2, attn2_drop = ctx.attn
        rt = dec.rt
        T = rt.dtype
        rng = rt.rng
        p = dec.dropout if dec.training else 0.0
        p_att = dec.dropout
        p_pre = dec.dropout_prenet if dec.training else 0.0
        N, H, d = dec.N, dec.heads, dec.d_model
        dk = d // H
        B, t, L, mel_dim = ctx.dims
        M, Me = B * t, B * L
        tp, Lp = _tp(t), _tp(L)
        dev = dh.device
        scale = 1.0 / math.sqrt(dk)
        e2 = ctx.e2
        dh = dh.contiguous()
        dx = None                         # fp32 gradient of the residual stream coming from above
        de = None                         # fp32 gradient of e_outputs, summed over the layers
        flash = ctx.flash is not None
        if flash:
            stats1, stats2, keep1, keep2, kinfo_trg, kinfo_src, trg_km, src_km = ctx.flash
            aux = torch.empty((B, H, t, 4), dtype=torch.float32, device=dev)
        else:
            dP = torch.empty((B, H, t, tp), dtype=T, device=dev)
            dP2 = torch.empty((B, H, t, Lp), dtype=T, device=dev)
        for i in reversed(range(N)):
            layer, Lr = dec.layers[i], layers[i]
            nn_ = dec.layers[i + 1].norm_1 if i + 1 < N else dec.norm
            dx2, dyff = ops.add_ln_bwd(dx, dh, Lr["x3"], nn_.weight.detach(), Lr["mn"], Lr["rn"], grad_of(nn_.weight),
                                       grad_of(nn_.bias), p, rng, layer.site_res3)
            ff = layer.ff
            lnf = ff.layer_norm
            g = ops.ffn_ln_bwd(dyff, Lr["f2"], Lr["h3"], lnf.weight.detach(), Lr["mf"], Lr["rf"], grad_of(lnf.weight),
                               grad_of(lnf.bias), p, rng, layer.site_ffn, dcolsum=grad_of(ff.f_2.bias))
            kk = ff.f_1.weight.shape[2]
            pad = kk // 2
            _conv_wgrad(rt, g, Lr["f1"], ff.f_2, pad, bias_done=True)
            dz1 = ops.conv(g, rt.w_dgrad(ff.f_2.weight), kk, kk - 1 - pad, relu_mask=Lr["f1"], colsum=grad_of(ff.f_1.bias))
            _conv_wgrad(rt, dz1, Lr["h3"], ff.f_1, pad, bias_done=True)
            dh3 = ops.conv(dz1, rt.w_dgrad(ff.f_1.weight), kk, kk - 1 - pad, residual=g)
            # ---- encoder-decoder attention
            n3, at2 = layer.norm_3, layer.attn_2
            dx1, da2 = ops.add_ln_bwd(dx2, dh3, Lr["x2"], n3.weight.detach(), Lr["m3"], Lr["r3"], grad_of(n3.weight),
                                      grad_of(n3.bias), p, rng, layer.site_res2, dcolsum=grad_of(at2.out.bias))
            da2_ = da2.view(M, d)
            _linear_wgrad(rt, da2_, Lr["O2"].view(M, d), at2.out, bias_done=True)
            dO2 = ops.linear(da2_, rt.w_dgrad(at2.out.weight)).view(B, t, H, dk).permute(0, 2, 1, 3)
            (qc,) = _heads(Lr["q2"], B, t, 1, H, dk)
            vc, kc = _heads(Lr["vk"], B, L, 2, H, dk)
            dq2 = torch.empty((M, d), dtype=T, device=dev)
            dvk = torch.empty((Me, 2 * d), dtype=T, device=dev)
            (dqc,) = _heads(dq2, B, t, 1, H, dk)
            dvc, dkc = _heads(dvk, B, L, 2, H, dk)
            if flash:       # probabilities recomputed from q, k and the row statistics; the three bias gradients are fused
                ops.flash_attention_bwd(qc, kc, vc, src_km, Lr["O2"].permute(0, 2, 1, 3), dO2, stats2[i], keep2[i] if keep2 is not None else None,
                                        aux, dqc, dkc, dvc, scale, p_att, causal=False, key_info=kinfo_src,
                                        dbias=[grad_of(lin.bias) for lin in (at2.q_linear, at2.k_linear, at2.v_linear)])
            else:
                P2, Pd2 = attn2[:, i], attn2_drop[:, i]
                ops.bmm(Pd2, dO2, dvc, trans_a=True, trans_b=False)                # dV = Pd^T dO
                ops.bmm(dO2, vc, dP2[..., :L], trans_b=True)                        # dP = dO V^T
                ops.softmax_rect_bwd(dP2, P2, L, p_att, rng, layer.site_attn2)      # -> dS (pad columns 0)
                ops.bmm(dP2, kc, dqc, trans_b=False, alpha=scale)                   # dQ = dS K / sqrt(dk)
                ops.bmm(dP2, qc, dkc, trans_a=True, trans_b=False, alpha=scale)     # dK = dS^T Q / sqrt(dk)
            _, wd2, _ = rt.qkv(at2)                                             # (d, 3d): columns [q | v | k]
            with rt.side(dq2, dvk):
                if not flash:
                    ops.colsum(dq2, grad_of(at2.q_linear.bias))
                ops.wgrad(dq2, Lr["hq"].view(M, d), grad_of(at2.q_linear.weight), defer=rt.defer_wgrad)
                if not flash:
                    ops.colsum_blocks(dvk, [grad_of(at2.v_linear.bias), grad_of(at2.k_linear.bias)])
                ops.wgrad_batched(dvk, e2, [grad_of(at2.v_linear.weight), grad_of(at2.k_linear.weight)], defer=rt.defer_wgrad)
            dhq = ops.linear(dq2, wd2[:, :d]).view(B, t, d)
            de = ops.linear(dvk, wd2[:, d:], residual=de, out_dtype=torch.float32)
            # ---- masked self-attention
            n2, at1 = layer.norm_2, layer.attn_1
            dx, da = ops.add_ln_bwd(dx1, dhq, Lr["x1"], n2.weight.detach(), Lr["m2"], Lr["r2"], grad_of(n2.weight),
                                    grad_of(n2.bias), p, rng, layer.site_res1, dcolsum=grad_of(at1.out.bias))
            da_ = da.view(M, d)
            _linear_wgrad(rt, da_, Lr["O"].view(M, d), at1.out, bias_done=True)
            dO = ops.linear(da_, rt.w_dgrad(at1.out.weight)).view(B, t, H, dk).permute(0, 2, 1, 3)
            q, v, k = _heads(Lr["qkv"], B, t, 3, H, dk)
            dqkv = torch.empty((M, 3 * d), dtype=T, device=dev)
            dq, dv, dk_ = _heads(dqkv, B, t, 3, H, dk)
            if flash:
                ops.flash_attention_bwd(q, k, v, trg_km, Lr["O"].permute(0, 2, 1, 3), dO, stats1[i], keep1[i] if keep1 is not None else None, aux,
                                        dq, dk_, dv, scale, p_att, causal=True, key_info=kinfo_trg,
                                        dbias=[grad_of(lin.bias) for lin in (at1.q_linear, at1.k_linear, at1.v_linear)])
            else:
                P1, Pd1 = attn1[:, i], attn1_drop[:, i]
                ops.bmm(Pd1, dO, dv, trans_a=True, trans_b=False)
                ops.bmm(dO, v, dP[..., :t], trans_b=True)
                ops.softmax_rect_bwd(dP, P1, t, p_att, rng, layer.site_attn1)
                ops.bmm(dP, k, dq, trans_b=False, alpha=scale)
                ops.bmm(dP, q, dk_, trans_a=True, trans_b=False, alpha=scale)
            with rt.side(dqkv):
                if not flash:
                    ops.colsum_blocks(dqkv, [grad_of(lin.bias) for lin in (at1.q_linear, at1.v_linear, at1.k_linear)])
                ops.wgrad_batched(dqkv, Lr["h"].view(M, d), [grad_of(lin.weight) for lin in (at1.q_linear, at1.v_linear, at1.k_linear)], defer=rt.defer_wgrad)
            _, wd1, _ = rt.qkv(at1)
            dh = ops.linear(dqkv, wd1).view(B, t, d)
            rt.announce(announce_list(layer, id(nn_), lambda: [q_ for name, q_ in layer.named_parameters() if not name.startswith("norm_1.")]
                                      + list(nn_.parameters())))

        n1 = dec.layers[0].norm_1
        dx0 = ops.layernorm_bwd(dh, sv["x_pe"], n1.weight.detach(), sv["mean0"], sv["rstd0"], grad_of(n1.weight),
                                grad_of(n1.bias), dx=dx)
        pre = dec.decoder_prenet.layer
        # PE (+ its dropout), then Dropout(ReLU(fc2)) and Dropout(ReLU(fc1)) backwards
        dh2d = ops.pe_add_bwd(dx0, dec.pe.table(dev), T, grad_of(dec.pe.alpha), p, rng, dec.pe.site).view(M, d)
        dz2 = ops.dropout(dh2d, p_pre, rng, dec.site_pre2, relu_gate=sv["h2"])
        _linear_wgrad(rt, dz2, sv["h1d"], pre.fc2)
        dh1d = ops.linear(dz2, rt.w_dgrad(pre.fc2.weight))
        dz1_ = ops.dropout(dh1d, p_pre, rng, dec.site_pre1, relu_gate=sv["h1"])
        _linear_wgrad(rt, dz1_, sv["x0"], pre.fc1)
        rt.announce(list(pre.parameters()) + [dec.pe.alpha] + list(n1.parameters()))
        rt.side_join()
        de_T = None
        if de is not None:
            de_T = (de if T == torch.float32 else ops.cast(de, T)).view(B, L, d)
        return (None, None, de_T, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


class LinearFunction(torch.autograd.Function):
    """nn.Linear on (B,t,K) -> (B,t,N) fp32 output (Models/transformer.py:62,88-89,104,108: encoder->decoder Linear, mel
    output projection, stop-token projection)."""

    @staticmethod
    def forward(ctx, mod, rt, x, out_fp32, *params):
        B, t, K = x.shape
        N = mod.weight.shape[0]
        x2 = x.reshape(B * t, K)
        ctx.mod, ctx.rt, ctx.x2, ctx.shape = mod, rt, x2, (B, t, K, N)
        ctx.set_materialize_grads(False)
        if N < 8:       # the stop token (d -> reduction_rate): one (d -> 1) dot product per frame and output
            ones = rt.zeros(("ones", B, t), (B, t), torch.bool, x.device)
            ones.fill_(True)
            ctx.ones = ones
            w, b = mod.weight.detach(), mod.bias.detach()
            cols = [ops.linear1_fwd(x2.view(B, t, K), w[n].contiguous(), b[n:n + 1], ones) for n in range(N)]
            return cols[0].view(B, t, 1) if N == 1 else torch.stack(cols, dim=2)
        return ops.linear(x2, rt.w_fwd(mod.weight), mod.bias.detach(), out_dtype=torch.float32 if out_fp32 else None).view(B, t, N)

    @staticmethod
    @fp8_bwd
    def backward(ctx, dy):
        mod, rt, x2 = ctx.mod, ctx.rt, ctx.x2
        B, t, K, N = ctx.shape
        if dy is None:
            return (None,) * (4 + len(list(mod.parameters())))
        T = rt.dtype
        if N < 8:
            gw, gb = grad_of(mod.weight), grad_of(mod.bias)
            dy3 = dy.reshape(B, t, N).float()
            dx = None
            for n in range(N):      # the kernel accumulates into dw / db: the gradient rows are passed directly
                dxn = ops.linear1_bwd(dy3[:, :, n].contiguous(), x2.view(B, t, K), mod.weight.detach()[n].contiguous(), ctx.ones,
                                      gw[n], gb[n:n + 1])
                dx = dxn if dx is None else dx.add_(dxn)        # (reduction_rate > 1 only)
            rt.announce(announce_list(mod, "all", mod.parameters))
            return (None, None, dx.view(B, t, K), None) + (None,) * len(list(mod.parameters()))
        dy2 = dy.reshape(B * t, N).contiguous()
        dy2 = dy2 if dy2.dtype == T else ops.cast(dy2, T)
        _linear_wgrad(rt, dy2, x2, mod)
        dx = ops.linear(dy2, rt.w_dgrad(mod.weight))
        rt.announce(announce_list(mod, "all", mod.parameters))
        rt.side_join()
        return (None, None, dx.view(B, t, K), None) + (None,) * len(list(mod.parameters()))


class BCEWithLogitsFunction(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(x, y, reduction='mean', pos_weight) of train.py:217."""

    @staticmethod
    def forward(ctx, x, y, pos_weight):
        x, y = x.contiguous(), y.contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        ops.bce_logits_fwd(x, y, pos_weight, loss)
        ctx.x, ctx.y, ctx.pw = x, y, pos_weight
        return loss[0]

    @staticmethod
    @fp8_bwd
    def backward(ctx, g):
        gs = g.reshape(1).to(torch.float32).contiguous()
        return ops.bce_logits_bwd(ctx.x, ctx.y, ctx.pw, gs, ctx.x.dtype), None, None


def bce_with_logits(x, y, pos_weight):
    return BCEWithLogitsFunction.apply(x, y, pos_weight)


def postnet_update_statistics(mod, mel_pred):
    """PostConvNet.forward with prev_version=False (Models/postnets.py:64-79) as Models/transformer.py:92,105 uses it: the
    five causal convolutions run on the mel prediction, but the branch RETURNS ITS INPUT -- their only effect is the update
    of the BatchNorm running statistics (train()).  No gradient reaches the post-net's parameters."""
    rt = mod.rt
    T = rt.dtype
    rng = rt.get_rng(mel_pred.device)
    p = mod.dropout if mod.training else 0.0
    if not mod.training:
        return
    B, t, C0 = mel_pred.shape
    M = B * t
    with torch.no_grad():
        h = mel_pred.detach().contiguous()
        h = h if h.dtype == T else ops.cast(h, T)
        convs = [mod.conv1] + list(mod.conv_list)
        bns = [mod.pre_batchnorm] + list(mod.batch_norm_list)
        for li, (cv, bn) in enumerate(zip(convs, bns)):
            C = cv.weight.shape[0]
            sums = torch.zeros(2 * C + 4, dtype=torch.float32, device=h.device)
            c = ops.conv(h, rt.w_fwd(cv.weight), 5, 4, cv.bias.detach(), colstats=sums)
            count = None
            if rt.dp is not None:
                sums[2 * C:].fill_(float(M))
                rt.dp.allreduce_sum(sums)
                count = sums[2 * C:2 * C + 1]
            mean, rstd = ops.bn_finalize(sums, M, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                         bn.num_batches_tracked, count_dev=count)
            if li < 3:      # the last block's output only feeds conv2, whose result is discarded
                h = ops.bn_tanh_fwd(c, mean, rstd, bn.weight.detach(), bn.bias.detach(), p, rng, mod.sites[li])

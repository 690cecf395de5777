"""Shared test helpers: load golden fixtures, rebuild the fixture model in the oracle."""
import os
import re

import numpy as np
import torch

from golden_configs import CONFIGS, hp_namespace, digest, sample_index  # noqa: F401
from transformer_tts_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))        # numpy's default allow_pickle=False: plain arrays only


def golden_shapes(g, prefix=""):
    """{state_dict key: shape} from the pickle-free pair (unicode keys, (n,4) int dims padded with -1)"""
    return {str(k): tuple(int(x) for x in d if x >= 0) for k, d in zip(g[prefix + "shape_keys"], g[prefix + "shape_dims"])}


def oracle_model(name, dtype=torch.float32):
    from oracle.model import FastSpeech2
    cfg = CONFIGS[name]
    hp = hp_namespace(cfg)
    g = load_golden(cfg.get("shapes_from", name))       # (the inference-only option configs take their state_dict shapes from the opt_* fixture)
    m = FastSpeech2.from_hp(hp, dropout=0.0, dropout_postnet=0.0, dropout_variance_adaptor=0.0)
    m.load_state_dict(synthetic.recipe_state_dict(golden_shapes(g), cfg["weight_seed"]))
    m.train()
    if dtype == torch.float64:
        m.double()
    return m, hp, g


def check_digest(actual, dig, rtol, atol, what):
    """Compare a tensor with a stored digest [sum, sum|x|, l2, n, samples...]."""
    x = torch.as_tensor(actual).detach().double().cpu().reshape(-1)
    n = int(dig[3])
    assert x.numel() == n, f"{what}: numel {x.numel()} != {n}"
    samples = x[torch.from_numpy(sample_index(n))].numpy()
    np.testing.assert_allclose(samples, dig[4:], rtol=rtol, atol=atol, err_msg=f"{what}: samples")
    l2 = float((x * x).sum().sqrt())
    assert abs(l2 - dig[2]) <= rtol * abs(dig[2]) + atol * np.sqrt(n), f"{what}: l2 {l2} vs {dig[2]}"
    s_abs = float(x.abs().sum())
    assert abs(s_abs - dig[1]) <= rtol * abs(dig[1]) + atol * n, f"{what}: sum|x| {s_abs} vs {dig[1]}"


_NULL_GRAD = re.compile(r"(attn(_\d)?\.k_linear\.bias|postnet\.conv1\.bias|postnet\.conv_list\.\d+\.bias)$")


def is_null_gradient_param(key):
    """Parameters whose loss gradient is exactly 0 in exact arithmetic: the key bias (softmax is
    invariant to a per-query constant) and conv biases feeding a batch-statistics BatchNorm (the
    mean subtraction removes them).  Their computed gradients are rounding noise (~1e-9) which
    Adam's g/(sqrt(v)+1e-9) turns into +-lr steps: chaotic in the reference itself, so the values
    after an optimizer step are not pinned for them."""
    return _NULL_GRAD.search(key) is not None


def product_model(name, amp=False, dropout=0.0, device="cpu", return_attn=True):
    """The HIP-backed FastSpeech2 (transformer_tts_amd) with the fixture weights of config `name`."""
    from transformer_tts_amd.Models.fastspeech2 import FastSpeech2
    from transformer_tts_amd.utils.utils import fill_variables
    cfg = CONFIGS[name]
    hp = hp_namespace(cfg)
    hp.amp = amp
    hp.return_attn = return_attn
    fill_variables(hp, verbose=False)
    g = load_golden(cfg.get("shapes_from", name))
    m = FastSpeech2(hp=hp, src_vocab=hp.vocab_size, trg_vocab=hp.mel_dim, d_model_encoder=hp.d_model_encoder,
                    N_e=hp.n_layer_encoder, n_head_encoder=hp.n_head_encoder,
                    ff_conv_kernel_size_encoder=hp.ff_conv_kernel_size_encoder,
                    concat_after_encoder=hp.concat_after_encoder, d_model_decoder=hp.d_model_decoder,
                    N_d=hp.n_layer_decoder, n_head_decoder=hp.n_head_decoder,
                    ff_conv_kernel_size_decoder=hp.ff_conv_kernel_size_decoder,
                    concat_after_decoder=hp.concat_after_decoder, dropout_variance_adaptor=dropout,
                    reduction_rate=hp.reduction_rate, dropout=dropout, dropout_postnet=dropout, n_bins=hp.nbins,
                    f0_min=hp.f0_min, f0_max=hp.f0_max, energy_min=hp.energy_min, energy_max=hp.energy_max,
                    pitch_pred=hp.pitch_pred, energy_pred=hp.energy_pred, accent_emb=hp.accent_emb,
                    output_type=hp.output_type, num_group=hp.num_group, multi_speaker=hp.is_multi_speaker,
                    spk_emb_dim=hp.spk_emb_dim, spk_emb_architecture=hp.spk_emb_architecture)
    m.load_state_dict(synthetic.recipe_state_dict(golden_shapes(g), cfg["weight_seed"]))
    m.train()
    return m.to(device), hp, g


def batch_to(batch, device):
    return tuple(b.to(device) if torch.is_tensor(b) else b for b in batch)


def record_measure(name, value):
    """append a measured error to gpurun_out/measured.jsonl (when that directory exists: the GPU box): the stated tolerances of the
    bf16 / fp8 modes are twice the values measured there"""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "measured.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "value": float(value)}) + "\n")

"""Host utilities of the FastSpeech2 training path (reference: ``utils/utils.py``).

``fill_variables`` (:184-201), ``init_weight`` (:153-177), ``get_learning_rate`` (:204-215),
``load_model`` (:107-134), ``log_config`` (:57-66) keep the reference's names, arguments and
behaviour; the bodies are new.
"""
import torch
import torch.nn as nn

# defaults injected when the hparams file does not define a key (reference utils/utils.py:185-192)
DEFAULTS = {
    "spm_model": None, "mean_file": None, "var_file": None, "log_dir": "logs", "positive_weight": 5.0,
    "is_multi_speaker": False, "num_speaker": None, "spk_emb_type": None, "spk_emb_architecture": "",
    "output_type": None, "num_group": None, "pitch_pred": True, "energy_pred": True, "model": "Fastspeech2",
    "amp": True, "gst": False, "encoder_type": "transformer", "clip": 1.0, "decoder_type": "transformer",
    "accent_emb": False, "channel_wise": False, "tail_alignment": "_alignment", "gender_emb": False,
    "ctc_out": False, "concat": False, "vq_code": False, "speaker_emb": False, "spk_emb_postprocess_type": None,
    "spk_emb_dim_postprocess": None, "mask": False, "post_conformer": False, "fix_mask": None,
    "use_cosine_emb_loss": False, "n_layer_post_model": 6, "semantic_mask": False, "time_weight": None,
    "mask_probability": 0.06, "ff_conv_kernel_size_post": 5, "concat_after_post": True,
    "intermediate_layers_out": None, "dropout_variance_adaptor": 0.5, "use_sq_vae": False, "spk_emb_dim": None,
    "use_rnn_length": False, "use_pos": False, "p_scheduled_sampling": 0.0, "use_ssim": False, "spk_emb_vers": 1,
    "use_hop": False,
    # keys that exist only in this build
    "return_attn": True,       # keep the (B,N,H,t,t) attention maps in the 14-tuple (reference always does)
    "overlap_wgrad": False,    # weight-gradient GEMMs on a second HIP stream (measured slower on a saturated GPU: DESIGN.md section 4)
    "log_every": 1,            # print losses every N steps (the reference prints every step)
    "use_graph": True,         # train_loop replays one hipGraph per batch shape (False: every kernel launched from Python)
}


def fill_variables(hp, verbose=True):
    for key, value in DEFAULTS.items():
        if not hasattr(hp, key):
            if verbose:
                print(f"{key} is not found in hparams. defalut {value} is used.")
            setattr(hp, key, value)
    if hp.spk_emb_postprocess_type == "x_vector" and hp.spk_emb_dim_postprocess is None:
        hp.spk_emb_dim_postprocess = 512
    assert not hasattr(hp, "spkr_emb"), \
        "hp.spkr_emb is future depricated, please use hp.spk_emb_architecture instead."


def log_config(hp):
    print("PARAMETER ......")
    for key, value in vars(hp).items():
        if not key.startswith("_"):
            print(f"{key} = {value}")
    print()


def get_learning_rate(step, d_model, warmup_factor, warmup_step):
    """Noam schedule: warmup_factor * min(step^-0.5, step * warmup_step^-1.5) * d_model^-0.5."""
    return warmup_factor * min(step ** -0.5, step * warmup_step ** -1.5) * (d_model ** -0.5)


def init_weight(m):
    """Class-name driven init of the reference (applied with ``model.apply``): modules whose class
    name contains ``Conv1d``/``Conv2d`` get a Kaiming-normal weight and a zero bias, ``LSTM``
    parameters get Kaiming / zero; the reference's test for the lower-case substring ``linear``
    never matches ``Linear``, so Linear / Embedding / LayerNorm / BatchNorm keep PyTorch defaults."""
    name = m.__class__.__name__
    if "linear" in name:
        m.weight.data.uniform_(-0.1, 0.1)
        if isinstance(m.bias, nn.Parameter):
            m.bias.data.fill_(0)
    if "LSTM" in name:
        for pname, param in m.named_parameters():
            if "weight" in pname:
                nn.init.kaiming_normal_(param.data)
            if "bias" in pname:
                param.data.fill_(0)
    if "Conv1d" in name or "Conv2d" in name:
        nn.init.kaiming_normal_(m.weight.data)
        if isinstance(m.bias, nn.Parameter):
            m.bias.data.fill_(0)


def load_model(model_file, map_location=None):
    """Load a state_dict saved from a single-GPU or a DDP-wrapped (``module.`` prefixed) model and
    add / strip the prefix to match the current GPU count (reference utils/utils.py:107-134).
    ``weights_only=True``: nothing is executed from the file."""
    state = torch.load(model_file, map_location=map_location, weights_only=True)
    want_prefix = torch.cuda.device_count() > 1
    has_prefix = "module" in next(iter(state.keys()))
    if want_prefix == has_prefix:
        return state
    if want_prefix:
        return {"module." + k: v for k, v in state.items()}
    return {k[7:]: v for k, v in state.items()}

"""Inference branch (predicted durations / pitch / energy, eval() mode), CPU side:
the oracle against fixtures produced by the reference's own eval-mode forward (tests/golden/infer_*.npz, generated
by tests/golden/make_golden.py infer), the product's host composition against the same fixtures with the HIP ops
replaced by the oracle primitives, and the synthesis script end to end on a checkpoint."""
import os

import numpy as np
import pytest
import torch

from fake_backend import fake_ops  # noqa: F401
from helpers import GOLDEN, oracle_model, product_model

NAMES = ["mel_before", "mel_after", "log_d", "p_pred", "e_pred", "va_out", "text_dur"]
ALL = ["tiny", "small", "inf_concat", "inf_nopitch", "inf_noenergy"]      # (inf_*: the option fixtures, make_golden.py infer_options)


def load(name):
    return np.load(os.path.join(GOLDEN, f"infer_{name}.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ALL)
def test_oracle_inference_matches_reference(name):
    g = load(name)
    model = oracle_model(name)[0]
    model.eval()
    for b in range(int(g["n_utt"])):
        text = torch.from_numpy(g[f"u{b}.text"])
        with torch.no_grad():
            out = model(text, torch.ones(1, 1, text.shape[1], dtype=torch.bool))
        dur = torch.clamp(torch.round(torch.exp(out[2]) - 1), min=0)
        assert np.array_equal(dur.numpy(), g[f"u{b}.duration_rounded"]), "rounded durations are integers: exact"
        for i, k in enumerate(NAMES):
            if out[i] is None:          # p_pred / e_pred with hp.pitch_pred / hp.energy_pred False
                assert f"u{b}.{k}" not in g.files
                continue
            np.testing.assert_allclose(out[i].numpy(), g[f"u{b}.{k}"], rtol=1e-5, atol=1e-6, err_msg=f"utt {b} {k}")


@pytest.mark.parametrize("name", ALL)
def test_product_inference_composition_matches_reference(fake_ops, name):
    g = load(name)
    model = product_model(name)[0]
    model.eval()
    for b in range(int(g["n_utt"])):
        text = torch.from_numpy(g[f"u{b}.text"])
        pos = torch.arange(1, text.shape[1] + 1).unsqueeze(0)
        with torch.no_grad():
            out = model(text, (pos != 0).unsqueeze(-2))
        assert len(out) == 14 and all(o is None for o in out[9:])
        T = int(g[f"u{b}.duration_rounded"].sum())
        assert out[0].shape == (1, T, 80) and out[8].shape[-2:] == (T, T)
        for i, k in enumerate(NAMES):
            if out[i] is None:
                assert f"u{b}.{k}" not in g.files
                continue
            np.testing.assert_allclose(out[i].float().numpy(), g[f"u{b}.{k}"], rtol=2e-5, atol=2e-5, err_msg=f"utt {b} {k}")
    # eval() must not have touched the BatchNorm running statistics
    assert int(model.postnet.pre_batchnorm.num_batches_tracked) == 0
    with pytest.raises(AssertionError):
        model.train()
        model(text, (pos != 0).unsqueeze(-2))


def test_synthesis_script_writes_mels(fake_ops, tmp_path, monkeypatch):
    """test_fastspeech2.py end to end (reference CLI): hparams.py next to the checkpoint, a test script with two
    utterances, mean/var de-normalisation, .npy + _alignment.npy outputs equal to a direct model call."""
    from golden_configs import CONFIGS, _BASE
    from transformer_tts_amd import test_fastspeech2 as synth
    g = load("tiny")
    model = product_model("tiny")[0]
    ckpt_dir = tmp_path / "ckpt"
    ckpt_dir.mkdir()
    torch.save(model.state_dict(), ckpt_dir / "network.average_epoch3")
    mean, var = np.linspace(-1, 1, 80, dtype=np.float32), np.linspace(0.5, 2.0, 80, dtype=np.float32)
    np.save(tmp_path / "mean.npy", mean)
    np.save(tmp_path / "var.npy", var)
    script = tmp_path / "test.txt"
    script.write_text("".join(f"{tmp_path}/utt{b}.npy|{' '.join(str(int(i)) for i in g[f'u{b}.text'][0])}\n" for b in range(2)))
    hpd = dict(_BASE)
    hpd.update(CONFIGS["tiny"]["hp"])
    hpd.update(test_script=str(script), mean_file=str(tmp_path / "mean.npy"), var_file=str(tmp_path / "var.npy"))
    (ckpt_dir / "hparams.py").write_text("".join(f"{k} = {v!r}\n" for k, v in hpd.items()))
    synth.main(["--load_name", str(ckpt_dir / "network.average_epoch3")])
    for b in range(2):
        mel = np.load(ckpt_dir / "dev.7" / "epoch3" / f"utt{b}.npy")
        ali = np.load(ckpt_dir / "dev.7" / "epoch3" / f"utt{b}_alignment.npy")
        assert np.array_equal(ali, g[f"u{b}.duration_rounded"][0])
        np.testing.assert_allclose(mel, g[f"u{b}.mel_after"][0] * np.sqrt(var) + mean, rtol=2e-5, atol=5e-5)

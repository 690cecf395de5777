"""Seeded synthetic (phoneme-seq, mel) batches and a deterministic weight recipe.

The batch has the field order and dtypes of the 16-tuple the reference's
``collate_fn`` produces (``datasets/datasets_fastspeech2.py:521-616`` of the
reference; single-speaker return at ``:613``): text i64 (B,L) pad 0, mel f32
(B,T,80) pad -0.5, pos_text i64 1..len pad 0, pos_mel i64, text_len, mel_len,
stop_token f32 pad 1.0, spk_emb None, f0 f32 pad 0, energy f32 pad 0,
alignment i64 pad 0, accent None, gender None, spk_emb_postprocess None,
mel_name list[str], hop_size list[None].

``benchmark_batch`` is the headline workload of BASELINE.json configs[1]
(SURVEY.md section 8(d)): rng 2024, B=48 -> L_pad=128, T_pad=925, sum(mel)=32172.
"""
import numpy as np
import torch

FIELDS = ("text", "mel", "pos_text", "pos_mel", "text_lengths", "mel_lengths", "stop_token",
          "spk_emb", "f0", "energy", "alignment", "accent", "gender", "spk_emb_postprocess",
          "mel_name", "hop_size")


def make_batch(seed, batch_size, l_range=(64, 129), dur_range=(2, 13), vocab=152, mel_dim=80,
               f0_range=(71.0, 799.8), energy_range=(0.0, 403.8), zero_dur_prob=0.0):
    """Draw one padded batch. Draw order: all phoneme lengths, then per-utterance durations,
    then per utterance ids, mel, f0, energy."""
    rng = np.random.default_rng(seed)
    Ls = rng.integers(l_range[0], l_range[1], size=batch_size)
    durs = [rng.integers(dur_range[0], dur_range[1], size=int(L)) for L in Ls]
    if zero_dur_prob > 0.0:
        for d in durs:
            d[rng.random(d.shape[0]) < zero_dur_prob] = 0
            if d.sum() == 0:
                d[0] = 1
    Ts = [int(d.sum()) for d in durs]
    L_pad, T_pad = int(max(Ls)), int(max(Ts))
    B = batch_size
    text = np.zeros((B, L_pad), np.int64)
    mel = np.full((B, T_pad, mel_dim), -0.5, np.float32)
    pos_text = np.zeros((B, L_pad), np.int64)
    pos_mel = np.zeros((B, T_pad), np.int64)
    stop = np.ones((B, T_pad), np.float32)
    f0 = np.zeros((B, T_pad), np.float32)
    energy = np.zeros((B, T_pad), np.float32)
    align = np.zeros((B, L_pad), np.int64)
    for b in range(B):
        L, T = int(Ls[b]), Ts[b]
        text[b, :L] = rng.integers(1, vocab, size=L)
        mel[b, :T] = rng.standard_normal((T, mel_dim)).astype(np.float32)
        f0[b, :T] = rng.uniform(f0_range[0], f0_range[1], size=T).astype(np.float32)
        energy[b, :T] = rng.uniform(energy_range[0], energy_range[1], size=T).astype(np.float32)
        pos_text[b, :L] = np.arange(1, L + 1)
        pos_mel[b, :T] = np.arange(1, T + 1)
        stop[b, :T] = 0.0
        if T > 0:
            stop[b, T - 1] = 1.0
        align[b, :L] = durs[b]
    t = torch.from_numpy
    return (t(text), t(mel), t(pos_text), t(pos_mel), t(np.asarray(Ls, np.int64)),
            t(np.asarray(Ts, np.int64)), t(stop), None, t(f0), t(energy), t(align), None, None, None,
            [f"synthetic_{seed}_{b}" for b in range(B)], [None] * B)


def benchmark_batch(seed=2024, batch_size=48):
    """BASELINE.json configs[1] batch: L in [64,128], durations in [2,12]."""
    return make_batch(seed, batch_size)


def tiny_batch(seed=7, batch_size=3, vocab=40):
    """Small ragged batch for parity tests (contains zero durations)."""
    return make_batch(seed, batch_size, l_range=(5, 13), dur_range=(1, 6), vocab=vocab, zero_dur_prob=0.15)


def recipe_state_dict(shapes, seed):
    """Deterministic weights for a ``{key: shape}`` template (sorted-key order, numpy RNG).

    Non-trivial biases / norm gains / BN running stats on purpose so that a kernel that drops one
    of them fails parity.  Used both by ``tests/golden/make_golden.py`` (loaded into the imported
    reference model) and by the tests (loaded into the oracle and the HIP-backed model).
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        n = int(np.prod(shape)) if len(shape) else 1
        z = rng.standard_normal(n).astype(np.float32).reshape(shape)
        is_norm = ("norm" in key) and len(shape) == 1
        if leaf == "running_mean":
            w = 0.1 * z
        elif leaf == "running_var":
            w = 1.0 + 0.1 * np.abs(z)
        elif leaf == "alpha":
            w = 1.0 + 0.05 * z
        elif is_norm and leaf == "weight":
            w = 1.0 + 0.1 * z
        elif is_norm and leaf == "bias":
            w = 0.05 * z
        elif leaf == "bias":
            w = 0.05 * z
        elif "embed" in key and len(shape) == 2 and "decoder.embed" not in key:
            w = 0.5 * z
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            w = z / np.sqrt(max(fan_in, 1))
        out[key] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out

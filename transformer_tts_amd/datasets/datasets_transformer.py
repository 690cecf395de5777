"""Producer of the 8-tuple batches the autoregressive Transformer-TTS trainer consumes (single-speaker case).

Contract of the reference's ``datasets/datasets_transformer.py`` (``TrainDatasets`` :18-103, ``collate_fn`` :335-383, ``_pad_mel`` /
``_pad_stop_token`` :410-429, ``LengthsBatchSampler`` :431-490, ``NumBatchSampler`` :492-522), which differs from the FastSpeech2
reader in four ways the decoder depends on:
  * every mel starts with an ALL-ZERO "go" frame (:93): it is what synthesis feeds first, and frame 0 of the real mel is predicted;
  * ``mel_length`` (and with it ``pos_mel`` and the frame padding of a batch) is rounded UP to a multiple of ``hp.reduction_rate``
    (:96, :418, :426), so that ``mel[:, :-r:r]`` gives exactly (T - r) / r decoder steps whose r-frame outputs cover mel[:, r:];
  * mel frames are padded with -5.0 when the corpus is not mean/variance normalised (no ``hp.mean_file`` / ``hp.var_file``), -0.5
    when it is (:412-415); the stop token is 0 on real frames (the go frame included) and 1.0 on padding;
  * a batch is sorted by mel length, longest first (:347-354; Python's stable sort, so equal lengths keep their order).
Script format: one ``<mel.npy>|<space separated ids>`` line per utterance.  Speaker embeddings, sentencepiece and the .htk / .mel
containers are outside the accelerated path (the constructor refuses them).
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset, Sampler


def _round_up(x, multiple):
    return -(-int(x) // int(multiple)) * int(multiple)


class TrainDatasets(Dataset):
    def __init__(self, csv_file, hp):
        assert not getattr(hp, "is_multi_speaker", False) and getattr(hp, "spm_model", None) is None, \
            "speaker embeddings / sentencepiece are outside the accelerated path"
        self.hp = hp
        self.items = []
        with open(csv_file) as f:
            for line in f:
                line = line.rstrip("\n")
                if line:
                    name, text = line.split("|")[:2]
                    self.items.append((name, text.strip()))
        self.mean_value = self.var_value = None
        if getattr(hp, "mean_file", None) is not None and getattr(hp, "var_file", None) is not None:
            self.mean_value = np.load(hp.mean_file).reshape(-1, hp.mel_dim)
            self.var_value = np.load(hp.var_file).reshape(-1, hp.mel_dim)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        mel_name, text = self.items[idx]
        if not mel_name.endswith(".npy"):
            raise ValueError(f"{mel_name}: only .npy mel files are read on this path")
        ids = np.array([int(t) for t in text.split(" ")], dtype=np.int32)
        mel = np.load(mel_name)
        assert mel.shape[0] == self.hp.mel_dim or mel.shape[1] == self.hp.mel_dim, f"{mel_name} does not have strange shape {mel.shape}"
        if mel.shape[1] != self.hp.mel_dim:
            mel = mel.reshape(-1, self.hp.mel_dim)
        if self.mean_value is not None:
            mel = (mel - self.mean_value) / np.sqrt(self.var_value)
        mel = np.concatenate([np.zeros((1, self.hp.mel_dim), np.float32), mel.astype(np.float32, copy=False)], axis=0)      # go frame
        frames = mel.shape[0]
        mel_length = _round_up(frames, self.hp.reduction_rate)
        return dict(text=ids, text_length=len(ids), mel_input=mel, mel_length=mel_length, pos_mel=np.arange(1, mel_length + 1),
                    pos_text=np.arange(1, len(ids) + 1), stop_token=np.zeros(frames, np.float32), spk_emb=None)


def _pad_rows(xs, value=0):
    n = max(len(x) for x in xs)
    return np.stack([np.pad(np.asarray(x), (0, n - len(x)), constant_values=value) for x in xs])


def make_collate_fn(hp):
    """the reference's ``collate_fn`` reads its module-level hparams; here the hparams are bound explicitly"""
    r = int(hp.reduction_rate)
    normalised = not (getattr(hp, "mean_file", None) is None and getattr(hp, "var_file", None) is None)
    mel_pad = -0.5 if normalised else -5.0

    def collate_fn(batch):
        order = sorted(range(len(batch)), key=lambda i: batch[i]["mel_length"], reverse=True)      # stable: ties keep their order
        b = [batch[i] for i in order]
        T = _round_up(max(d["mel_input"].shape[0] for d in b), r)
        mel = np.stack([np.pad(d["mel_input"], [[0, T - d["mel_input"].shape[0]], [0, 0]], constant_values=mel_pad) for d in b])
        stop = np.stack([np.pad(d["stop_token"], (0, T - d["stop_token"].shape[0]), constant_values=1.0) for d in b])
        return (torch.from_numpy(_pad_rows([d["text"] for d in b]).astype(np.int64)), torch.from_numpy(mel.astype(np.float32)),
                torch.from_numpy(_pad_rows([d["pos_text"] for d in b]).astype(np.int64)),
                torch.from_numpy(_pad_rows([d["pos_mel"] for d in b]).astype(np.int64)),
                torch.tensor([d["text_length"] for d in b], dtype=torch.int64), torch.tensor([d["mel_length"] for d in b], dtype=torch.int64),
                torch.from_numpy(stop.astype(np.float32)), None)
    return collate_fn


class LengthsBatchSampler(Sampler):
    """Variable-size batches under a budget of mel frames (reference :431-490): utterances are taken in script order until the next
    one would exceed ``n_lengths``; the scan stops once at most one utterance is left (the reference's ``count + 1 < len`` loop), and
    an utterance longer than the budget yields an empty batch exactly as it does there.  The lengths come from ``lengths_file``
    (written on first use: the padded length of every utterance collated alone)."""

    def __init__(self, dataset, n_lengths, lengths_file=None, shuffle=True, shuffle_one_time=False, reverse=False):
        assert not (shuffle and reverse), "shuffle and reverse cannot set True at the same time."
        if lengths_file is None or not os.path.exists(lengths_file):
            collate = make_collate_fn(dataset.hp)
            self.lengths_np = np.array([collate([dataset[i]])[1].shape[1] for i in range(len(dataset))])
            if lengths_file is not None:
                np.save(lengths_file, self.lengths_np)
        else:
            self.lengths_np = np.load(lengths_file)
            assert len(dataset) == len(self.lengths_np), f"mismatch the number of lines between dataset and {lengths_file}"
        self.n_lengths = n_lengths
        self.all_indices = self._batch_indices()
        if shuffle_one_time:
            random.shuffle(self.all_indices)
        self.shuffle, self.shuffle_one_time, self.reverse = shuffle, shuffle_one_time, reverse

    def _batch_indices(self):
        n, count, out = len(self.lengths_np), 0, []
        while count + 1 < n:
            indices, total = [], 0
            while count < n and total + self.lengths_np[count] <= self.n_lengths:
                total += self.lengths_np[count]
                indices.append(count)
                count += 1
            out.append(indices)
            if not indices:          # one utterance above the budget: the reference would spin here; stop instead
                break
        return out

    def __iter__(self):
        if self.shuffle and not self.shuffle_one_time:
            random.shuffle(self.all_indices)
        if self.reverse:
            self.all_indices.reverse()
        return iter(self.all_indices)

    def __len__(self):
        return len(self.all_indices)


class NumBatchSampler(Sampler):
    """Fixed-size batches in script order plus the ragged tail, batch ORDER shuffled at construction and at every epoch with the
    numpy global RNG (reference :492-522)."""

    def __init__(self, dataset, batch_size, drop_last=True, shuffle=True):
        n = len(dataset)
        full = n - n % batch_size
        self.all_indices = np.arange(full).reshape(-1, batch_size).tolist()
        if n % batch_size:
            self.all_indices.append(list(range(full, n)))
        self.batch_size, self.drop_last, self.shuffle = batch_size, drop_last, shuffle
        if shuffle:
            np.random.shuffle(self.all_indices)

    def __iter__(self):
        if self.shuffle:
            np.random.shuffle(self.all_indices)
        return iter(self.all_indices)

    def __len__(self):
        return len(self.all_indices)

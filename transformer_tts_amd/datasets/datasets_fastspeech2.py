"""Producer of the 16-tuple batches the training path consumes (single-speaker FastSpeech2 case).

Mirrors the on-disk format and the collate contract of the reference's
``datasets/datasets_fastspeech2.py`` (``TrainDatasets`` :19-174, ``collate_fn`` :521-616, ``_pad_mel``
:728-739, ``LengthsBatchSampler`` :749-813, ``NumBatchSampler`` :815-845, ``DistributedSamplerWrapper`` :847-890): a script file with
one ``<mel.npy>|<space separated ids>`` line per utterance and sibling ``*_alignment.npy`` /
``*_f0.npy`` / ``*_energy.npy`` files; mel pad value -0.5, stop-token pad 1.0, everything else 0.
``TestDatasets`` / ``collate_fn_test`` (:326-411, :469-519) feed the synthesis script ``test_fastspeech2.py``.
Speaker / accent / gender / hop / sentencepiece options are outside the accelerated path.
"""
import numpy as np
import torch
from torch.utils.data import Dataset, Sampler


class TrainDatasets(Dataset):
    def __init__(self, csv_file, hp, alignment_pred=True, pitch_pred=True, energy_pred=True, accent_emb=False):
        assert not accent_emb and not getattr(hp, "is_multi_speaker", False) and getattr(hp, "spm_model", None) is None
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
        self.pred_alignment, self.pred_f0, self.pred_energy = alignment_pred, pitch_pred, energy_pred

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        mel_name, text = self.items[idx]
        tail = getattr(self.hp, "tail_alignment", "_alignment") + ".npy"
        ids = np.array([int(t) for t in text.split(" ")], dtype=np.int32)
        mel = np.load(mel_name)
        assert mel.shape[0] == self.hp.mel_dim or mel.shape[1] == self.hp.mel_dim, \
            f"{mel_name} does not have strange shape {mel.shape}"
        if mel.shape[1] != self.hp.mel_dim:
            mel = mel.reshape(-1, self.hp.mel_dim)
        mel = mel.astype(np.float32)
        if self.mean_value is not None:
            mel = (mel - self.mean_value) / np.sqrt(self.var_value)
        return dict(text=ids, text_length=len(ids), mel_input=mel, mel_length=mel.shape[0],
                    pos_mel=np.arange(1, mel.shape[0] + 1), pos_text=np.arange(1, len(ids) + 1),
                    stop_token=np.zeros(mel.shape[0], np.float32),
                    alignment=np.load(mel_name.replace(".npy", tail)) if self.pred_alignment else None,
                    f0=np.load(mel_name.replace(".npy", "_f0.npy")) if self.pred_f0 else None,
                    energy=np.load(mel_name.replace(".npy", "_energy.npy")) if self.pred_energy else None,
                    mel_name=mel_name)


class TestDatasets(Dataset):
    """Synthesis script reader (reference ``TestDatasets`` :326-411, single-speaker case): one
    ``<output name>|<space separated ids>`` line per utterance; no acoustic files are read."""

    def __init__(self, csv_file, hp, accent_emb=False):
        assert not accent_emb and not getattr(hp, "is_multi_speaker", False) and getattr(hp, "spm_model", None) is None \
            and not getattr(hp, "use_hop", False) and not getattr(hp, "gender_emb", False), \
            "speaker / accent / gender / hop / sentencepiece options are outside the accelerated path"
        self.hp = hp
        self.items = []
        with open(csv_file) as f:
            for line in f:
                line = line.rstrip("\n")
                if line:
                    name, text = line.split("|")[:2]
                    self.items.append((name, text.strip()))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        mel_output, text = self.items[idx]
        ids = np.array([int(t) for t in text.split(" ")], dtype=np.int32)
        return dict(text=ids, text_length=len(ids), mel_output=mel_output, pos_text=np.arange(1, len(ids) + 1))


def collate_fn_test(batch):
    """the 9-tuple of the reference's ``collate_fn_test`` (:469-519): (text, names, pos_text, text_length, spk_emb,
    accent, gender, spk_emb_postprocess, hop_size) with the optional conditioning slots None"""
    text = _pad1([d["text"] for d in batch], dtype=np.int32)
    pos_text = _pad1([d["pos_text"] for d in batch], dtype=np.int32)
    return (torch.LongTensor(text), [d["mel_output"] for d in batch], torch.LongTensor(pos_text),
            torch.LongTensor([d["text_length"] for d in batch]), None, None, None, None, None)


def _pad1(xs, value=0, dtype=None):
    n = max(len(x) for x in xs)
    out = np.stack([np.pad(np.asarray(x), (0, n - len(x)), constant_values=value) for x in xs])
    return out.astype(dtype) if dtype is not None else out


def collate_fn(batch):
    """list of TrainDatasets samples -> the reference's 16-tuple (single-speaker return, reference :613)."""
    T = max(d["mel_input"].shape[0] for d in batch)
    mel = np.stack([np.pad(d["mel_input"], [[0, T - d["mel_input"].shape[0]], [0, 0]], constant_values=-0.5)
                    for d in batch])
    f0 = torch.from_numpy(_pad1([d["f0"] for d in batch], dtype=np.float32)) if batch[0]["f0"] is not None else None
    energy = torch.from_numpy(_pad1([d["energy"] for d in batch], dtype=np.float32)) \
        if batch[0]["energy"] is not None else None
    align = torch.from_numpy(_pad1([d["alignment"] for d in batch], dtype=np.int64)) \
        if batch[0]["alignment"] is not None else None
    return (torch.from_numpy(_pad1([d["text"] for d in batch], dtype=np.int64)), torch.from_numpy(mel.astype(np.float32)),
            torch.from_numpy(_pad1([d["pos_text"] for d in batch], dtype=np.int64)),
            torch.from_numpy(_pad1([d["pos_mel"] for d in batch], dtype=np.int64)),
            torch.tensor([d["text_length"] for d in batch], dtype=torch.int64),
            torch.tensor([d["mel_length"] for d in batch], dtype=torch.int64),
            torch.from_numpy(_pad1([d["stop_token"] for d in batch], value=1.0, dtype=np.float32)),
            None, f0, energy, align, None, None, None, [d["mel_name"] for d in batch], [None] * len(batch))


class LengthsBatchSampler(Sampler):
    """Variable-size batches under a budget of padded mel frames (reference :749-813): utterances are taken in
    dataset order and a batch is closed when  max(mel length in the batch) * (batch size + 1)  would exceed
    ``n_lengths``; the ORDER of the batches is shuffled every epoch (or once, or reversed).  ``lengths_file`` holds
    the mel length of every utterance (.npy); when it does not exist it is built from the dataset and saved to
    ``hp.lengths_file``.  As in the reference the last utterance of the list is never started as a batch of its own."""

    def __init__(self, dataset, n_lengths, hp, lengths_file=None, shuffle=True, shuffle_one_time=False, reverse=False):
        import os
        import random
        assert not ((shuffle == reverse) and shuffle is True), "shuffle and reverse cannot set True at the same time."
        if lengths_file is None or not os.path.exists(lengths_file):
            print("lengths_file is not exists. Make...")
            self.lengths_np = np.array([dataset[i]["mel_length"] for i in range(len(dataset))])
            np.save(hp.lengths_file, self.lengths_np)
        else:
            print("{} is loading.".format(lengths_file))
            self.lengths_np = np.load(lengths_file)
            assert len(dataset) == len(self.lengths_np), \
                "mismatch the number of lines between dataset and {}".format(lengths_file)
        self.n_lengths = n_lengths
        self._random = random
        self.all_indices = self._batch_indices()
        if shuffle_one_time:
            random.shuffle(self.all_indices)
        self.shuffle, self.shuffle_one_time, self.reverse = shuffle, shuffle_one_time, reverse

    def _batch_indices(self):
        count, n = 0, len(self.lengths_np)
        all_indices = []
        while count + 1 < n:
            indices, max_len = [], 0
            while count < n:
                curr_len = int(self.lengths_np[count])
                if max(max_len, curr_len) * (len(indices) + 1) > self.n_lengths:
                    break
                max_len = max(max_len, curr_len)
                indices.append(count)
                count += 1
            assert indices, f"utterance {count} alone ({int(self.lengths_np[count])} frames) exceeds max_seqlen={self.n_lengths}"
            all_indices.append(indices)
        return all_indices

    def __iter__(self):
        if self.shuffle and not self.shuffle_one_time:
            self._random.shuffle(self.all_indices)
        if self.reverse:
            self.all_indices.reverse()
        for indices in self.all_indices:
            yield indices

    def __len__(self):
        return len(self.all_indices)


class NumBatchSampler(Sampler):
    """Fixed-size batches of consecutive indices, batch ORDER shuffled (reference :815-845)."""

    def __init__(self, dataset, batch_size, drop_last=True, shuffle=True):
        n = len(dataset)
        full = n - n % batch_size
        self.all_indices = np.arange(full).reshape(-1, batch_size).tolist()
        if n % batch_size:
            self.all_indices.append(list(range(full, n)))
        self.shuffle = shuffle
        if shuffle:
            np.random.shuffle(self.all_indices)

    def __iter__(self):
        if self.shuffle:
            np.random.shuffle(self.all_indices)
        yield from self.all_indices

    def __len__(self):
        return len(self.all_indices)


class DistributedSamplerWrapper(Sampler):
    """Shard the LIST OF BATCHES of a batch sampler over the ranks (reference :847-919, a
    ``torch.utils.data.DistributedSampler`` over the batch list with its default ``shuffle=True, seed=0``): the
    list is materialised once per epoch, permuted with ``torch.randperm`` under a generator seeded ``seed + epoch``,
    padded by repetition to a multiple of the world size, and rank r takes entries r, r+world, ...  The reference
    never calls ``set_epoch``, so the permutation (not the batch contents) is the same every epoch."""

    def __init__(self, sampler, num_replicas=None, rank=None, shuffle=True, seed=0):
        import torch.distributed as dist
        self.sampler = sampler
        self.num_replicas = num_replicas if num_replicas is not None else dist.get_world_size()
        self.rank = rank if rank is not None else dist.get_rank()
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        batches = list(self.sampler)
        n = len(batches)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        total = -(-n // self.num_replicas) * self.num_replicas
        while len(order) < total:                       # pad by repetition (more than once if world > 2 n)
            order += order[: total - len(order)]
        yield from (batches[i] for i in order[self.rank:total:self.num_replicas])

    def __len__(self):
        return -(-len(self.sampler) // self.num_replicas)


def write_synthetic_corpus(root, n_utt=16, seed=1234, vocab=152, mel_dim=80):
    """BASELINE.json configs[0] data: 16 synthetic utterances in the on-disk format above
    (L~U{8..23}, dur~U{1..8}, mel N(0,1), f0 U(71,799.8), energy U(0,403.8))."""
    import os
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    lines, lengths = [], []
    for u in range(n_utt):
        L = int(rng.integers(8, 24))
        dur = rng.integers(1, 9, size=L)
        T = int(dur.sum())
        base = os.path.join(root, f"utt{u:03d}")
        np.save(base + ".npy", rng.standard_normal((T, mel_dim)).astype(np.float32))
        np.save(base + "_alignment.npy", dur.astype(np.int64))
        np.save(base + "_f0.npy", rng.uniform(71.0, 799.8, size=T).astype(np.float32))
        np.save(base + "_energy.npy", rng.uniform(0.0, 403.8, size=T).astype(np.float32))
        lines.append(base + ".npy|" + " ".join(str(int(i)) for i in rng.integers(1, vocab, size=L)))
        lengths.append(T)
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    np.save(os.path.join(root, "lengths.npy"), np.asarray(lengths))
    return os.path.join(root, "train.txt")

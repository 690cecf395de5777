"""Checkpoint averaging tool (reference average_checkpoints.py:9-45)."""
import os
import time

import torch

from transformer_tts_amd import average_checkpoints as A


def _ckpt(path, scale):
    sd = {"layer.weight": torch.full((3, 2), float(scale)), "layer.bias": torch.arange(3.0) * scale,
          "bn.num_batches_tracked": torch.tensor(int(scale), dtype=torch.int64)}
    torch.save(sd, path)


def test_average_by_epoch_range_and_by_mtime(tmp_path):
    for e in range(1, 5):
        _ckpt(tmp_path / f"network.epoch{e}", e)
        os.utime(tmp_path / f"network.epoch{e}", (time.time() + e, time.time() + e))
    snaps = [str(tmp_path / f"network.epoch{e}") for e in range(1, 5)]
    A.main(["--snapshots", *snaps, "--out", str(tmp_path / "avg_range"), "--start", "2", "--end", "4"])
    avg = torch.load(tmp_path / "avg_range", weights_only=True)
    assert torch.allclose(avg["layer.weight"], torch.full((3, 2), 3.0)) and torch.allclose(avg["layer.bias"], torch.arange(3.0) * 3)
    assert avg["bn.num_batches_tracked"].dtype.is_floating_point and float(avg["bn.num_batches_tracked"]) == 3.0
    A.main(["--snapshots", *snaps, "--out", str(tmp_path / "avg_last"), "--num", "2"])
    avg = torch.load(tmp_path / "avg_last", weights_only=True)
    assert torch.allclose(avg["layer.weight"], torch.full((3, 2), 3.5))

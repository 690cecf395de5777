"""Diagnostic: poison every torch.empty* allocation with NaN and report where NaNs surface in one fp32 train step."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import torch

_empty, _empty_like = torch.empty, torch.empty_like


def poison(t):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(-123456)
        elif t.dtype == torch.uint8:
            t.fill_(77)
    return t


torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
_new_empty = torch.Tensor.new_empty
torch.Tensor.new_empty = lambda self, *a, **k: poison(_new_empty(self, *a, **k))

from helpers import product_model, batch_to   # noqa: E402
from transformer_tts_amd import synthetic     # noqa: E402
from transformer_tts_amd.optim import FusedAdam  # noqa: E402
from transformer_tts_amd.train_fastspeech2 import train_step  # noqa: E402

for amp in (False, True):
    model, hp, _ = product_model("small", amp=amp, dropout=0.1, device="cuda")
    opt = FusedAdam(model)
    for i in range(2):
        b = batch_to(synthetic.make_batch(100 + i, 4, l_range=(9, 20), dur_range=(1, 9), vocab=60), "cuda")
        out = train_step(model, opt, 4000 + i, b, hp)
        print("amp", amp, "step", i, "loss", float(out[0]), flush=True)
        for n, p in model.named_parameters():
            if not torch.isfinite(p).all():
                print("   non-finite param after step:", n, int((~torch.isfinite(p)).sum()), "/", p.numel())

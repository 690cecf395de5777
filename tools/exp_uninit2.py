"""Diagnostic: growth of run-to-run differences (different float-atomic orders) over training steps."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import torch
from helpers import product_model, batch_to
from transformer_tts_amd import synthetic
from transformer_tts_amd.optim import FusedAdam
from transformer_tts_amd.train_fastspeech2 import train_step
from transformer_tts_amd.Models import functional


def run():
    functional._site_counter[0] = 1000
    model, hp, _ = product_model("small", amp=False, dropout=0.1, device="cuda")
    opt = FusedAdam(model)
    snaps = []
    for i in range(6):
        b = batch_to(synthetic.make_batch(100 + (i % 2), 4, l_range=(9, 20), dur_range=(1, 9), vocab=60), "cuda")
        out = train_step(model, opt, 4000 + i, b, hp)
        snaps.append((float(out[0].detach()), opt.arena.g.clone(), opt.arena.p.clone()))
    return snaps, model


a, ma = run()
b, mb = run()
base = next(mb.parameters()).data_ptr()
spans = [(n, (p.data_ptr() - base) // 4, p.numel()) for n, p in mb.named_parameters()]
for i in range(6):
    print(f"step {i}: loss {a[i][0]!r} vs {b[i][0]!r}")
    rows = []
    for n, o, k in spans:
        ga, gb = a[i][1][o:o + k], b[i][1][o:o + k]
        pa, pb = a[i][2][o:o + k], b[i][2][o:o + k]
        gd = float((ga - gb).abs().max()) / (float(ga.abs().max()) + 1e-30)
        pd = float((pa - pb).abs().max())
        rows.append((pd, gd, n, float(ga.abs().max())))
    rows.sort(reverse=True)
    for pd, gd, n, gm in rows[:8]:
        print(f"    {n:50s} param maxdiff {pd:.2e}  grad rel maxdiff {gd:.2e} (grad max {gm:.2e})")

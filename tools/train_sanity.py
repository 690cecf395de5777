"""Full-size numerics sanity: 400 graph-replayed steps of the config-2 model on a pool of 8 synthetic batches at the peak of the
Noam schedule; prints the loss terms every 50 steps (must stay finite and fall)."""
import sys, torch
sys.path.insert(0, ".")
import bench
from transformer_tts_amd import ops, synthetic
from transformer_tts_amd.optim import FusedAdam
from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep, build_model
from transformer_tts_amd.utils.utils import init_weight
dev = torch.device("cuda", 0)
ops.lib()
hp = bench.bench_hp(amp=True, workload="cfg2", fp8=False, return_attn=False)
torch.manual_seed(1234)
model = build_model(hp); model.apply(init_weight); model.train(); model = model.to(dev)
opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0)
pool = [tuple(b.to(dev) if torch.is_tensor(b) else b for b in synthetic.benchmark_batch(2024 + i, hp.batch_size)) for i in range(8)]
st = GraphedTrainStep(model, opt, hp)
hist = []
for i in range(400):
    loss, parts, _ = st(4000 + i, pool[i % 8])
    if i % 50 == 0 or i == 399:
        row = {k: round(float(v), 4) for k, v in parts.items()}
        print(i, round(float(loss), 4), row, flush=True)
        hist.append(float(loss))
assert all(x == x and abs(x) < 1e9 for x in hist), hist
assert hist[-1] < hist[0], hist
print("ok: finite and falling")

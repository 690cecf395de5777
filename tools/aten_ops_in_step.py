"""Which torch (aten) operators still launch device work inside one eager train step (config 2): torch.profiler table + the Python
call sites of copy_/fill_/add/zeros (the product's arithmetic is in libfs2_hip.so; these are leftovers to remove or to justify)."""
import sys, collections, traceback
sys.path[:0] = [".", "tests", "tests/golden"]
import torch
import bench
from transformer_tts_amd import synthetic
from transformer_tts_amd.optim import FusedAdam
from transformer_tts_amd.train_fastspeech2 import build_model, train_step
from transformer_tts_amd.utils.utils import init_weight

hp = bench.bench_hp()
torch.manual_seed(1234)
model = build_model(hp); model.apply(init_weight); model.train(); model = model.cuda()
opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0)
batch = tuple(b.cuda() if torch.is_tensor(b) else b for b in synthetic.benchmark_batch(2024, 48))
for s in range(3):
    train_step(model, opt, 1 + s, batch, hp)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(model, opt, 5, batch, hp)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::zeros", "aten::sum", "aten::ne", "aten::eq", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::mul", "aten::div") and ev.device_time_total > 0 or ev.name in ("aten::copy_",):
        st = [f for f in (ev.stack or []) if "transformer_tts_amd" in f or "bench.py" in f]
        sites[(ev.name, st[0] if st else "?")] += 1
for (name, site), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(n, name, site)

"""pure graph replay period vs the stepper's period (is there a host-made bubble between two replays?)"""
import sys, time, torch
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
pool = [tuple(b.to(dev) if torch.is_tensor(b) else b for b in synthetic.benchmark_batch(2024 + i, hp.batch_size)) for i in range(2)]
st = GraphedTrainStep(model, opt, hp)
step = 1
for i in range(8):
    st(step, pool[i % 2]); step += 1
torch.cuda.synchronize()
N = 60
t0 = time.perf_counter()
for i in range(N):
    st(step, pool[i % 2]); step += 1
torch.cuda.synchronize()
print("stepper period ms", (time.perf_counter() - t0) / N * 1e3)
graphs = [v[0] for v in st.graphs.values()]
t0 = time.perf_counter()
for i in range(N):
    graphs[i % len(graphs)].replay()
torch.cuda.synchronize()
print("bare replay period ms", (time.perf_counter() - t0) / N * 1e3, "graphs", len(graphs))

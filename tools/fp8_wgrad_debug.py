"""which weight-gradient products of a configs[4] step run on the fp8 copies (diagnostics)"""
import sys, collections, torch
sys.path.insert(0, ".")
import bench
from transformer_tts_amd import ops, synthetic
from transformer_tts_amd.train_fastspeech2 import build_model, compute_losses, create_masks

hp = bench.bench_hp(amp=True, workload="cfg4", fp8=True)
torch.manual_seed(0)
model = build_model(hp).cuda().train()
orig = ops._wgrad_fp8
stat = collections.Counter()
def spy(g, dy, x):
    r = orig(g, dy, x)
    a, b = getattr(dy, "_fs2_q8", None), getattr(x, "_fs2_q8", None)
    why = "fp8" if r is not None else ("no dy copy" if a is None else "no x copy" if b is None else f"formats {a[2]} {b[2]}" if (not a[2] or b[2]) else "plan/shape")
    stat[(g.M, g.N, g.K, g.conv, g.batch1 * max(1, g.batch2), why)] += 1
    return r
ops._wgrad_fp8 = spy
batch = [t.cuda() if torch.is_tensor(t) else t for t in synthetic.make_batch(2024, 64)]
text, mel, pos_text, pos_mel, _, _, _, _, f0, energy, align = batch[:11]
src_mask, mel_mask = create_masks(pos_text, pos_mel, task="fastspeech2")
out = model(text, src_mask, mel_mask, align, f0, energy)
total, parts = compute_losses(hp, out, mel, align, f0, energy)
total.backward()
torch.cuda.synchronize()
for k, v in sorted(stat.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
    print(v, k)

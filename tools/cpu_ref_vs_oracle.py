"""the oracle (CPU restatement, what bench.py times as cpu_baseline, kind "port") against the IMPORTED
reference on the same config-2 batch, same thread count, in the build container (the reference never travels to the GPU box).
Both run the reference's dropout rates; one warm-up step, then the median of N timed train steps each (fwd + bwd + clip + Adam).

    python tools/cpu_ref_vs_oracle.py [steps] [threads]
"""
import contextlib
import io
import os
import runpy
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ns = runpy.run_path(os.path.join(ROOT, "tests", "golden", "make_golden.py"), run_name="recipe")      # import shims + reference by path
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from golden_configs import CONFIGS  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.set_num_threads(threads)
    cfg = CONFIGS["bench"]
    batch = cfg["batch"]()
    frames = int(batch[5].sum())
    # ---- the reference's own model and train_loop at the dropout rates its trainer builds it with (train_fastspeech2.py:381-389:
    #      hp.dropout 0.1, dropout_postnet 0.5, hp.dropout_variance_adaptor 0.5; a rate of 0 would skip F.dropout's mask altogether)
    T = ns["ref_trainer"]()
    model, hp, _ = ns["build_reference"](cfg, dropouts=(0.1, 0.5, 0.5))
    hp.amp = False
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    times = []
    step = cfg["start_step"]
    for i in range(steps + 1):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            step = T.train_loop(model, opt, step, 0, SimpleNamespace(n_gpus=0), hp, 1, [batch])
        times.append(time.perf_counter() - t0)
        print(f"reference step {i}: {times[-1]:.1f} s", flush=True)
    ref = sorted(times[1:])[len(times[1:]) // 2]
    # ---- the oracle, as bench.py's cpu_baseline runs it
    from oracle import train as otrain
    from oracle.model import FastSpeech2 as OracleFS2
    sys.path.insert(0, os.path.join(ROOT))
    import bench
    ohp = bench.bench_hp(amp=False)
    torch.manual_seed(0)
    m = OracleFS2.from_hp(ohp, dropout=ohp.dropout, dropout_postnet=0.5, dropout_variance_adaptor=ohp.dropout_variance_adaptor)
    m.train()
    oopt = otrain.make_optimizer(m)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        otrain.train_step(m, oopt, 2 + i, batch, ohp.d_model_decoder)
        times.append(time.perf_counter() - t0)
        print(f"oracle step {i}: {times[-1]:.1f} s", flush=True)
    orc = sorted(times[1:])[len(times[1:]) // 2]
    print(f"config-2 batch ({frames} valid mel frames), {threads} threads: reference {ref:.2f} s/step ({frames / ref:.0f} frames/s), "
          f"oracle {orc:.2f} s/step ({frames / orc:.0f} frames/s), oracle / reference = {orc / ref:.3f}")


if __name__ == "__main__":
    main()

"""Host-side cost of ONE eager train step (the Python launch path) without a GPU: libfs2_hip.so is replaced by a stub whose entry
points return at once, tensors live on the CPU and are never computed on.  What remains is exactly the work the training thread does
per step in eager mode -- descriptor filling, allocation, autograd bookkeeping, ctypes calls -- under cProfile.
    FS2_WGRAD_SLICED=0 python tools/host_profile.py [n_steps]"""
import cProfile
import ctypes
import os
import pstats
import sys
import time

os.environ.setdefault("FS2_WGRAD_SLICED", "0")
sys.path[:0] = [".", "tests", "tests/golden"]
import torch  # noqa: E402

import bench  # noqa: E402
from transformer_tts_amd import ops, synthetic  # noqa: E402


class _Fn:
    def __init__(self, name):
        self.name = name
        self.calls = 0

    def __call__(self, *a):
        self.calls += 1
        if self.name == "fs2_flash_attn_keep_words":
            B, H, t = a
            return B * H * ((t + 63) // 64) * t * 4
        if self.name == "fs2_flash_attn_keep_words_rect":
            B, H, tq, tk = a
            return B * H * ((tk + 63) // 64) * tq * 4
        if self.name == "fs2_gemm_last_splits":
            return 5
        if self.name == "fs2_gemm_last_tile":
            return 192
        if self.name == "fs2_last_error":
            return b""
        return 0


class _Lib:
    def __init__(self):
        self._fns = {}

    def __getattr__(self, name):
        f = self._fns.get(name)
        if f is None:
            f = self._fns[name] = _Fn(name)
        return f


stub = _Lib()
ops.lib = lambda: stub
ops._p = lambda t: None if t is None else t.data_ptr()
ops._stream = lambda: 0
ops.Rng.__init__ = lambda self, seed, device: setattr(self, "state", torch.tensor([seed, 0], dtype=torch.int64))


def main():
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import build_model, train_step
    from transformer_tts_amd import train_fastspeech2 as TF
    TF.DEVICE = torch.device("cpu")
    hp = bench.bench_hp()
    torch.manual_seed(1234)
    model = build_model(hp)
    model.train()
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0)
    batch = synthetic.benchmark_batch(2024, 48)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    for s in range(2):
        train_step(model, opt, 1 + s, batch, hp)
    t0 = time.perf_counter()
    for s in range(n):
        train_step(model, opt, 10 + s, batch, hp)
    dt = (time.perf_counter() - t0) / n
    calls = sum(f.calls for f in stub._fns.values())
    print(f"host time per eager step: {dt * 1e3:.2f} ms ({calls // (n + 2)} library calls per step)")
    pr = cProfile.Profile()
    pr.enable()
    for s in range(n):
        train_step(model, opt, 20 + s, batch, hp)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()

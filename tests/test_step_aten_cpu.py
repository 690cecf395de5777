"""Which torch (aten) arithmetic does the host composition of one train step issue by ITSELF?  On the MI355X every such call is a
launch of a torch kernel inside a step whose arithmetic belongs to libfs2_hip.so (fills, adds, compares, reduces).
The step runs on CPU tensors with the HIP ops replaced by the oracle primitives (tests/fake_backend.py); a TorchDispatchMode records
every aten call that moves or computes data together with the innermost Python frame that is NOT the oracle's (the oracle's own torch
arithmetic stands for kernel work and is ignored; autograd-engine calls have no Python frame and are attributed to "<autograd>")."""
import collections
import os
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from fake_backend import fake_ops  # noqa: F401
from helpers import CONFIGS, product_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# metadata-only / allocation-only aten calls: no device work
FREE = {"view", "_unsafe_view", "reshape", "expand", "permute", "transpose", "t", "unsqueeze", "squeeze", "slice", "select", "as_strided",
        "detach", "alias", "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "unbind", "split", "split_with_sizes",
        "narrow", "view_as", "_local_scalar_dense", "is_same_size", "lift_fresh", "unfold", "result_type", "sym_size", "sym_stride",
        "sym_numel", "stride", "size", "numel", "dim", "is_contiguous", "_reshape_alias", "expand_as", "resize_", "set_", "_to_copy_meta"}


class Recorder(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in FREE:
            return out
        site = "<autograd>"
        for f in reversed(traceback.extract_stack()[:-1]):
            fn = f.filename
            if "/oracle/" in fn:
                return out                      # kernel work restated by the oracle: not the host composition's
            if "/transformer_tts_amd/" in fn:
                site = f"{os.path.relpath(fn, ROOT)}:{f.lineno}"
                break
        self.sites[(name, site)] += 1
        return out


def traced_step(name="tiny"):
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import train_step
    model, hp, _ = product_model(name)
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0)
    batch = CONFIGS[name]["batch"]()
    train_step(model, opt, 1, batch, hp)            # first step: one-off allocations (gradient views, caches)
    rec = Recorder()
    with rec:
        train_step(model, opt, 2, batch, hp)
    return rec.sites


# What a steady-state step may still issue through torch: the asynchronous upload of the four Adam hyper-parameters (one tiny copy,
# outside the captured graph).  Everything else -- masks, zero fills, gradient fan-in adds, loss sums, the backward's seed gradient --
# runs in the product's own kernels.
# (`ne`: create_masks on CPU tensors is the reference's torch expression; on the GPU it is the one-launch fs2_pad_mask_info --
#  tests/test_kernels_gpu.py::test_pad_mask_info_is_create_masks_plus_mask_info, and tools/aten_ops_in_step.py lists what a real
#  step still launches through torch.)
ALLOWED = {("copy_", "transformer_tts_amd/optim.py"), ("ne", "transformer_tts_amd/train_fastspeech2.py")}


def test_the_step_issues_no_torch_arithmetic_of_its_own(fake_ops):
    sites = traced_step()
    extra = {k: n for k, n in sites.items() if (k[0], k[1].split(":")[0]) not in ALLOWED}
    assert not extra, "aten calls issued by the host composition of a train step:\n" + "\n".join(f"  {n} x {k[0]} at {k[1]}" for k, n in sorted(extra.items()))


if __name__ == "__main__":
    import sys
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import inspect
    from oracle import primitives
    from transformer_tts_amd import ops
    for n_, fn_ in inspect.getmembers(primitives, inspect.isfunction):
        if not n_.startswith("_") and hasattr(ops, n_):
            setattr(ops, n_, fn_)
    ops.Rng = primitives.Rng
    for k, n in sorted(traced_step(sys.argv[1] if len(sys.argv) > 1 else "tiny").items(), key=lambda kv: kv[0][1]):
        print(n, k[0], k[1])

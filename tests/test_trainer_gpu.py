"""BASELINE.json configs[0] on the PRODUCT: the dataset -> DataLoader -> train_epoch -> checkpoint -> resume route of
the shipped trainer on the HIP backend (tests/test_trainer_cpu.py runs the same route on the oracle-backed test seam,
which checks host logic only), the graph-replay fast path of train_loop against eager launches, and the data-parallel
path on a one-rank RCCL group."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import CONFIGS, batch_to, product_model
from test_trainer_cpu import SMALL_HP

pytestmark = pytest.mark.gpu


def _hp(tmp_path, extra=""):
    from transformer_tts_amd.datasets import datasets_fastspeech2 as D
    from transformer_tts_amd.utils import HParams
    from transformer_tts_amd.utils.utils import fill_variables
    script = D.write_synthetic_corpus(str(tmp_path / "synthetic16"), n_utt=16)
    hp_file = tmp_path / "hparams.py"
    save = str(tmp_path / "ckpt")
    hp_file.write_text(SMALL_HP.format(save=save, script=script) + extra)
    hp = HParams()
    hp.configure(hp_file)
    fill_variables(hp, verbose=False)
    os.makedirs(save, exist_ok=True)
    return hp, save


def _losses(out):
    return [float(l.split("=")[1]) for l in out.splitlines() if l.startswith("loss_total")]


@pytest.mark.parametrize("amp", [False, True])
def test_plumbing_epoch_checkpoint_resume_on_the_hip_backend(tmp_path, capsys, amp):
    """16 synthetic utterances, batch 2, 2 epochs (8 + 8 steps), save, resume for a third epoch; the saved optimizer file
    must load into torch.optim.Adam (the reference's optimizer) and carry step / moments."""
    from transformer_tts_amd import ops, train_fastspeech2 as T
    ops.lib()
    hp, save = _hp(tmp_path, f"\namp = {amp}\nmax_epoch = 2\nlog_every = 1\n")
    args = SimpleNamespace(n_gpus=1)
    torch.manual_seed(0)
    T.run_training(0, args, hp, None)
    out = capsys.readouterr().out
    assert "EPOCH 1 end" in out and "EPOCH 2 end" in out and "step 16 / 8" in out
    ls = _losses(out)
    assert len(ls) == 16 and all(np.isfinite(ls))
    sd = torch.load(os.path.join(save, "network.epoch2"), weights_only=True, map_location="cpu")
    assert sd["encoder.layers.0.ff.f_1.weight"].shape == (128, 32, 9) and list(sd) == list(T.build_model(hp).state_dict())
    osd = torch.load(os.path.join(save, "network.optimizer.epoch2"), weights_only=True, map_location="cpu")
    assert int(osd["state"][0]["step"]) == 16
    # the file is torch.optim.Adam's format: the reference's optimizer loads it
    ref_model = T.build_model(hp)
    ref_model.load_state_dict(sd)
    adam = torch.optim.Adam(ref_model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    adam.load_state_dict(osd)
    first = next(iter(ref_model.parameters()))
    assert torch.equal(adam.state[first]["exp_avg"], osd["state"][0]["exp_avg"]) and float(adam.state[first]["exp_avg"].abs().sum()) > 0
    # resume (reference :428-446): epoch counter, step, lr schedule position and moments come back
    hp.loaded_epoch, hp.loaded_dir, hp.max_epoch = 2, save, 3
    T.run_training(0, args, hp, None)
    out = capsys.readouterr().out
    # (as in the reference, the loop counter restarts at Adam's update count, i.e. one below where it stopped: 16..23)
    assert "epoch 2 loaded" in out and "EPOCH 3 end" in out and "step 23 / 8" in out
    osd3 = torch.load(os.path.join(save, "network.optimizer.epoch3"), weights_only=True, map_location="cpu")
    assert int(osd3["state"][0]["step"]) == 24
    from transformer_tts_amd.utils.utils import get_learning_rate
    assert osd3["param_groups"][0]["lr"] == pytest.approx(get_learning_rate(23, hp.d_model_decoder, hp.warmup_factor, hp.warmup_step))
    assert any("libfs2_hip" in l for l in open("/proc/self/maps")), "the HIP library must be the thing that ran"


def test_frame_budget_epoch_on_the_hip_backend(tmp_path, capsys):
    """hp.max_seqlen batching (LengthsBatchSampler): a different batch shape at almost every step, through the graph
    stepper (first occurrence of a shape eager, second captured) for two epochs."""
    from transformer_tts_amd import train_fastspeech2 as T
    hp, save = _hp(tmp_path, f"\nbatch_size = None\nmax_seqlen = 400\nmax_epoch = 3\nlog_every = 1\nlengths_file = {str(tmp_path / 'lengths.npy')!r}\n")
    T.run_training(0, SimpleNamespace(n_gpus=1), hp, None)
    out = capsys.readouterr().out
    assert "EPOCH 3 end" in out
    sizes = {int(l.split("=")[1]) for l in out.splitlines() if l.startswith("batch size")}
    assert len(sizes) > 1
    assert all(np.isfinite(_losses(out)))


def test_train_loop_graph_replay_gives_the_eager_first_step(tmp_path, capsys):
    """the shipped train_loop with hp.use_graph True vs False from identical state: the first logged losses agree (same
    Philox streams; later steps drift by float-atomic order, see test_hipgraph_replay_equals_eager_training)"""
    from transformer_tts_amd import train_fastspeech2 as T
    from transformer_tts_amd.Models import functional
    firsts = []
    for use_graph in (False, True):
        hp, save = _hp(tmp_path / f"g{int(use_graph)}", f"\nuse_graph = {use_graph}\nlog_every = 1\nmax_epoch = 2\n")
        functional._site_counter[0] = 5000
        torch.manual_seed(0)
        np.random.seed(0)
        T.run_training(0, SimpleNamespace(n_gpus=1), hp, None)
        firsts.append(_losses(capsys.readouterr().out))
    assert len(firsts[0]) == len(firsts[1]) == 16
    np.testing.assert_allclose(firsts[1][0], firsts[0][0], rtol=2e-6)
    np.testing.assert_allclose(firsts[1], firsts[0], rtol=5e-2)        # same training trajectory


def test_eval_forward_after_graphed_steps_sees_the_updated_weights():
    """GraphedTrainStep must invalidate the weight shadows after a replay: an eval() forward between graphed training steps
    must equal the same forward after an explicit invalidate() (which re-derives every shadow from the fp32 masters the replayed
    Adam kernel just wrote) -- bit for bit -- and must differ from the forward taken before the last replay.
    (A twin model stepped eagerly is no yardstick: float-atomics order makes two trajectories drift by ~2e-3 after four Adam
    steps, occasionally 2e-2.)"""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep
    batch = batch_to(CONFIGS["small"]["batch"](), "cuda")
    functional._site_counter[0] = 7000
    model, hp, _ = product_model("small", amp=True, dropout=0.0, device="cuda")
    opt = FusedAdam(model)
    stepper = GraphedTrainStep(model, opt, hp)
    from transformer_tts_amd.train_fastspeech2 import create_masks
    text, mel, pos_text, pos_mel, _, _, _, _, f0, energy, align = batch[:11]
    src_mask, mel_mask = create_masks(pos_text, pos_mel, task="fastspeech2")

    def infer():        # teacher-forced eval() forward outside the graph (the synthesis branch can refuse all-zero durations)
        model.eval()
        with torch.no_grad():
            out = model(text, src_mask, mel_mask, align, f0, energy)
        model.train()
        return out[0].float().cpu()

    for i in range(3):      # eager, capture + replay, replay
        stepper(4000 + i, batch)
    before = infer()
    stepper(4003, batch)    # one more replay: the parameters moved again
    assert len(stepper.graphs) == 1
    after = infer()
    model.rt.invalidate()
    fresh = infer()
    assert torch.equal(after, fresh), "stale weight shadows after graph replay"
    assert float((after - before).abs().max()) > 0, "the replayed step did not change the prediction"


def test_bf16_fast_path_learns_one_batch():
    """The benchmarked configuration of the code path (bf16 operands, dropout on, flash attention because the maps are not kept,
    hipGraph replay, kernel-layout conv gradients in the fused clip + Adam): 150 steps on ONE batch must drive every loss term down.
    Parity tests pin single steps; this pins that the pieces still add up to a training procedure."""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep
    batch = batch_to(CONFIGS["small"]["batch"](), "cuda")
    functional._site_counter[0] = 9000
    model, hp, _ = product_model("small", amp=True, dropout=0.1, device="cuda", return_attn=False)
    opt = FusedAdam(model)
    stepper = GraphedTrainStep(model, opt, hp)
    hist = []
    for i in range(150):
        loss, parts, _ = stepper(4000 + i, batch)          # Noam schedule near its peak (warm-up 4000)
        if i % 10 == 0 or i == 149:
            hist.append([float(loss.detach())] + [float(parts[k].detach()) for k in sorted(parts)])
    assert len(stepper.graphs) == 1
    hist = torch.tensor(hist)
    assert torch.isfinite(hist).all(), hist
    first, last = hist[0], hist[-1]
    names = ["total"] + sorted(parts)
    report = {n: (round(float(a), 4), round(float(b), 4)) for n, a, b in zip(names, first, last)}
    print("first -> last:", report)
    assert (last < first).all(), f"a loss term did not go down: {report}"
    # the synthetic f0 / energy targets are hundreds of units wide (L1 terms of ~1e2 that move slowly at lr ~1e-3); the mel terms
    # and the log-duration term start at O(1) and must fall clearly
    for n in ("frame_before", "frame_after", "duration"):
        i = names.index(n)
        assert last[i] < 0.7 * first[i], f"{n}: {report}"


def test_graph_stepper_falls_back_to_eager_when_capture_fails(monkeypatch, capsys):
    """bench.py runs the multi-GPU case through GraphedTrainStep(eager_fallback=True): if the runtime refuses to capture the step
    (simulated here), the stepper reports it once and keeps training with eager launches instead of raising"""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd import train_fastspeech2 as T
    batch = batch_to(CONFIGS["small"]["batch"](), "cuda")
    functional._site_counter[0] = 9500
    model, hp, _ = product_model("small", amp=True, dropout=0.0, device="cuda", return_attn=False)
    opt = FusedAdam(model)

    class Refuses:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            raise RuntimeError("operation not permitted when stream is capturing (simulated)")

        def __exit__(self, *a):
            return False

    monkeypatch.setattr(torch.cuda, "graph", Refuses)
    strict = T.GraphedTrainStep(model, opt, hp)
    strict(4000, batch)
    with pytest.raises(RuntimeError):
        strict(4001, batch)
    stepper = T.GraphedTrainStep(model, opt, hp, eager_fallback=True)
    p0 = next(model.parameters()).detach().clone()
    losses = [float(stepper(4002 + i, batch)[0].detach()) for i in range(4)]
    assert stepper.broken and not stepper.graphs
    assert "capture failed" in capsys.readouterr().out
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert float((next(model.parameters()).detach() - p0).abs().max()) > 0


def _one_rank_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)       # "nccl" is RCCL on ROCm
    return dist


@pytest.mark.parametrize("graphed", [False, True])
def test_data_parallel_path_on_a_one_rank_rccl_group_equals_the_plain_step(graphed):
    """parallel.DataParallel (bucketed in-place RCCL all-reduce announced from the backward, SyncBatchNorm statistics,
    1/world folded into Adam) on a 1-rank RCCL group, eager and captured in the hipGraph, against the step without it."""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.parallel import DataParallel
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep, train_step
    dist = _one_rank_group()
    batch = batch_to(CONFIGS["small"]["batch"](), "cuda")
    res = []
    for dp in (False, True):
        functional._site_counter[0] = 9000
        model, hp, _ = product_model("small", amp=False, dropout=0.1, device="cuda")
        opt = FusedAdam(model)
        if dp:
            opt.dp = DataParallel(model, opt.arena)
            assert opt.dp.world == 1
        stepper = GraphedTrainStep(model, opt, hp) if graphed else None
        losses = []
        for i in range(3):
            out = stepper(4000 + i, batch) if graphed else train_step(model, opt, 4000 + i, batch, hp)
            losses.append(out[0].item())
        torch.cuda.synchronize()
        if dp:
            assert not opt.dp.works and opt.dp.pending is None
        res.append((losses, opt.arena.p.clone(), opt.arena.g.clone()))
    np.testing.assert_allclose(res[1][0][0], res[0][0][0], rtol=2e-6)
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-3)
    gerr = float((res[0][2] - res[1][2]).norm() / res[0][2].norm())
    assert gerr < 5e-2, gerr


def test_graph_stepper_evicts_the_least_recently_replayed_shape():
    """more batch shapes than max_graphs: every shape still reaches the replay path (capture on its second occurrence evicts the
    shape replayed longest ago instead of leaving every new shape on eager launches for good), the losses of a shape replayed
    after its graph was evicted and re-captured stay finite and the stepper never holds more than max_graphs graphs"""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep
    full = batch_to(CONFIGS["small"]["batch"](), "cuda")
    B = full[0].shape[0]
    assert B >= 4
    shapes = [tuple(x[:k] if torch.is_tensor(x) else x for x in full) for k in (2, 3, 4)]      # three (B, L, T) keys
    functional._site_counter[0] = 9000
    model, hp, _ = product_model("small", amp=True, dropout=0.1, device="cuda")
    opt = FusedAdam(model)
    stepper = GraphedTrainStep(model, opt, hp, max_graphs=2, policy="lru")
    losses, step = [], 4000
    for rnd in range(4):
        for b in shapes:
            loss, _, _ = stepper(step, b)
            losses.append(loss)
            step += 1
            assert len(stepper.graphs) <= 2
    torch.cuda.synchronize()
    assert all(np.isfinite(float(l)) for l in losses)
    st = stepper.stats
    assert st["eager"] == 3 and st["evicted"] >= 1 and st["captured"] >= 3 and st["replayed"] == 9, st


def test_graph_stepper_does_not_recapture_a_cycling_shape_set():
    """policy "frequency" (the default): more shapes than max_graphs coming round in a cycle -- the first max_graphs shapes are captured
    once and replayed, the others are launched eagerly every time (same speed: DESIGN.md section 6) instead of being re-captured at
    every occurrence, which is what always-evict does with such a cycle; a shape that then shows up MORE often than a cached one takes
    its place"""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep
    full = batch_to(CONFIGS["small"]["batch"](), "cuda")
    assert full[0].shape[0] >= 4
    shapes = [tuple(x[:k] if torch.is_tensor(x) else x for x in full) for k in (2, 3, 4)]
    functional._site_counter[0] = 9000
    model, hp, _ = product_model("small", amp=True, dropout=0.1, device="cuda")
    opt = FusedAdam(model)
    stepper = GraphedTrainStep(model, opt, hp, max_graphs=2)
    assert stepper.policy == "frequency"
    losses, step = [], 4000
    for rnd in range(4):
        for b in shapes:
            loss, _, _ = stepper(step, b)
            losses.append(float(loss))
            step += 1
    st = dict(stepper.stats)
    # round 1: three eager; round 2: shapes 0 and 1 captured (+ replayed), shape 2 eager; rounds 3-4: two replays + one eager each
    assert st["captured"] == 2 and st["evicted"] == 0 and st["eager"] == 3 + 3 and st["replayed"] == 6, st
    for _ in range(8):              # shape 2 alone: once it has been seen clearly more often (+2) than the least recently replayed cached shape it is captured
        loss, _, _ = stepper(step, shapes[2])
        losses.append(float(loss))
        step += 1
    torch.cuda.synchronize()
    st = stepper.stats
    assert st["captured"] == 3 and st["evicted"] == 1 and len(stepper.graphs) == 2, st
    assert all(np.isfinite(l) for l in losses)


def test_eager_steps_do_not_accumulate_device_memory():
    """eager train steps on one batch: the bytes the caching allocator has handed out are the same after every step (a list that kept
    the one-hot rows of the pitch / energy embedding gradients alive grew by 45 MB per configs[1] step until round 4)"""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import train_step
    functional._site_counter[0] = 9500
    model, hp, _ = product_model("small", amp=True, dropout=0.1, device="cuda")
    opt = FusedAdam(model)
    batch = batch_to(CONFIGS["small"]["batch"](), "cuda")
    seen = []
    for i in range(10):
        loss, _, _ = train_step(model, opt, 4000 + i, batch, hp)
        del loss
        torch.cuda.synchronize()
        if i >= 3:
            seen.append(torch.cuda.memory_allocated())
    assert len(set(seen)) == 1, seen
    assert len(model.rt._keep) == 0


def test_captured_graphs_do_not_pin_the_activations_of_their_step():
    """a graph entry keeps the values a replay rewrites (loss, parts), not the autograd graph of the captured step: the device memory
    a further captured shape costs is a fraction of one step's activation footprint (it used to be the whole footprint -- 2 GiB per
    configs[1] graph, out of memory at ~140 graphs: profiles/r04_h_graph_memory.txt)"""
    from transformer_tts_amd.Models import functional
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep, train_step
    full = batch_to(CONFIGS["small"]["batch"](), "cuda")
    assert full[0].shape[0] >= 4
    shapes = [tuple(x[:k] if torch.is_tensor(x) else x for x in full) for k in (4, 3, 2)]
    functional._site_counter[0] = 9700
    model, hp, _ = product_model("small", amp=True, dropout=0.1, device="cuda")
    opt = FusedAdam(model)
    train_step(model, opt, 4000, shapes[0], hp)                 # caches, workspaces
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    train_step(model, opt, 4001, shapes[0], hp)
    torch.cuda.synchronize()
    footprint = torch.cuda.max_memory_allocated() - base        # what one step of the LARGEST of the shapes needs at its peak
    assert footprint > 0
    stepper = GraphedTrainStep(model, opt, hp)
    after = []
    for i, b in enumerate(shapes):
        stepper(4002 + 2 * i, b)                                # first sight: eager
        loss, parts, _ = stepper(4003 + 2 * i, b)               # second sight: captured + replayed
        torch.cuda.synchronize()
        assert loss.grad_fn is None and all(v.grad_fn is None for v in parts.values())
        after.append(torch.cuda.memory_allocated())
    assert stepper.stats["captured"] == 3
    per_graph = (after[-1] - after[0]) / 2
    assert per_graph < 0.25 * footprint, (per_graph, footprint, after)

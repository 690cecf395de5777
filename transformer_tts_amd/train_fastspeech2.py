# -*- coding: utf-8 -*-
"""FastSpeech2 trainer -- the reference's ``train_fastspeech2.py`` surface on the MI355X-native path.

Same CLI (``--hp_file``), same module-level functions and arguments as the reference
(``create_masks`` :55-82, ``train_loop`` :100-323, ``train_epoch`` :330-350, ``init_distributed`` :352-362,
``cleanup`` :364-365, ``run_distributed`` :368-374, ``run_training`` :376-454), same checkpoint file names.
Differences, all on purpose:
* the model is ``transformer_tts_amd.Models.fastspeech2.FastSpeech2`` (HIP kernels, no CPU path);
* clip_grad_norm_ + Adam are one fused kernel over flat arenas (``optim.FusedAdam``); ``hp.amp`` selects
  bf16 MFMA compute (fp32 master weights, no GradScaler needed) instead of fp16 autocast;
* one process per GPU is started by ``torch.distributed.run`` / ``mp.spawn`` as in the reference, but the
  gradient exchange is ``parallel.DataParallel`` (bucketed in-place RCCL all-reduce overlapped with
  backward + SyncBatchNorm statistics), not DistributedDataParallel;
* losses are printed every ``hp.log_every`` steps (reference: every step, ~10 host syncs per step).
"""
import argparse
import collections
import filecmp
import gc
import os
import random
import shutil
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader

from .Models.fastspeech2 import FastSpeech2
from .Models.functional import l1_loss, l1_loss_multi
from .optim import FusedAdam
from .utils import hparams as hp
from .utils.utils import fill_variables, get_learning_rate, init_weight, load_model, log_config
from .datasets import datasets_fastspeech2 as datasets

random.seed(77)
DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def npeak_mask(size):
    """reference :42-53: (1, size, size) boolean lower-triangular mask (diagonal included) on DEVICE."""
    return torch.tril(torch.ones((1, size, size), dtype=torch.bool, device=DEVICE))


def _pad_mask(pos, pad):
    """(pos != pad).unsqueeze(-2).  On the GPU: one launch that also leaves the row bounds / ranking the attention kernels of the stack
    read (ops.pad_mask_info) with the mask (`_fs2_kinfo`: EncoderStackFunction picks it up instead of scanning the mask again)."""
    from . import ops
    if pos.is_cuda and pos.dim() == 2 and pos.dtype == torch.int64 and pos.shape[0] <= ops.PAD_MASK_MAX_B and 0 < pos.shape[1] <= ops.PAD_MASK_MAX_T:
        mask, info = ops.pad_mask_info(pos, pad)
        mask = mask.unsqueeze(-2)
        mask._fs2_kinfo = info
        return mask
    return (pos != pad).unsqueeze(-2)


def create_masks(src_pos, trg_pos, task="transformer", src_pad=0, trg_pad=0, debug=False):
    """reference :55-82.  (B,t) positions -> (B,1,t) bool key masks for the FastSpeech2 / LightSpeech tasks; for the general
    transformer task the target mask is additionally combined with the no-peak mask -> (B,t,t).  (debug: the +-3 context
    window of the reference is computed and then unused there; it is not reproduced.)"""
    assert not debug, "the reference's debug branch builds a context-window mask it never uses"
    if task.lower() in ("fastspeech2", "lightspeech"):
        return _pad_mask(src_pos, src_pad), _pad_mask(trg_pos, trg_pad)
    src_mask = (src_pos != src_pad).unsqueeze(-2)
    if trg_pos is None:
        return src_mask, None
    trg_mask = (trg_pos != trg_pad).unsqueeze(-2)
    return src_mask, trg_mask & npeak_mask(trg_pos.size(1)).to(trg_pos.device)


def mse_loss_arelbo(input, target):
    """reference :85-88 ("Preventing Posterior Collapse Induced by Oversmoothing in Gaussian VAE"): plain torch arithmetic on
    the caller's tensors -- a host-level helper of the module surface, not on the FastSpeech2 hot path (no call site there)."""
    return 0.5 * (target.numel() // target.size(0)) * torch.log(torch.mean((input - target) ** 2))


def loss_mel(hp, pred, y, channel_wise=False, loss="l1", channel_weight=None):
    """reference :90-98: nn.L1Loss over every element, or two weighted L1 terms over mel channels [0,20) and [20,M) when
    channel_wise (hp.channel_weight).  Computed by the fused L1 kernel (with its hand-written backward)."""
    if channel_wise:
        print("channel")
        return hp.channel_weight[0] * l1_loss(pred[:, :, :20].contiguous(), y[:, :, :20].contiguous()) + \
            hp.channel_weight[1] * l1_loss(pred[:, :, 20:].contiguous(), y[:, :, 20:].contiguous())
    return l1_loss(pred, y)


def compute_losses(hp, outputs, mel, alignment, f0, energy):
    """The five nn.L1Loss() terms of the reference (:212-259); returns (total, dict of parts)."""
    outputs_prenet, outputs_postnet, log_d_prediction, p_prediction, e_prediction = outputs[:5]
    if not getattr(hp, "channel_wise", False):      # every term in one launch each way (sum in the reference's order of terms)
        names, items = ["frame_before"], [(outputs_prenet, mel, False)]
        if hp.postnet_pred:
            names.append("frame_after"); items.append((outputs_postnet, mel, False))
        if hp.pitch_pred:
            names.append("f0"); items.append((p_prediction, f0, False))
        if hp.energy_pred:
            names.append("energy"); items.append((e_prediction, energy, False))
        names.append("duration"); items.append((log_d_prediction, alignment, True))     # target log(alignment + 1)
        terms, total = l1_loss_multi(items)
        return total, dict(zip(names, terms))
    parts = {"frame_before": l1_loss(outputs_prenet, mel)}
    loss = parts["frame_before"]
    if hp.postnet_pred:
        parts["frame_after"] = l1_loss(outputs_postnet, mel)
        loss = loss + parts["frame_after"]
    parts["duration"] = l1_loss(log_d_prediction, alignment, True)       # target log(alignment + 1)
    if hp.pitch_pred:
        parts["f0"] = l1_loss(p_prediction, f0)
        loss = loss + parts["f0"]
    if hp.energy_pred:
        parts["energy"] = l1_loss(e_prediction, energy)
        loss = loss + parts["energy"]
    loss = loss + parts["duration"]
    return loss, parts


def _set_lr(optimizer, step, hp):
    if hp.optimizer.lower() != "radam":
        lr = get_learning_rate(step, hp.d_model_decoder, hp.warmup_factor, hp.warmup_step)
        for param_group in optimizer.param_groups:
            param_group["lr"] = lr


def step_body(model, optimizer, hp, text, mel, pos_text, pos_mel, f0, energy, alignment):
    """Device work of one iteration (reference :153-315): masks, zero_grad, forward, losses, backward,
    clip + Adam kernels, dropout-stream advance.  No host synchronisation: capturable in a hipGraph."""
    src_mask, trg_mask = create_masks(pos_text, pos_mel, task=hp.model)
    optimizer.zero_grad()
    outputs = model(text, src_mask, trg_mask, alignment, f0, energy, None, spkr_emb=None, fix_mask=hp.fix_mask,
                    temperature=None, hop_size=None)
    loss, parts = compute_losses(hp, outputs, mel, alignment, f0, energy)
    loss.backward(model.rt.seed_grad(loss))
    if isinstance(optimizer, FusedAdam):
        optimizer.launch()                                  # global-norm clip (1.0) fused into the Adam kernel
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        model.rt.invalidate()
    model.rt.get_rng(text.device).advance()
    return loss, parts


def train_step(model, optimizer, step, d, hp):
    """One iteration of the reference loop body (:117-315); returns (loss tensor, parts, batch size)."""
    _set_lr(optimizer, step, hp)
    text, mel, pos_text, pos_mel, text_lengths, mel_lengths, stop_token, spk_emb, f0, energy, alignment = d[:11]
    mv = lambda x: x.to(DEVICE, non_blocking=True)
    text, mel, pos_text, pos_mel, f0, energy, alignment = (mv(x) for x in (text, mel, pos_text, pos_mel, f0, energy, alignment))
    if isinstance(optimizer, FusedAdam):
        optimizer.host_update()
    loss, parts = step_body(model, optimizer, hp, text, mel, pos_text, pos_mel, f0, energy, alignment)
    return loss, parts, mel.shape[0]


_GC_SETTLED = [False]


def settle_gc(force=False):
    """Once per process, when the long-lived objects of a run exist (model, parameter arenas, weight-shadow tables, cached descriptor
    lists, the first captured graphs): collect what the set-up left behind and FREEZE the survivors (gc.freeze: they move to a permanent
    generation the collector never walks again).  Without it every full collection of the cyclic garbage a training step produces (the
    autograd contexts of the hand-written Functions: ~10 young collections per 64 steps) also traverses all of those: 58 ms measured
    (`profiles/r04_f_gc_settle.txt`, round 4), during which the launch thread enqueues nothing -- one such pause inside a 100-step window reads as
    +0.3-0.6 ms per step, and the eager launch path (6 ms of host time per 7.2 ms step) falls behind the GPU for the steps after it."""
    if _GC_SETTLED[0] and not force:
        return
    _GC_SETTLED[0] = True
    gc.collect()
    gc.freeze()


def unsettle_gc():
    """undo settle_gc at the end of a run"""
    if _GC_SETTLED[0]:
        _GC_SETTLED[0] = False
        gc.unfreeze()


def _dist_alive():
    return dist.is_available() and dist.is_initialized()


STEP_INPUTS = (0, 1, 2, 3, 8, 9, 10)      # entries of the collate 16-tuple a FastSpeech2 step reads: text, mel, pos_text, pos_mel, f0, energy, alignment


class GraphedTrainStep:
    """train_step with the device work of each batch SHAPE captured once in a hipGraph and replayed:
    the ~600 kernel launches of a step stop costing host time (the step is launch-bound in Python otherwise).
    First occurrence of a shape runs eagerly (allocator warm-up), the second is captured, later ones replay.
    Padding is semantically live in this model (BatchNorm statistics and the L1 losses include padded
    positions), so batches are never padded to a common shape -- one graph per (B, L_pad, T_pad).
    At most `max_graphs` graphs are kept (256: a corpus batched by a frame budget, reference datasets_fastspeech2.py:749-813, produces
    hundreds of shapes; the graphs share one memory pool -- a capture reuses what the captures before it freed -- and a capture costs
    7-16 ms once the garbage collector is settled, so a shape is worth capturing when it comes round a second time).  When the
    cache is full, policy "frequency" (default) captures a shape only if it has been seen clearly more often (by max(2, 25 %)) than the
    least recently replayed cached shape (which it then evicts); otherwise the step is launched eagerly -- the eager launch path runs at the replay speed
    ("Dynamic shapes"), while a capture costs tens of milliseconds, so a shape set larger than the cache must not
    be re-captured in a cycle.  Policy "lru" (the round-4a behaviour, FS2_GRAPH_POLICY=lru): always capture, evict the least recently
    replayed."""

    def __init__(self, model, optimizer, hp, max_graphs=256, eager_fallback=False, body=None, inputs=None, eager=None, set_lr=None,
                 policy=None):
        """body / inputs / eager / set_lr: the device part of a step, the batch entries it reads, the eager step and the learning-rate rule
        of ANOTHER trainer of this package (transformer_tts_amd.train: the autoregressive model); default: this module's"""
        assert isinstance(optimizer, FusedAdam)
        self.model, self.optimizer, self.hp = model, optimizer, hp
        self.body, self.inputs = body or step_body, inputs or STEP_INPUTS
        self.eager, self.set_lr = eager or train_step, set_lr or _set_lr
        self.max_graphs = max_graphs
        self.seen, self.graphs = collections.Counter(), collections.OrderedDict()
        self.tick, self.last_used = 0, {}
        self.policy = policy or os.environ.get("FS2_GRAPH_POLICY", "frequency")
        assert self.policy in ("frequency", "lru")
        self.stats = {"eager": 0, "captured": 0, "replayed": 0, "evicted": 0}
        self.pool = None
        # eager_fallback: if a capture raises (e.g. a collective that refuses stream capture on some multi-GPU setup), say so once
        # and run every later step eagerly instead of dying -- every rank sees the same failure, so the ranks stay in step
        self.eager_fallback, self.broken = eager_fallback, False

    def __call__(self, step, d):
        self.set_lr(self.optimizer, step, self.hp)
        tensors = [d[i] for i in self.inputs]       # text, mel, pos_text, pos_mel, f0, energy, alignment
        key = (tuple(tensors[0].shape), tuple(tensors[1].shape))
        entry = self.graphs.get(key)
        if self.broken:
            return self.eager(self.model, self.optimizer, step, d, self.hp)
        self.seen[key] += 1
        self.tick += 1
        if entry is not None:
            self.last_used[key] = self.tick
        if entry is None and self.seen[key] == 1:
            self.stats["eager"] += 1
            return self.eager(self.model, self.optimizer, step, d, self.hp)
        if entry is None and self.policy == "frequency" and len(self.graphs) >= max(1, self.max_graphs):
            victim = next(iter(self.graphs))            # (the least recently replayed cached shape)
            stale = self.tick - self.last_used.get(victim, 0) > 64 * max(1, self.max_graphs)      # (a shape the corpus no longer produces)
            # (clearly more often: shapes that come round once per epoch differ by one sight depending on where the epoch's shuffle put
            #  them -- without the margin such a set keeps re-capturing, at ~100 ms per capture against 7.5 ms per eager step)
            margin = max(2, self.seen[victim] // 4)
            if self.seen[key] <= self.seen[victim] + margin and not stale:     # eager launch, the cache stays as it is
                self.stats["eager"] += 1
                return self.eager(self.model, self.optimizer, step, d, self.hp)
        if entry is None:
            while len(self.graphs) >= max(1, self.max_graphs):      # the shape replayed longest ago makes room
                okey, old = self.graphs.popitem(last=False)
                self.last_used.pop(okey, None)
                del old
                self.stats["evicted"] += 1
            static = [t.to(DEVICE).clone() for t in tensors]
            g = torch.cuda.CUDAGraph()
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            torch.cuda.synchronize()
            # the captured launches are NOT executed: capture with the optimizer's host state untouched, then run the
            # step through the replay below like every later occurrence of this shape
            # Capture mode: with a process group alive, torch.distributed's watchdog thread polls the completion events of
            # earlier eager collectives (hipEventQuery) whenever it wakes up; under the default "global" mode such a call
            # from ANOTHER thread invalidates the capture in progress.  "thread_local" restricts the unsafe-call check to
            # the capturing thread; work submitted to the capturing stream by the autograd thread is captured either way.
            mode = os.environ.get("FS2_CAPTURE_ERROR_MODE") or ("thread_local" if _dist_alive() else "global")
            # No garbage collection while the stream is capturing: a collection that happens to run inside the capture destroys whatever
            # cyclic garbage earlier code left behind -- CUDA graphs, events, tensors of another stepper -- and a HIP call refused in
            # capture mode inside such a destructor ends the process (seen once in the GPU suite: "Fatal Python error: Aborted" with the
            # collector on the stack of this capture).  Collect first, then keep the collector off until the capture has ended.
            gc_on = gc.isenabled()
            gc.collect()
            gc.disable()
            try:
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode=mode):
                    loss, parts = self.body(self.model, self.optimizer, self.hp, *static)
            except Exception as e:      # noqa: BLE001  (whatever the runtime raises for an operation it cannot capture)
                if gc_on:
                    gc.enable()
                if not self.eager_fallback:
                    raise
                print(f"GraphedTrainStep: capture failed ({type(e).__name__}: {e}); continuing with eager launches", flush=True)
                self.broken = True
                torch.cuda.synchronize()
                return self.eager(self.model, self.optimizer, step, d, self.hp)
            if gc_on:
                gc.enable()
            # Keep the VALUES the replays rewrite, not the autograd graph behind them: the hand-written Functions hold their activations
            # as plain attributes of their contexts, so a `loss` that still carries its grad_fn pins every activation of the captured
            # step in the graph pool -- 2 GiB per configs[1] graph, 128 GiB for 64 graphs (measured, round 4).  Detached, the capture's
            # intermediates go back to the shared pool and the next capture reuses them (graphs replay one after the other).
            loss = loss.detach()
            parts = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}
            entry = self.graphs[key] = (g, static, loss, parts)
            self.last_used[key] = self.tick
            self.stats["captured"] += 1
        self.graphs.move_to_end(key)
        self.stats["replayed"] += 1
        self.optimizer.host_update()
        g, static, loss, parts = entry
        if all(src.is_cuda and src.is_contiguous() and src.dtype == dst.dtype and src.device == dst.device for dst, src in zip(static, tensors)):
            from . import ops
            ops.copy_batched(static, tensors)           # the step's inputs into the graph's static buffers: one launch, not one per tensor
        else:
            for dst, src in zip(static, tensors):
                dst.copy_(src, non_blocking=True)
        g.replay()
        # the replayed Adam step rewrote the parameters through raw pointers: a forward outside the graph (evaluation or
        # synthesis between training steps) must not reuse weight shadows derived before it
        self.model.rt.invalidate()
        return loss, parts, static[1].shape[0]


class DevicePrefetcher:
    """Iterates a loader ONE batch ahead: while step i runs, the tensors of batch i+1 (pinned by the DataLoader) are
    copied host -> device on a separate HIP stream, so the 15.8 MB of a config-2 batch never sit on the compute stream.
    The consumer's stream waits on the copy's event when it takes the batch."""

    def __init__(self, loader, device, indices=None):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        # indices: the batch entries the step reads (None = every tensor).  A host -> device copy call costs ~50 us of host time
        # and the training thread is busy with the graph launch for the rest of the step, so the calls sit BETWEEN two steps:
        # the four entries of the 11-tuple that the FastSpeech2 step never touches are left on the host.
        self.indices = None if indices is None else frozenset(indices)

    def _stage(self, d):
        if self.stream is None:
            return d, None
        with torch.cuda.stream(self.stream):
            out = tuple(x.to(self.device, non_blocking=True) if torch.is_tensor(x) and (self.indices is None or i in self.indices) else x
                        for i, x in enumerate(d))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        staged = None
        for d in self.loader:
            nxt = self._stage(d)
            if staged is not None:
                yield self._take(staged)
            staged = nxt
        if staged is not None:
            yield self._take(staged)

    def _take(self, staged):
        d, ev = staged
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for x in d:
                if torch.is_tensor(x) and x.is_cuda:
                    x.record_stream(cur)
        return d


def train_loop(model, optimizer, step, epoch, args, hp, rank, dataloader):
    log_every = max(1, int(getattr(hp, "log_every", 1)))
    # the fast path of the shipped trainer = the benchmarked path: hipGraph replay per batch shape (FusedAdam on the GPU;
    # hp.use_graph = False keeps eager launches) fed by a one-batch-ahead copy stream
    on_gpu = isinstance(optimizer, FusedAdam) and optimizer.arena.p.is_cuda
    run = train_step
    # (scheduled sampling draws host random numbers inside the forward, reference Models/varianceadaptor.py:261-282: eager launches)
    if on_gpu and bool(getattr(hp, "use_graph", True)) and float(getattr(hp, "p_scheduled_sampling", 0.0)) == 0.0:
        stepper = getattr(optimizer, "_fs2_graphed", None)
        if stepper is None or stepper.model is not model:
            stepper = optimizer._fs2_graphed = GraphedTrainStep(model, optimizer, hp, eager_fallback=_dist_alive())   # (N > 1 capture has never run on hardware: a refusal falls back to eager launches, as in bench.py)
        run = lambda m, o, st, d, h: stepper(st, d)
    batches = DevicePrefetcher(dataloader, optimizer.arena.p.device, indices=STEP_INPUTS) if on_gpu else dataloader
    n_run = 0
    for d in batches:
        loss, parts, batch_size = run(model, optimizer, step, d, hp)
        n_run += 1
        if n_run == 3:              # (the model, the optimizer and the first graph exist: keep the collector off them from here on)
            settle_gc()
        if step % log_every == 0:
            print(f"loss_frame_before = {parts['frame_before'].item()}")
            print(f"loss_duration = {parts['duration'].item()}")
            if "f0" in parts:
                print(f"loss_f0 = {parts['f0'].item()}")
            if "energy" in parts:
                print(f"loss_energy = {parts['energy'].item()}")
            if "frame_after" in parts:
                print(f"loss_frame_after = {parts['frame_after'].item()}")
            print(f"loss_total = {loss.item()}")
            print(f"batch size = {batch_size}")
            print(f"step {step} / {len(dataloader)}")
            assert not torch.isnan(loss), "loss is nan"
            sys.stdout.flush()
        step += 1
    if rank == 0 and (epoch + 1) >= (hp.max_epoch - 10):
        torch.save(model.state_dict(), hp.save_dir + "/network.epoch{}".format(epoch + 1))
    elif rank == 0 and ((epoch + 1) % hp.save_per_epoch >= (hp.save_per_epoch - 10) or ((epoch + 1) % hp.save_per_epoch == 0)):
        torch.save(model.state_dict(), hp.save_dir + "/network.epoch{}".format(epoch + 1))
    if rank == 0 and (epoch + 1) % hp.save_per_epoch == 0:
        torch.save(optimizer.state_dict(), hp.save_dir + "/network.optimizer.epoch{}".format(epoch + 1))
    return step


def train_epoch(model, optimizer, step, start_epoch, args, hp, rank):
    alignment_pred = hp.model.lower() in ("fastspeech2", "lightspeech")
    dataset_train = datasets.TrainDatasets(hp.train_script, hp, alignment_pred=alignment_pred, pitch_pred=hp.pitch_pred,
                                           energy_pred=hp.energy_pred, accent_emb=hp.accent_emb)
    if hp.batch_size is not None:                                    # reference :338-341
        sampler = datasets.NumBatchSampler(dataset_train, hp.batch_size)
    elif hp.max_seqlen is not None:
        # frame-budget batching: the batch shape changes from step to step.  Nothing in the kernels depends on the
        # shape; GraphedTrainStep (if used) captures one hipGraph per shape it sees twice and runs the rest eagerly.
        sampler = datasets.LengthsBatchSampler(dataset_train, hp.max_seqlen, hp, hp.lengths_file, shuffle=True,
                                               shuffle_one_time=False)
    else:
        raise ValueError("set hp.batch_size or hp.max_seqlen")
    train_sampler = datasets.DistributedSamplerWrapper(sampler) if args.n_gpus > 1 else sampler
    dataloader = DataLoader(dataset_train, batch_sampler=train_sampler, num_workers=int(getattr(hp, "num_workers", 8)),
                            collate_fn=datasets.collate_fn, pin_memory=True)
    for epoch in range(start_epoch, hp.max_epoch):
        start_time = time.time()
        step = train_loop(model, optimizer, step, epoch, args, hp, rank, dataloader)
        print("EPOCH {} end".format(epoch + 1))
        print(f"elapsed time {time.time() - start_time}")


def init_distributed(rank, n_gpus, port):
    assert torch.cuda.is_available(), "Distributed mode requires CUDA."
    torch.cuda.set_device(rank % n_gpus)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", world_size=n_gpus, rank=rank)   # "nccl" is RCCL on ROCm


def cleanup():
    if dist.is_initialized():
        dist.destroy_process_group()


def run_distributed(fn, args, hp):
    port = "60" + str(int(time.time()))[-4:]
    print(f"port = {port}")
    try:
        mp.spawn(fn, args=(args, hp, port), nprocs=args.n_gpus, join=True)
    finally:
        cleanup()


def clip_norm(hp):
    """max gradient norm of the reference trainer (:304-315): ``hp.clip`` under amp (None = no clipping), else 1.0"""
    return getattr(hp, "clip", 1.0) if getattr(hp, "amp", False) else 1.0


def build_model(hp):
    """Argument wiring of the reference (:381-389), dropout_postnet hard-coded to 0.5 there."""
    return FastSpeech2(hp=hp, src_vocab=hp.vocab_size, trg_vocab=hp.mel_dim, d_model_encoder=hp.d_model_encoder,
                       N_e=hp.n_layer_encoder, n_head_encoder=hp.n_head_encoder,
                       ff_conv_kernel_size_encoder=hp.ff_conv_kernel_size_encoder,
                       concat_after_encoder=hp.concat_after_encoder, d_model_decoder=hp.d_model_decoder,
                       N_d=hp.n_layer_decoder, n_head_decoder=hp.n_head_decoder,
                       ff_conv_kernel_size_decoder=hp.ff_conv_kernel_size_decoder,
                       concat_after_decoder=hp.concat_after_decoder,
                       dropout_variance_adaptor=hp.dropout_variance_adaptor, reduction_rate=hp.reduction_rate,
                       dropout=hp.dropout, dropout_postnet=0.5, n_bins=hp.nbins, f0_min=hp.f0_min, f0_max=hp.f0_max,
                       energy_min=hp.energy_min, energy_max=hp.energy_max, pitch_pred=hp.pitch_pred,
                       energy_pred=hp.energy_pred, accent_emb=hp.accent_emb, output_type=hp.output_type,
                       num_group=hp.num_group, multi_speaker=hp.is_multi_speaker, spk_emb_dim=hp.spk_emb_dim,
                       spk_emb_architecture=hp.spk_emb_architecture)


def run_training(rank, args, hp, port=None):
    if args.n_gpus > 1:
        init_distributed(rank, args.n_gpus, port)
    if hp.model.lower() != "fastspeech2":
        raise AttributeError(hp.model)
    model = build_model(hp)
    model.apply(init_weight)
    model.train()
    print(model)
    device = torch.device("cuda", rank) if torch.cuda.is_available() else torch.device("cpu")
    model = model.to(device)
    assert hp.optimizer.lower() != "radam", "the reference's radam branch is unreachable (SURVEY section 2.1 #20)"
    # reference :304-315: the amp branch clips with hp.clip (no clipping when it is None), the fp32 branch with 1.0
    optimizer = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=clip_norm(hp))
    if args.n_gpus > 1:
        from .parallel import DataParallel
        optimizer.dp = DataParallel(model, optimizer.arena)
        dist.barrier()
    if hp.loaded_epoch is not None:
        start_epoch = hp.loaded_epoch
        path = os.path.join(hp.loaded_dir, "network.epoch{}".format(hp.loaded_epoch))
        print("epoch {} loaded".format(hp.loaded_epoch))
        state = load_model(path, map_location=device)
        state = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
        model.load_state_dict(state)
        model.rt.invalidate()
        opt_state = torch.load(os.path.join(hp.loaded_dir, "network.optimizer.epoch{}".format(hp.loaded_epoch)),
                               map_location=device, weights_only=True)
        optimizer.load_state_dict(opt_state)
        step = int(opt_state["state"][0]["step"])
    else:
        start_epoch, step = 0, 1
    print("params = {0:.2f}M".format(sum(p.numel() for p in model.parameters()) / 1000 / 1000))
    try:
        train_epoch(model, optimizer, step, start_epoch, args, hp, rank)
    finally:
        unsettle_gc()       # (a process that goes on after the run -- tests, notebooks -- gets the ordinary collector back)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--hp_file", type=str, default="hparams.py")
    args = parser.parse_args(argv)
    hp.configure(args.hp_file)
    fill_variables(hp)
    log_config(hp)
    assert hp.architecture == "text-mel", f"invalid architecture {hp.architecture}"
    os.makedirs(hp.save_dir, exist_ok=True)
    dst = f"{hp.save_dir}/hparams.py"
    if args.hp_file != dst and not (os.path.exists(dst) and filecmp.cmp(args.hp_file, dst)):
        shutil.copyfile(args.hp_file, dst)
    args.n_gpus = torch.cuda.device_count()
    if args.n_gpus > 1:
        run_distributed(run_training, args, hp)
    else:
        run_training(0, args, hp, None)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of a FastSpeech2 train step (d_model=256, 4+4 FFT layers, 80 mel,
batch 48 per GPU, bf16) on 1/2/4/8 MI355X  --  BASELINE.json `metric`, configs[1] (configs[2] for N > 1).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  When the process was not started by torch.distributed.run (no WORLD_SIZE in
the environment) bench.py starts N fresh rank processes itself -- `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 bench.py ...` as a CHILD, before anything touches the GPU here (the
reference spawns its own ranks too: train_fastspeech2.py:368-374) -- waits for them and exits with their code; rank 0
prints the JSON line.

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
forward + 5 L1 losses + backward + global-norm clip + Adam (+ gradient all-reduce and SyncBatchNorm
statistics over RCCL when N > 1), with the reference's dropout rates (0.1 / 0.5 / 0.5, attention dropout
always on).  `value` = valid (un-padded) mel frames of all ranks / wall time of K steps bracketed by
barrier + synchronize, MAX over ranks.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (the MFMA GEMM variant with the largest summed time): algorithmic FLOPs
                  and algorithmic bytes (every operand element once) of its launches / their summed duration,
                  measured with a HIP event pair on the launch stream around every launch of instrumented eager
                  steps that follow the timed region (graph replays cannot be bracketed per kernel).  `bound` is
                  the roof that binds those launches on average (FLOP/byte against 2.5 PFLOP/s : 8 TB/s); both
                  fractions are given.  `traffic` = HBM bytes per launch from the PMC pass in profiles/.
  cpu_baseline -- the oracle (CPU restatement of the reference step, kind "port") timed on this host's cores
                  on one config-2 batch (~10 s); a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak (6.3 TB/s achievable), same table
POOL = 8                        # pre-built batches cycled


def bench_hp(amp=True, workload="cfg2", fp8=False, return_attn=False):
    from types import SimpleNamespace
    from golden_configs import _BASE, CONFIGS
    from transformer_tts_amd.utils.utils import fill_variables
    d = dict(_BASE)
    d.update(CONFIGS["bench"]["hp"])
    if workload == "cfg3":      # BASELINE.json configs[3]: autoregressive Transformer-TTS (encoder-decoder cross-attention + post-net)
        from golden_configs import _AR
        d.update(_AR)
        d.update(dropout_prenet=0.5, dropout_postnet=0.5)      # (reference defaults; the attention dropout follows hp.dropout)
    if workload == "cfg4":      # BASELINE.json configs[4]: d_model 512, 6+6 FFT layers, batch 64 per GPU, fp8 MFMA GEMMs
        d.update(batch_size=64, d_model_encoder=512, n_layer_encoder=6, n_head_encoder=4, d_model_decoder=512, n_layer_decoder=6,
                 n_head_decoder=4)
    hp = SimpleNamespace(**d)
    hp.amp, hp.dropout, hp.dropout_variance_adaptor, hp.fp8 = amp, 0.1, 0.5, fp8
    hp.return_attn = return_attn        # the training loop never reads the attention maps (reference train_fastspeech2.py:173-174,283-287: only commented-out plotting code touches them)
    fill_variables(hp, verbose=False)
    return hp


def gemm_work(g):
    """(FLOPs, algorithmic bytes, shape tuple) of one fs2_gemm descriptor.  Algorithmic bytes = every DISTINCT operand element once +
    every output element once: a batch index that does not move an operand (batch stride 0: the X of a fused q/k/v weight gradient,
    wgrad_batched) or that only shifts its rows (conv = 2: the taps of a Conv1d weight gradient read the same dY and the same X)
    does not multiply that operand's bytes; conv = 1 reads its A rows once, not once per tap."""
    taps = g.taps if g.conv == 1 else 1
    b1, b2 = max(1, g.batch1), max(1, g.batch2)
    nb = b1 * b2
    flops = 2.0 * g.M * g.N * g.K * taps * nb
    es, cs = {0: 4, 1: 2, 2: 1, 3: 1}[g.dtype], (2 if g.c_dtype == 1 else 4)
    n_a = (b1 if (b1 > 1 and g.sA1 != 0) else 1) * (b2 if (b2 > 1 and g.sA2 != 0 and g.conv != 2) else 1)
    n_b = (b1 if (b1 > 1 and g.sB1 != 0) else 1) * (b2 if (b2 > 1 and g.sB2 != 0 and g.conv != 2) else 1)
    abytes = float(es * (g.M * g.K * n_a + g.N * g.K * taps * n_b) + cs * g.M * g.N * nb)
    flags = ("b" if g.bias else "") + ("r" if g.relu else "") + ("m" if g.relu_mask else "") + ("+" if g.residual else "") + \
            (f"s{g.colstats_mode}" if g.colstats else "") + ("a" if g.accumulate else "") + ("f" if g.c_dtype == 0 and g.dtype == 1 else "")
    shape = (g.M, g.N, g.K, taps if g.conv == 1 else (g.batch2 if g.conv == 2 else 1), g.batch1 * g.batch2, g.split_k, flags or "-")
    return flops, abytes, shape


def family(tile):
    """kernel family of bench.py's tile id (fs2_gemm_last_tile): the instances of ONE kernel template are one family"""
    return "km" if tile == 129 else "ws" if tile == 131 else "ring" if tile >= 130 else "4wave"


class GemmTimer:
    """Wraps ops._gemm_call: a HIP event pair on the launch stream around every fs2_gemm launch."""

    def __init__(self):
        self.records = []      # (key, flops, start, end)
        self._pool = []        # events created ahead of the instrumented steps (creation is host time)
        self.overhead_ms = 0.0  # elapsed time of an EMPTY event pair on a busy stream (subtracted from every launch)

    def calibrate(self):
        """what a back-to-back event pair measures with nothing between the two records: the record/launch gap that
        every bracketed kernel time contains (rocprofv3's kernel durations do not)"""
        torch.cuda._sleep(2_000_000)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for s, e in pairs:
            s.record(); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in pairs)
        self.overhead_ms = ts[len(ts) // 2]

    def reserve(self, n):
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(n)]

    def install(self):
        from transformer_tts_amd import ops
        self._orig = ops._gemm_call
        self._last_tile = ops.lib().fs2_gemm_last_tile

        work = gemm_work

        def record(g, s, e):
            flops, abytes, shape = work(g)
            # block tile the launcher picked: 64 / 128 = rows of gemm.hip's tile, 130 / 192 / 256 = the 128 / 192 / 256-row tile of
            # gemm_ring.hip, 131 = the weights-stationary streaming kernel (gemm_ws.hip), 129 = the 16-wave weight-gradient kernel
            # (gemm_big_km.hip)
            tile = self._last_tile()
            key = ({0: "f32", 1: "bf16", 2: "fp8", 3: "bf8xfp8"}[g.dtype], "km" if g.a_kmajor else "rm", "km" if g.b_kmajor else "rm", tile)
            self.records.append((key, flops, s, e, shape, abytes))

        def events():
            return (self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True),
                    self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True))

        def timed(g):
            s, e = events()
            s.record()
            self._orig(g)
            e.record()
            record(g, s, e)

        def timed_wgrad(g, out, defer, extra_bytes=0, keep=()):
            # a weight gradient launched on its own: the event pair brackets the product kernel; the reduce of its partial tiles
            # (wgrad_reduce_k) runs after the pair.  Deferred products are only queued here: their group launch is timed below.
            if defer and ops._WG.group and ops._WG.enabled:
                self._orig_wgrad(g, out, True, extra_bytes, keep)
                return
            # (grouping is switched off for this call: with it on, defer=True would only QUEUE the product -- the flush below would then
            #  launch it through timed_group, and record() would count it a second time with an empty pair)
            s, e = events()
            grouped, ops._WG.group = ops._WG.group, False
            try:
                ops._WG._launch_pending()  # products queued before this one keep their order (timed by timed_group)
                ops._gemm_call, ops._WG.on_one = self._orig, None
                s.record()
                self._orig_wgrad(g, out, True, extra_bytes, keep)
                e.record()
            finally:
                ops._gemm_call, ops._WG.on_one = timed, timed_one
                ops._WG.group = grouped
            if not defer:
                ops.wgrad_flush()
            record(g, s, e)

        def timed_one(g, launch):
            # a product of a group that the grouped launch did not take (balanced stream: the encoder's Conv1d weight gradients), launched
            # on its own with its partial tiles in the workspace; 0: not in that form either (fs2_gemm follows: timed there)
            s, e = events()
            s.record()
            used = launch()
            e.record()
            if used > 0:
                record(g, s, e)
            return used

        def timed_group(descs, launch):
            # the weight gradients of one layer in one launch (fs2_wgrad_grouped): one event pair, the group's FLOPs and bytes summed
            s, e = events()
            ops._gemm_call = self._orig          # (a group that does not run as one falls back to single launches: not recorded twice)
            try:
                s.record()
                taken = launch()
                e.record()
            finally:
                ops._gemm_call = timed
            ws = [work(d) for d, t in zip(descs, taken) if t]
            if not ws:
                return
            shape = (sum(w[2][0] * w[2][1] * w[2][4] for w in ws), len(ws), max(w[2][2] for w in ws), 0, len(ws), 0, "group")
            self.records.append((("bf16", "km", "km", 129), sum(w[0] for w in ws), s, e, shape, sum(w[1] for w in ws)))
        self._orig_wgrad = ops._wgrad_call
        ops._gemm_call = timed
        ops._wgrad_call = timed_wgrad
        ops._WG.on_group = timed_group
        ops._WG.on_one = timed_one

    def remove(self):
        from transformer_tts_amd import ops
        ops._gemm_call = self._orig
        ops._wgrad_call = self._orig_wgrad
        ops._WG.on_group = ops._WG.on_one = None

    def summary(self):
        agg = {}
        for key, flops, s, e, shape, abytes in self.records:
            ms = max(s.elapsed_time(e) - self.overhead_ms, 1e-4)
            a = agg.setdefault(key, [0.0, 0.0, 0, 0.0])
            a[0] += flops; a[1] += ms; a[2] += 1; a[3] += abytes
        return agg

    def by_shape(self):
        """per product shape: launches, time, TFLOP/s, algorithmic MB per launch and GB/s, and the product's own roofline floor
        max(bytes / 8 TB/s, FLOP / 2.5 PFLOP/s) with the ratio measured / floor; a footer sums both per kernel family"""
        agg = {}
        for key, flops, s, e, shape, abytes in self.records:
            a = agg.setdefault(key[:3] + shape, [0.0, 0.0, 0, 0.0, key[3]])
            a[0] += flops; a[1] += max(s.elapsed_time(e) - self.overhead_ms, 1e-4); a[2] += 1; a[3] += abytes
        lines = ["variant M N K taps batch split epilogue(b=bias r=relu m=relu_mask +=residual sN=colstats a=accumulate f=fp32 out) | launches total_ms avg_us "
                 "TFLOP/s | algorithmic MB/launch GB/s | floor_us = max(MB / 8 TB/s, FLOP / 2.5 PFLOP/s) bound measured/floor"]
        fam = {}
        for k, (fl, ms, n, by, tile) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            t_hbm, t_mfma = by / 8.0e12 * 1e3, fl / 2.5e15 * 1e3          # ms over all launches of the shape
            floor = max(t_hbm, t_mfma)
            lines.append(f"{'/'.join(k[:3])} {k[3]} {k[4]} {k[5]} {k[6]} {k[7]} {k[8]} {k[9]} | {n} {ms:.2f} {ms * 1e3 / n:.1f} "
                         f"{fl / (ms * 1e-3) / 1e12:.1f} | {by / n / 1e6:.1f} {by / (ms * 1e-3) / 1e9:.0f} | {floor * 1e3 / n:.1f} "
                         f"{'hbm' if t_hbm >= t_mfma else 'mfma'} {ms / floor:.2f}")
            f = fam.setdefault(family(tile), [0.0, 0.0, 0.0, 0.0])
            f[0] += ms; f[1] += floor; f[2] += t_hbm; f[3] += t_mfma
        lines.append("family: measured_ms  sum of per-product floors_ms (hbm-only, mfma-only)  measured/floor   [over the instrumented steps]")
        for name, (ms, floor, th, tm) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
            lines.append(f"{name}: {ms:.2f}  {floor:.2f} ({th:.2f}, {tm:.2f})  {ms / floor:.2f}")
        tot = [sum(v[i] for v in fam.values()) for i in range(4)]
        lines.append(f"all GEMMs: {tot[0]:.2f}  {tot[1]:.2f} ({tot[2]:.2f}, {tot[3]:.2f})  {tot[0] / tot[1]:.2f}")
        return "\n".join(lines)


def cpu_baseline(hp, batch):
    """Oracle train step (fp32 eager PyTorch, reference dropout rates) on this host's cores."""
    from oracle import train as otrain
    from oracle.model import FastSpeech2 as OracleFS2
    torch.manual_seed(0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # the box's CPU share for one GPU
    torch.set_num_threads(cores)
    m = OracleFS2.from_hp(hp, dropout=hp.dropout, dropout_postnet=0.5, dropout_variance_adaptor=hp.dropout_variance_adaptor)
    m.train()
    opt = otrain.make_optimizer(m)
    cut = lambda n: tuple(b[:n] if torch.is_tensor(b) else b for b in batch)
    print("[cpu_baseline] warm-up step on 2 utterances ...", file=sys.stderr, flush=True)
    otrain.train_step(m, opt, 1, cut(2), hp.d_model_decoder)
    n_utt = 48                                # the whole config-2 batch: ~10 s per step on the box's 16 cores
    sample = cut(n_utt)
    frames = int(sample[5].sum())
    n_timed = 3                               # 1 full-size warm-up + 3 timed steps (~45 s of CPU work in all)
    print(f"[cpu_baseline] 1 warm-up + {n_timed} timed steps on {n_utt} utterances with {cores} threads ...", file=sys.stderr, flush=True)
    times = []
    for i in range(n_timed + 1):
        t0 = time.perf_counter()
        otrain.train_step(m, opt, 2 + i, sample, hp.d_model_decoder)
        dt = time.perf_counter() - t0
        print(f"[cpu_baseline] step {i}: {dt:.1f} s{' (warm-up, not counted)' if i == 0 else ''}", file=sys.stderr, flush=True)
        if i > 0:
            times.append(dt)
    med = sorted(times)[len(times) // 2]
    return dict(value=round(frames / med, 1), unit="mel-frames/s", cores=cores, kind="port",
                sample=f"median of {n_timed} train steps after one warm-up step (fwd+bwd+clip+Adam, fp32 eager PyTorch oracle, reference "
                       f"dropout rates) on the {n_utt} utterances of the config-2 batch (T_pad {sample[1].shape[1]}, {frames} valid mel "
                       f"frames): {med:.1f} s per step (min {min(times):.1f}, max {max(times):.1f})")


def ar_batch(seed, batch_size, r=1):
    """(text, mel, pos_text, pos_mel, text_lengths, mel_lengths, stop_token, None) as datasets_transformer.collate_fn hands them to the
    autoregressive trainer, made from the synthetic FastSpeech2 batch of the same seed: an all-zero go frame in front of every mel,
    frame counts rounded up to the reduction rate, mel pad -5.0 (no mean/variance files), longest utterance first"""
    import numpy as np
    from transformer_tts_amd import synthetic
    text, mel, pos_text, pos_mel, tl, ml = (x.numpy() for x in synthetic.benchmark_batch(seed, batch_size)[:6])
    order = np.argsort(-(ml + 1), kind="stable")
    frames = ml[order] + 1
    lens = -(-frames // r) * r
    T = int(-(-int(frames.max()) // r) * r)
    B = batch_size
    mel2 = np.full((B, T, mel.shape[2]), -5.0, np.float32)
    stop = np.ones((B, T), np.float32)
    pm = np.zeros((B, T), np.int64)
    for i, b in enumerate(order):
        n = int(frames[i])
        mel2[i, 0] = 0.0
        mel2[i, 1:n] = mel[b, :n - 1]
        stop[i, :n] = 0.0
        pm[i, :int(lens[i])] = np.arange(1, int(lens[i]) + 1)
    t = torch.from_numpy
    return (t(text[order]), t(mel2), t(pos_text[order]), t(pm), t(tl[order]), t(lens.astype(np.int64)), t(stop), None)


def bench_ar(args, hp, dev):
    """--workload cfg3: BASELINE.json configs[3] (a parity case, reported for completeness: never the headline).  A step = reference
    train.py:156-262 on one resident batch; timed region = hipGraph replay per batch shape (train.graphed_train_step; --no-graph: eager
    launches), the GEMM event timer runs over a few eager steps behind it as for configs[1]."""
    from transformer_tts_amd import train as T
    from transformer_tts_amd import train_fastspeech2 as TF
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.utils.utils import init_weight
    T.DEVICE = TF.DEVICE = dev
    torch.manual_seed(1234)
    model = T.build_model(hp)
    model.apply(init_weight)
    model.train()
    model = model.to(dev)
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=hp.clip)
    pool = [tuple(b.to(dev) if torch.is_tensor(b) else b for b in ar_batch(2024 + i, hp.batch_size, hp.reduction_rate)) for i in range(POOL)]
    frames = [int(b[5].sum()) - b[1].shape[0] for b in pool]            # valid mel frames without the go frames
    use_graph = not args.no_graph
    stepper = T.graphed_train_step(model, opt, hp) if use_graph else (lambda st, d: T.train_step(model, opt, st, d, hp))
    warm = max(args.warmup, 2 * POOL) if use_graph else args.warmup     # each shape: 1 eager + 1 capture before replay
    step = 1
    for i in range(warm):
        _, _, step = stepper(step, pool[i % POOL])
    TF.settle_gc(force=True)            # (as the trainers do after their first steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for i in range(args.steps):
        _, _, step = stepper(step, pool[(warm + i) % POOL])
        done += frames[(warm + i) % POOL]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer = GemmTimer()                 # roofline leg: eager, instrumented steps behind the timed region
    timer.calibrate()
    n_inst = min(args.steps, POOL)
    timer.reserve(2 * 700 * n_inst)
    timer.install()
    for i in range(n_inst):
        _, _, step = T.train_step(model, opt, step, pool[i % POOL], hp)
    torch.cuda.synchronize()
    timer.remove()
    return timer, dt, done, int(sum(b[1].shape[0] * b[1].shape[1] for b in pool) / len(pool)), use_graph, n_inst


def bench_line(args, timer, dt, done, world, warm, use_graph, padded, BATCH, cpu):
    """the ONE JSON line of the contract (metric, value, roofline of the dominant GEMM variant, cpu_baseline)"""
    agg = timer.summary()
    roof = None
    n_inst_steps = max(1, (min(args.steps, POOL) if use_graph else args.steps))
    if agg:
        # the dominant KERNEL: the instances of one kernel template (e.g. the 128- / 192- / 256-row tiles of the ring kernel) are one family
        fam = {}
        for k, v in agg.items():
            a = fam.setdefault((k[0], family(k[3])), [0.0, 0.0, 0, 0.0])
            for i in range(4):
                a[i] += v[i]
        key, (fl, ms, cnt, by) = max(fam.items(), key=lambda kv: kv[1][1])
        tflops = fl / (ms * 1e-3) / 1e12
        gbs = by / (ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tpath):       # HBM bytes per launch of this family from the PMC passes (tools/summarize_profiles.py)
            traffic = json.load(open(tpath)).get("by_family", {}).get("/".join(key))
        # the roof that binds the dominant kernel's launches on average: algorithmic FLOP per algorithmic byte
        # against the machine balance 2.5 PFLOP/s / 8 TB/s
        hbm_bound = (fl / by) < (PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9))
        names = {"ring": f"fs2_gemm_ring_kernel<{key[0]}{'' if key[0] == 'bf16' else ' operands, block-scaled 16x16x128 MFMA'}, 128 / 192 / 256 x 256 tiles, 16 waves, "
                         "LDS-DMA ring> A row-major B row-major (linear layers and implicit-GEMM Conv1d, forward and data gradient)",
                 "ws": f"fs2_gemm_ws_kernel<{key[0]}, weights-stationary 256-column tile (K = 256 in registers), activation rows streamed through a 4-deep LDS-DMA ring, 8 waves>",
                 "km": f"fs2_gemm_big_km_kernel<{key[0]}, 128x128 tile, 16 waves = 2 k-groups x 8 (fp8: 4 x 4), 4-deep LDS-DMA ring> A k-major B k-major; a launch = the weight "
                       "gradients of one layer (fs2_wgrad_grouped, up to 4 products) or one product; partial tiles of the k-split to a workspace, added by wgrad_reduce_k",
                 "4wave": f"gemm_kernel<{key[0]}, 64 / 128-row tiles, 4 waves> (batched attention products, N = 80, short reductions)"}
        roof = dict(bound="hbm" if hbm_bound else "mfma", kernel=names[key[1]],
                    achieved=round(gbs if hbm_bound else tflops, 2), peak=PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS,
                    unit="GB/s" if hbm_bound else "TFLOP/s",
                    frac=round((gbs / PEAK_HBM_GBS) if hbm_bound else (tflops / PEAK_BF16_TFLOPS), 4), traffic=traffic,
                    traffic_over_algorithmic=(round(traffic / (by / cnt), 3) if traffic else None),
                    launches=cnt, avg_launch_us=round(ms * 1e3 / cnt, 2), event_pair_overhead_us=round(timer.overhead_ms * 1e3, 2),
                    algorithmic_bytes_per_launch=round(by / cnt), algorithmic_flops_per_launch=round(fl / cnt),
                    achieved_tflops=round(tflops, 1), mfma_frac=round(tflops / PEAK_BF16_TFLOPS, 4),
                    achieved_gbs=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4),
                    family_ms_per_step=round(ms / n_inst_steps, 3),
                    gemm_ms_per_step=round(sum(v[1] for v in agg.values()) / n_inst_steps, 3),
                    all_families={"/".join(k): dict(tflops=round(v[0] / (v[1] * 1e-3) / 1e12, 1), mfma_frac=round(v[0] / (v[1] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                                    gbs=round(v[3] / (v[1] * 1e-3) / 1e9, 1), hbm_frac=round(v[3] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                                    ms_per_step=round(v[1] / n_inst_steps, 3), launches=v[2]) for k, v in fam.items()},
                    all_variants={"/".join(str(x) for x in k): dict(tflops=round(v[0] / (v[1] * 1e-3) / 1e12, 1), gbs=round(v[3] / (v[1] * 1e-3) / 1e9, 1),
                                                     ms=round(v[1], 2), launches=v[2]) for k, v in agg.items()})
        if args.workload == "cfg2":
            # path level: the step's algorithmic FLOPs -- 59.7 MFLOP per padded frame = 2.650 TFLOP at the
            # 44,400 padded frames of the config-2 batch -- over the measured step time, against the MFMA roof
            step_flop = 2.650e12 * padded / 44400.0
            roof["step_tflops"] = round(step_flop / (dt / args.steps) / 1e12, 1)
            roof["step_mfma_frac"] = round(step_flop / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4)
    line = {
        "metric": ("mel-frames/sec (train step) autoregressive Transformer-TTS d_model=256" if args.workload == "cfg3" else
                   "mel-frames/sec (train step) FastSpeech2 d_model=256"), "value": round(done / dt, 1),
        "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": warm,
        "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.fp32 else ("fp8" if args.fp8 else "bf16"), "data": "synthetic (fed from pinned host memory: PCIe-inclusive)" if args.from_host else "synthetic",
        "config": {"workload": ("BASELINE.json configs[1]: FastSpeech2 d_model=256, 4+4 FFT layers (H=2, k_enc=9, k_dec=1), "
                                "80-mel, batch 48/GPU (L_pad<=128, T_pad~925), fwd+bwd+clip+Adam, dropout 0.1/0.5/0.5")
                   if args.workload == "cfg2" else
                   ("BASELINE.json configs[3]: autoregressive Transformer-TTS (Models/transformer.py) d_model=256, 4+4 layers (H=2, k_enc=9, "
                    "k_dec=1), pre-net + encoder-decoder cross-attention + post-net statistics, 80-mel, batch 48 (L_pad<=128, T_pad~926 incl. the "
                    "go frame), reduction rate 1, fwd+bwd+clip+Adam, dropout 0.1 / pre-net 0.5; a parity case reported for completeness")
                   if args.workload == "cfg3" else
                   ("BASELINE.json configs[4]: FastSpeech2 d_model=512, 6+6 FFT layers (H=4, k_enc=9, k_dec=1), 80-mel, batch "
                    "64/GPU, fwd+bwd+clip+Adam, dropout 0.1/0.5/0.5, " + ("fp8 e4m3/e5m2 operands in the row-major GEMMs"
                                                                          if args.fp8 else "bf16 operands")),
                   "global_batch": BATCH * world, "parallelism": f"dp{world}",
                   "padded_frames_per_step": padded,
                   "attention": (("hp.return_attn=True: flash kernels (causal / rectangular) + the returned maps written after the fact (fs2_flash_attention_probs)" if args.return_attn
                                  else "hp.return_attn=False: flash kernels, causal (decoder self-attention) and rectangular (encoder-decoder) modes")
                                 if args.workload == "cfg3" else
                                 "hp.return_attn=True: LDS-strip kernels, attention maps written to HBM (exact-fp32 mode)" if args.fp32 else
                                 "hp.return_attn=True: flash kernels + the returned (B,N,H,t,t) maps written after the fact (fs2_flash_attention_probs)" if args.return_attn
                                 else "hp.return_attn=False: flash kernels (no (t x t) tensor in HBM; the loop never reads the maps)"),
                   "launch": "hipGraph replay per batch shape" if use_graph else "eager"},
        "roofline": roof, "cpu_baseline": cpu,
    }
    return line


def rank_stub(args):
    """FS2_BENCH_STUB=1: the rank side of `bench.py --gpus N` WITHOUT a GPU -- rendezvous (gloo), the barrier / MAX-over-ranks /
    SUM-of-frames protocol of the timed region and the one JSON line on rank 0, with a sleep in place of the train steps.  What the
    CPU test of the launcher glue runs (tests/test_trainer_cpu.py): the first real multi-GPU run must not die in this part."""
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps * (1 + rank))          # rank r is slower: the MAX must pick the last rank's time
    dist.barrier()
    dt = time.perf_counter() - t0
    t, n = torch.tensor([dt], dtype=torch.float64), torch.tensor([1000.0 * args.steps], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(float(n) / float(t), 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(float(t) * 1e3 / args.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "frames_all_ranks": float(n), "local_rank": int(os.environ.get("LOCAL_RANK", "-1"))}), flush=True)
    dist.destroy_process_group()


def launch_ranks(n):
    """start n rank processes of this script under torch.distributed.run (a child process; this one never initialises
    the GPU and only relays the exit code)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def shapes_leg(args, model, opt, hp, dev, BATCH):
    """N batches with N distinct (B, L_pad, T_pad) at the configs[1] frame budget (batch sizes BATCH - 4 .. BATCH + 4, fresh seeds until
    N distinct shapes exist), resident in HBM; epochs over them in a fixed shuffled order through GraphedTrainStep: epoch 1 = every shape
    at first sight (eager launches, allocator growing), epoch 2 = second sight (the stepper's capture policy decides), epochs 3-4 =
    steady state.  Reported per epoch: wall ms per step and ms per 44,400 padded frames (the fixed-shape headline's step size)."""
    import random as _random
    from transformer_tts_amd import synthetic
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep
    batches, seen, seed = [], set(), 50_000
    while len(batches) < args.shapes:
        bsz = BATCH - 4 + (seed % 9)
        b = synthetic.make_batch(seed, bsz)
        seed += 1
        key = (tuple(b[0].shape), tuple(b[1].shape))
        if key in seen:
            continue
        seen.add(key)
        batches.append(tuple(x.to(dev) if torch.is_tensor(x) else x for x in b))
    padded = [b[1].shape[0] * b[1].shape[1] for b in batches]
    frames = [int(b[5].sum()) for b in batches]
    graphed = GraphedTrainStep(model, opt, hp)
    order = list(range(len(batches)))
    step, epochs = 1, []
    # fixed-shape reference on the same process: eager, capture, then replays of batch 0
    for _ in range(4):
        graphed(step, batches[0]); step += 1
    from transformer_tts_amd.train_fastspeech2 import settle_gc
    settle_gc(force=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(32):
        graphed(step, batches[0]); step += 1
    torch.cuda.synchronize()
    fixed_ms = (time.perf_counter() - t0) * 1e3 / 32 * 44400.0 / padded[0]
    for ep in range(4):
        _random.Random(ep).shuffle(order)
        before = dict(graphed.stats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for n_done, i in enumerate(order):
            graphed(step, batches[i]); step += 1
            if os.environ.get("FS2_SHAPES_MEM") and n_done % 20 == 19:
                st = torch.cuda.memory_stats()
                print(f"epoch {ep + 1} step {n_done + 1}: allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, reserved "
                      f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB, graphs {len(graphed.graphs)}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        epochs.append({"epoch": ep + 1, "ms_per_step": round(dt * 1e3 / len(order), 3),
                       "ms_per_44400_padded_frames": round(dt * 1e3 * 44400.0 / sum(padded), 3),
                       "valid_frames_per_s": round(sum(frames) / dt, 1),
                       **{k: graphed.stats[k] - before[k] for k in graphed.stats}})
    print(json.dumps({"metric": "dynamic shapes: FastSpeech2 train step over distinct batch shapes", "distinct_shapes": len(batches),
                      "max_graphs": graphed.max_graphs, "fixed_shape_replay_ms_per_44400_padded_frames": round(fixed_ms, 3),
                      "epochs": epochs, "policy": getattr(graphed, "policy", None)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~2 s of warm-up + timed steps; the first ~0.3 s on an idle MI355X run ~9 % slow (clock ramp), so a short
    # warm-up under-reports a fresh box (measured: 16+16 steps 14.5 ms/step, the same process a second time 13.4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp32", action="store_true", help="exact-fp32 parity mode instead of bf16 (not the headline)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--no-gemm-timer", action="store_true",
                    help="with --no-graph: time the eager steps WITHOUT the event pair around every GEMM (the speed a never-repeating batch "
                         "shape trains at; no roofline block)")
    ap.add_argument("--gemm-report", type=str, default=None, help="write per-shape GEMM timings to this file")
    ap.add_argument("--no-overlap", action="store_true", help="(default) keep the weight-gradient GEMMs on the main stream")
    ap.add_argument("--overlap", action="store_true", help="run the weight-gradient GEMMs on a second stream (measured slower: DESIGN.md)")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4"], default="cfg2",
                    help="cfg2 = BASELINE.json configs[1] (the headline, default); cfg3 = configs[3] (autoregressive Transformer-TTS, same "
                         "sizes, eager); cfg4 = configs[4] (d_model 512, 6+6 layers, batch 64/GPU)")
    ap.add_argument("--return-attn", action="store_true",
                    help="hp.return_attn=True: keep the (B,N,H,t,t) attention maps (LDS-strip kernels) instead of the flash kernels")
    ap.add_argument("--from-host", action="store_true",
                    help="feed the timed steps from pinned host memory through the trainer's one-batch-ahead copy stream "
                         "(the PCIe-inclusive rate quoted in DESIGN.md; never the headline `value`)")
    ap.add_argument("--fp8", action="store_true", help="fp8 operand mode of the row-major GEMMs (configs[4]; not the headline)")
    ap.add_argument("--shapes", type=int, default=0,
                    help="dynamic-shape leg (SURVEY 8(f) N3: a corpus batched by a frame budget never repeats a (B, L_pad, T_pad) for "
                         "long): N batches of DISTINCT shapes at the configs[1] frame budget, three epochs through the graph stepper "
                         "(first sight eager, second captured, then replayed / evicted), ms per step reported per epoch beside the "
                         "fixed-shape replay; prints its own JSON line instead of the headline")
    args = ap.parse_args()

    import torch.distributed as dist
    from transformer_tts_amd import ops, synthetic
    from transformer_tts_amd.optim import FusedAdam
    from transformer_tts_amd.train_fastspeech2 import GraphedTrainStep, build_model, train_step
    from transformer_tts_amd.utils.utils import init_weight

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # parent: no HIP call has been made in this process
    if os.environ.get("FS2_BENCH_STUB") == "1":
        return rank_stub(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs the MI355X (the product has no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dp = os.environ.get("FS2_FORCE_DP", "0") == "1"      # exercise the RCCL path with a 1-rank group (testing)
    if world > 1 or force_dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    ops.lib()

    hp = bench_hp(amp=not args.fp32, workload=args.workload, fp8=args.fp8, return_attn=args.return_attn)
    BATCH = hp.batch_size
    if args.workload == "cfg3":
        assert world == 1, "configs[3] is a single-GPU parity case"
        timer, dt, done, padded, ar_graph, n_inst = bench_ar(args, hp, dev)
        line = bench_line(args, timer, dt, done, world, args.warmup, ar_graph, padded, BATCH, None)
        if line["roofline"] is not None:        # (the event timer ran over n_inst eager steps whatever the timed region was)
            line["roofline"]["gemm_ms_per_step"] = round(sum(v[1] for v in timer.summary().values()) / max(1, n_inst), 3)
        print(json.dumps(line), flush=True)
        return
    torch.manual_seed(1234)
    model = build_model(hp)
    model.apply(init_weight)
    model.train()
    model = model.to(dev)
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0)
    model.rt.overlap_wgrad = bool(args.overlap) and not args.no_overlap
    if world > 1 or force_dp:
        from transformer_tts_amd.parallel import DataParallel
        opt.dp = DataParallel(model, opt.arena)

    if args.shapes > 0:
        return shapes_leg(args, model, opt, hp, dev, BATCH)

    # synthetic batches, resident in HBM before the timed region; different data on every rank
    pool = [tuple(b.to(dev) if torch.is_tensor(b) else b for b in synthetic.benchmark_batch(2024 + 1000 * rank + i, BATCH))
            for i in range(POOL)]
    frames = [int(b[5].sum()) for b in pool]

    # hipGraph replay of the whole step (one graph per batch shape); the GEMM event timer needs eager launches,
    # so the roofline leg below re-runs a few eager, instrumented steps after the timed region
    # (also with RCCL: the all-reduces are captured into the graph; FS2_GRAPH_DP=0 keeps the multi-rank run eager)
    use_graph = not args.no_graph and ((world == 1 and not force_dp) or os.environ.get('FS2_GRAPH_DP', '1') == '1')
    graphed = GraphedTrainStep(model, opt, hp, eager_fallback=world > 1) if use_graph else None
    run = (lambda st, b: graphed(st, b)) if use_graph else (lambda st, b: train_step(model, opt, st, b, hp))
    step = 1
    warm = max(args.warmup, 2 * POOL) if use_graph else args.warmup      # each shape: 1 eager + 1 capture before replay
    for i in range(warm):
        run(step, pool[i % POOL])
        step += 1
    timer = GemmTimer()
    if not use_graph and not args.no_gemm_timer:
        timer.install()
    # what the shipped train_loop does after its first steps: the model, arenas, descriptor caches and captured graphs leave the
    # garbage collector's generations (a full collection walking them took 58 ms on the launch thread: DESIGN.md section 6)
    from transformer_tts_amd.train_fastspeech2 import settle_gc
    settle_gc(force=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    if args.from_host:      # the shipped trainer's input path: pinned host batches, copied one batch ahead on a second stream
        from transformer_tts_amd.train_fastspeech2 import STEP_INPUTS, DevicePrefetcher
        host_pool = [tuple(b.cpu().pin_memory() if torch.is_tensor(b) else b for b in bt) for bt in pool]
        feed = DevicePrefetcher([host_pool[(warm + i) % POOL] for i in range(args.steps)], dev, indices=STEP_INPUTS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _t = {"feed": 0.0, "run": 0.0}
        _it = iter(feed)
        for i in range(args.steps):
            _a = time.perf_counter()
            bt = next(_it)
            _b = time.perf_counter()
            run(step, bt)
            _c = time.perf_counter()
            _t["feed"] += _b - _a; _t["run"] += _c - _b
            done += frames[(warm + i) % POOL]
            step += 1
        if os.environ.get("FS2_BENCH_HOST_TIMES"):
            print("host ms/step: feed %.3f run %.3f" % (_t["feed"] / args.steps * 1e3, _t["run"] / args.steps * 1e3), file=sys.stderr)
    else:
        for i in range(args.steps):
            run(step, pool[(warm + i) % POOL])
            done += frames[(warm + i) % POOL]
            step += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_graph:       # roofline leg: the same kernels, launched eagerly with a HIP event pair around every GEMM;
        model.rt.overlap_wgrad = False      # one stream only, so that a kernel's event pair times that kernel alone
        # An event pair also counts any time the stream sits idle between the start event and the kernel, i.e. whenever
        # the Python launcher falls behind the GPU. Each instrumented step therefore starts with a spin kernel long
        # enough for the host to enqueue the whole step ahead of the GPU, so every pair brackets back-to-back work.
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(); torch.cuda._sleep(20_000_000); c1.record(); torch.cuda.synchronize()
        cycles_per_ms = 20_000_000 / max(c0.elapsed_time(c1), 1e-3)
        train_step(model, opt, step, pool[0], hp); step += 1          # (first eager step after the replays: allocator warm-up)
        torch.cuda.synchronize()
        t_e = time.perf_counter()
        train_step(model, opt, step, pool[0], hp); step += 1          # host time of one eager step WITHOUT the event pairs (config.eager_host_ms_per_step)
        host_ms = (time.perf_counter() - t_e) * 1e3
        timer.install()
        torch.cuda.synchronize()
        timer.records.clear()
        timer.calibrate()
        n_inst = min(args.steps, POOL)
        timer.reserve(2 * 400 * n_inst)
        for i in range(n_inst):
            torch.cuda._sleep(int(cycles_per_ms * min(60.0, 1.5 * host_ms + 5.0)))
            train_step(model, opt, step, pool[i % POOL], hp)
            step += 1
        torch.cuda.synchronize()
    if use_graph or not args.no_gemm_timer:
        timer.remove()
    attn_true_ms = None
    if world == 1 and not force_dp and args.workload == "cfg2" and use_graph and not args.return_attn and not args.fp32 and not args.from_host:
        # the drop-in DEFAULT, hp.return_attn = True (the 14-tuple's attention maps, reference Models/encoder.py:97,105), timed behind the
        # headline region on the same model and batches: LDS-strip attention kernels, probabilities written to HBM
        model.rt.return_attn = True
        g2 = GraphedTrainStep(model, opt, hp)
        for i in range(2 * POOL):
            g2(step, pool[i % POOL]); step += 1
        n2 = max(POOL, min(args.steps, 3 * POOL))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n2):
            g2(step, pool[i % POOL]); step += 1
        torch.cuda.synchronize()
        attn_true_ms = (time.perf_counter() - t1) * 1e3 / n2
        model.rt.return_attn = False
        del g2
    if args.gemm_report and rank == 0:
        with open(args.gemm_report, "w") as f:
            f.write(timer.by_shape() + "\n")
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        n = torch.tensor([float(done)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        dt, done = float(t.item()), float(n.item())

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.workload == "cfg2":
            cpu = cpu_baseline(hp, synthetic.benchmark_batch(2024, 48))
        line = bench_line(args, timer, dt, done, world, warm, use_graph, int(sum(b[1].shape[0] * b[1].shape[1] for b in pool) / len(pool)), BATCH, cpu)
        if attn_true_ms is not None:
            line["config"]["return_attn_true_ms_per_step"] = round(attn_true_ms, 3)
        if use_graph and world == 1:
            line["config"]["eager_host_ms_per_step"] = round(host_ms, 3)      # host time of ONE eager step (Python launch path), cf. ms_per_step
        if getattr(opt, "dp", None) is not None:        # the gradient all-reduce schedule of the last eager step (bytes, launched from where)
            plan = opt.dp.describe_plan()
            line["config"]["dp_plan"] = {"buckets": [[q["bytes"], q["launched"]] for q in plan],
                                         "bytes_in_backward": sum(q["bytes"] for q in plan if q["launched"] == "backward"),
                                         "bytes_total": sum(q["bytes"] for q in plan)}
        print(json.dumps(line), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

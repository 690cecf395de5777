"""Flat fp32 parameter / gradient / Adam-moment arenas and the fused clip + Adam step.

Replaces ``torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)`` + ``torch.optim.Adam(lr=1e-3,
betas=(0.9, 0.98), eps=1e-9).step()`` of the reference trainer (train_fastspeech2.py:304-315,411-416)
by two kernel launches over one contiguous buffer (fs2_sqnorm, fs2_adam_step).  Parameters become
views of the arena (state_dict keys / shapes unchanged); ``param.grad`` are views of the gradient
arena that the backward kernels accumulate into directly, so data-parallel all-reduce works on
arena slices in place.
"""
import torch

from . import ops


class ParamArena:
    ALIGN = 4   # elements: keeps every parameter 16-byte aligned for float4 loads
    ZTAIL = 16384

    def __init__(self, params, kernel_layout_grads=True):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no parameters"
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.p = torch.zeros(off, dtype=torch.float32, device=dev)
        # the gradient arena and, behind it, ZTAIL floats for the small zero-initialised accumulators of a step (loss terms, BatchNorm
        # sums, the gradient norm): optimizer.zero_grad() clears both with ONE launch (fs2_zero) instead of one fill per buffer
        self._gfull = torch.zeros(off + self.ZTAIL, dtype=torch.float32, device=dev)
        self.g = self._gfull[:off]
        self.ztail = self._gfull[off:]
        segs = []
        for p, o in zip(self.params, self.offsets):
            view = self.p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            if kernel_layout_grads and p.dim() == 3 and p.shape[2] > 1:
                # Conv1d weight (O,I,k): its gradient slot holds [o][j][i], the layout the weight-gradient GEMM writes
                # (p._fs2_grad_raw); param.grad is the (O,I,k) VIEW of it, so every reader sees the reference's layout and only
                # the fused Adam kernel -- which works on the flat arena -- needs the segment table
                O, I, k = p.shape
                p._fs2_grad_raw = self.g[o:o + p.numel()].view(O, k * I)
                p._fs2_grad = self.g[o:o + p.numel()].view(O, k, I).permute(0, 2, 1)
                segs.append([o, o + p.numel(), O, I, k])
            else:
                p._fs2_grad = self.g[o:o + p.numel()].view_as(p)
            p.grad = p._fs2_grad
        self.perm = torch.tensor(segs, dtype=torch.int64, device=dev).reshape(-1, 5) if segs else None
        self._index = {id(p): i for i, p in enumerate(self.params)}

    def span(self, params):
        """[lo, hi) element range of the arena covered by `params` (must be registered here)"""
        idx = [self._index[id(p)] for p in params if id(p) in self._index]
        if not idx:
            return None
        lo = min(self.offsets[i] for i in idx)
        hi = max(self.offsets[i] + (self.params[i].numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN for i in idx)
        return lo, hi


_HYPER = {}


def _hyper_ring():
    """the pinned staging rows of FusedAdam.host_update: ONE ring per process, never freed (a pinned block that the garbage collector
    releases while some stream is capturing a hipGraph aborts the process: the host allocator's free path queries events)"""
    if not _HYPER:
        _HYPER.update(pin=torch.empty((FusedAdam.HYPER_RING, 4), dtype=torch.float32).pin_memory(), ev=[None] * FusedAdam.HYPER_RING, next=0)
    return _HYPER


class FusedAdam:
    """Adam with the reference's hyper-parameters, global-norm clipping fused in, over a ParamArena.

    Keeps the pieces of the torch.optim interface the reference trainer touches:
    ``param_groups[i]['lr']`` (set every step by the Noam schedule), ``zero_grad()``, ``step()``,
    ``state_dict()`` / ``load_state_dict()`` in torch.optim.Adam's format (``state[0]['step']`` is what
    the reference reads on resume, train_fastspeech2.py:444)."""
    HYPER_RING = 16

    def __init__(self, model_or_params, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, max_norm=1.0, runtime=None, dp=None):
        params = list(model_or_params.parameters()) if hasattr(model_or_params, "parameters") else list(model_or_params)
        self.arena = ParamArena(params)
        self.runtime = runtime if runtime is not None else getattr(model_or_params, "rt", None)
        self.dp = dp
        self.betas, self.eps, self.max_norm = betas, eps, max_norm
        dev = self.arena.p.device
        self.m = torch.zeros_like(self.arena.p)
        self.v = torch.zeros_like(self.arena.p)
        self.gsq = self.arena.ztail[:1]            # (cleared by zero_grad with the gradients; launch() clears it itself if called twice)
        self._gsq_clean = False
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self.t = 0
        self.param_groups = [dict(params=self.arena.params, lr=lr, betas=betas, eps=eps)]
        if self.runtime is not None:
            self.runtime.invalidate()

    def zero_grad(self, set_to_none=False):
        # a step that raised inside its backward may have left weight-gradient products queued (ops._WG): they must not be added
        # into the gradients of this step (after a complete step nothing is queued and this is a no-op)
        if hasattr(ops, "wgrad_reset"):
            ops.wgrad_reset()
        ops.zero(self.arena._gfull)             # gradients + the step's small accumulators (Runtime.zsmall) in one launch
        self._gsq_clean = True
        if self.runtime is not None:
            self.runtime.zpool_reset(self.arena.ztail[4:])
        for p in self.arena.params:     # re-attach if a caller dropped the views
            if p.grad is None:
                p.grad = p._fs2_grad

    def step(self):
        self.host_update()
        self.launch()

    def host_update(self):
        """host half of a step: advance t and refresh the 4-float device buffer {lr, 1-b1^t, 1-b2^t, 1/world}
        (kept outside a captured hipGraph; the kernels read the buffer)"""
        self.t += 1
        world = self.dp.world if self.dp is not None else 1
        lr = float(self.param_groups[0]["lr"])
        b1, b2 = self.betas
        vals = [lr, 1.0 - b1 ** self.t, 1.0 - b2 ** self.t, 1.0 / world]
        if not self.hyper.is_cuda:
            self.hyper.copy_(torch.tensor(vals, dtype=torch.float32))
            return
        # Asynchronous, from a small ring of PINNED staging rows: a copy from pageable host memory returns only when the stream has
        # executed it, i.e. it made the training thread wait for the whole previous step at the start of every step -- the GPU then idled
        # while the host caught up (eager steps: 9.2 ms against 7.6 ms of device work; graph replays: the gap between two replays).
        # A row is rewritten only after the copy that read it last has completed (its event).  The ring belongs to the process.
        ring = _hyper_ring()
        i = ring["next"]
        ring["next"] = (i + 1) % self.HYPER_RING
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()
        row = ring["pin"][i]
        row[0], row[1], row[2], row[3] = vals
        self.hyper.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring["ev"][i] = ev

    def launch(self):
        """device half of a step (graph-capturable): finish the gradient exchange, global norm, clip + Adam"""
        if self.dp is not None:
            self.dp.finish()
        b1, b2 = self.betas
        if not self._gsq_clean:
            self.gsq.zero_()
        self._gsq_clean = False
        ops.sqnorm(self.arena.g, self.gsq)
        ops.adam_step(self.arena.p, self.arena.g, self.m, self.v, self.hyper, self.gsq, b1, b2, self.eps,
                      self.max_norm if self.max_norm is not None else 0.0, perm=self.arena.perm)
        if self.runtime is not None:
            self.runtime.invalidate()

    # ---- torch.optim.Adam-compatible checkpoints
    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.arena.params, self.arena.offsets)):
            n = p.numel()
            state[i] = dict(step=torch.tensor(float(self.t)), exp_avg=self.m[o:o + n].view_as(p).clone(),
                            exp_avg_sq=self.v[o:o + n].view_as(p).clone())
        groups = [dict(lr=self.param_groups[0]["lr"], betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False,
                       params=list(range(len(self.arena.params))))]
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        """torch.optim.Adam's format.  A parameter WITHOUT an entry in sd["state"] (torch's Adam creates the state of a parameter at its
        first gradient: the post-net convolutions of the autoregressive model never get one in the reference, train.py / postnets.py)
        starts from zero moments, exactly what the reference's optimizer would do if such a parameter ever received a gradient."""
        state = sd["state"]
        steps = []
        for i, (p, o) in enumerate(zip(self.arena.params, self.arena.offsets)):
            st = state.get(i)
            n = p.numel()
            if st is None:
                self.m[o:o + n].zero_()
                self.v[o:o + n].zero_()
                continue
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.append(int(st["step"]))
        self.t = max(steps) if steps else 0
        self.param_groups[0]["lr"] = sd["param_groups"][0]["lr"]

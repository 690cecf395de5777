"""FS2_CHECK_BOUNDS=1: host-side validation of every GEMM / weight-gradient / flash-attention descriptor before its launch (a debugging
aid: the product path is unchanged without the switch).

Every operand of an FS2Gemm / FS2WgradPart is turned into the byte range [ptr, ptr + extent) its kernel may touch -- extents computed
as the kernels' own guards do (16-byte chunks: a row is read up to the chunk that holds its last element; batch strides; the taps of a
Conv1d read rows of the SAME matrix) -- and the range must be non-null and lie inside ONE live block of PyTorch's caching allocator
(`torch.cuda.memory_snapshot()`), which is where every buffer of this package comes from.  A violation raises before the launch.
"""
import bisect
import os

import torch

ENABLED = os.environ.get("FS2_CHECK_BOUNDS", "0") == "1"
STATS = {"descriptors": 0, "ranges": 0, "refreshes": 0}
_blocks = {"starts": [], "ends": [], "age": 0}

if ENABLED:
    import atexit
    import sys
    atexit.register(lambda: print(f"FS2_CHECK_BOUNDS: {STATS['descriptors']} descriptors, {STATS['ranges']} operand ranges checked, "
                                  f"{STATS['refreshes']} allocator snapshots, no violation", file=sys.stderr, flush=True)
                    if STATS["descriptors"] else None)


def _refresh():
    starts, ends = [], []
    for seg in torch.cuda.memory_snapshot():
        addr = seg["address"]
        for b in seg["blocks"]:
            a = b.get("address", addr)
            if b["state"] == "active_allocated":
                starts.append(a)
                ends.append(a + b["size"])
            addr = a + b["size"]
    order = sorted(range(len(starts)), key=starts.__getitem__)
    _blocks["starts"] = [starts[i] for i in order]
    _blocks["ends"] = [ends[i] for i in order]
    _blocks["age"] = 0
    STATS["refreshes"] += 1


def _inside(ptr, nbytes):
    i = bisect.bisect_right(_blocks["starts"], ptr) - 1
    return i >= 0 and ptr + nbytes <= _blocks["ends"][i]


def check_ranges(ranges, what):
    """ranges: [(name, ptr, nbytes)]"""
    if torch.cuda.is_current_stream_capturing():
        return
    _blocks["age"] += 1
    if _blocks["age"] > 64 or not _blocks["starts"]:
        _refresh()
    for r in ranges:
        name, ptr, nbytes = r[:3]
        STATS["ranges"] += 1
        if not ptr:
            raise RuntimeError(f"FS2_CHECK_BOUNDS: {what}: operand {name} is a null pointer")
        if nbytes <= 0:
            raise RuntimeError(f"FS2_CHECK_BOUNDS: {what}: operand {name} has extent {nbytes}")
        if not _inside(ptr, nbytes):
            _refresh()          # (the block may be younger than the cached snapshot)
            items = r[3] if len(r) > 3 else None
            if items and all(_inside(q[1], q[2]) for q in items):
                continue
            if not _inside(ptr, nbytes):
                raise RuntimeError(f"FS2_CHECK_BOUNDS: {what}: operand {name} [{ptr:#x}, +{nbytes}) is not inside one live allocation")


def gemm_ranges(g):
    """byte ranges an fs2_gemm / fs2_wgrad_* launch of descriptor g may read or write (include/fs2_hip.h, FS2Gemm)"""
    es = {0: 4, 1: 2, 2: 1, 3: 1}[g.dtype]
    chunk = 16 // es
    b1, b2 = max(1, g.batch1), max(1, g.batch2)
    taps = g.taps if g.conv == 1 else 1
    nb2_ab = 1 if g.conv == 2 else b2            # conv = 2: batch2 enumerates taps, which shift rows inside the same matrices
    kb = g.Kb if g.Kb > 0 else g.K
    ru = lambda n: -(-n // chunk) * chunk
    out = []

    def span(name, ptr, rows, cols, ld, s1, s2, n2, esz):
        # one range over all batch items, and the per-item ranges behind it: check_ranges accepts the operand when the whole span
        # sits in one allocation, or else when every batch item does (gradients of separately allocated parameters that happen to
        # sit at a constant address stride are batched by wgrad_batched: the gaps between them are never touched)
        one = ((rows - 1) * ld + cols) * esz
        last = (b1 - 1) * abs(s1) + (n2 - 1) * abs(s2)
        items = None
        if ptr and last > 0:
            items = [(name, ptr + (i * s1 + j * s2) * esz, one) for i in range(b1) for j in range(n2)]
        out.append((name, ptr, last * esz + one, items))
    if g.a_kmajor:
        span("A", g.A, g.K, ru(g.M), g.lda, g.sA1, g.sA2, nb2_ab, es)
    else:
        span("A", g.A, g.M, ru(g.K), g.lda, g.sA1, g.sA2, nb2_ab, es)
    if g.b_kmajor:
        span("B", g.B, kb, ru(g.N), g.ldb, g.sB1, g.sB2, nb2_ab, es)
    else:
        span("B", g.B, g.N, ru(g.K * taps), g.ldb, g.sB1, g.sB2, nb2_ab, es)
    cs = 2 if g.c_dtype == 1 else 4
    if g.accumulate == 2:       # sliced split-K: split_k slices, sC1 elements apart
        out.append(("C", g.C, ((max(1, g.split_k) - 1) * g.sC1 + (g.M - 1) * g.ldc + g.N) * cs))
    else:
        span("C", g.C, g.M, g.N, g.ldc, g.sC1, g.sC2, b2, cs)
    if g.bias:
        out.append(("bias", g.bias, 4 * g.N))
    if g.residual:
        span("residual", g.residual, g.M, g.N, g.ldr, g.sC1, g.sC2, b2, 2 if g.res_dtype == 1 else 4)
    if g.relu_mask:
        span("relu_mask", g.relu_mask, g.M, g.N, g.ldm, g.sC1, g.sC2, b2, 4 if g.dtype == 0 else 2)
    if g.colstats:
        out.append(("colstats", g.colstats, 4 * g.N * (1 if g.colstats_mode == 1 else 2)))
    for nm in ("scale_a", "scale_b"):
        if getattr(g, nm):
            out.append((nm, getattr(g, nm), 4))
    if g.q8:
        out.append(("q8", g.q8, (g.M - 1) * g.ldc + g.N))
        out.append(("q8_state", g.q8_state, 8))
        out.append(("q8_prev", g.q8_prev, 4))
    return out


def check_gemm(g, what="fs2_gemm"):
    STATS["descriptors"] += 1
    assert g.M > 0 and g.N > 0 and g.K > 0, f"FS2_CHECK_BOUNDS: {what}: empty product {g.M} x {g.N} x {g.K}"
    check_ranges(gemm_ranges(g), f"{what} M={g.M} N={g.N} K={g.K} batch={g.batch1}x{g.batch2} conv={g.conv} km={g.a_kmajor}{g.b_kmajor}")


def part_ranges(p):
    """ranges of one FS2WgradPart: the partial tiles fs2_wgrad_reduce reads and the gradient it adds them into"""
    tiles = p.tilesM * p.tilesN
    n2 = max(1, p.n2)
    n1 = max(1, p.nbatch // n2)
    one = 4 * ((p.M - 1) * p.ldc + p.N)
    last = (n1 - 1) * abs(p.sC1) + (n2 - 1) * abs(p.sC2)
    items = [("part.dst", p.dst + 4 * (i * p.sC1 + j * p.sC2), one) for i in range(n1) for j in range(n2)] if (p.dst and last > 0) else None
    if p.splits > 0:
        slices = p.nbatch * p.splits * tiles
    else:               # balanced stream: slice (workgroup + tile); -splits units per workgroup, reserved & (2^30 - 1) stages per tile
        nstk = p.reserved & ((1 << 30) - 1)
        slices = -(-(p.nbatch * tiles * nstk) // -p.splits) + p.nbatch * tiles
    out = [("part.ws", p.ws, 4 * slices * 128 * 128), ("part.dst", p.dst, 4 * last + one, items)]
    for nm in ("scale_a", "scale_b"):
        if getattr(p, nm):
            out.append((nm, getattr(p, nm), 4))
    return out


def check_part(p, ws_ptr, ws_bytes, what="fs2_wgrad_reduce"):
    STATS["descriptors"] += 1
    assert p.splits != 0 and (p.splits > 0 or 1 <= -p.splits <= (p.reserved & ((1 << 30) - 1))) and p.tilesM >= 1 and p.tilesN >= 1 and p.nbatch >= 1, f"FS2_CHECK_BOUNDS: {what}: bad part"
    assert p.tilesM == -(-p.M // 128) and p.tilesN == -(-p.N // 128), f"FS2_CHECK_BOUNDS: {what}: tile counts do not match M, N"
    r = part_ranges(p)
    lo, n = r[0][1], r[0][2]
    if not (ws_ptr <= lo and lo + n <= ws_ptr + ws_bytes):
        raise RuntimeError(f"FS2_CHECK_BOUNDS: {what}: partial tiles [{lo:#x}, +{n}) outside the workspace [{ws_ptr:#x}, +{ws_bytes})")
    check_ranges(r, what)


def flash_ranges(d, backward, keep_words, probs=None, probs_batch=0):
    """byte ranges a flash attention launch of descriptor d (include/fs2_hip.h, FS2FlashAttn) may read or write: rows of d.dk contiguous
    bf16 at base + b * batch_stride + h * head_stride + row * row_stride for b < B, h < H, row < tq (queries: q, o, d_out, dq) or tk (keys:
    k, v, dk_out, dv_out); the key mask, the row statistics, the keep-bit stash, aux, the bias-gradient vectors and the map"""
    B, H, tq, tk = d.B, d.H, d.tq, d.tk

    def rows(name, ptr, n, row, batch, esz=2):
        return (name, ptr, ((B - 1) * abs(batch) + (H - 1) * abs(d.head_stride) + (n - 1) * abs(row) + d.dk) * esz)
    out = [rows("q", d.q, tq, d.q_row_stride, d.q_batch_stride), rows("k", d.k, tk, d.kv_row_stride, d.kv_batch_stride),
           rows("v", d.v, tk, d.kv_row_stride, d.kv_batch_stride), rows("o", d.o, tq, d.o_row_stride, d.o_batch_stride),
           ("key_mask", d.key_mask, B * tk), ("stats", d.stats, B * H * tq * 2 * 4)]
    if d.key_info:
        out.append(("key_info", d.key_info, B * 3 * 4))
    if d.p > 0:
        out.append(("keep_bits", d.keep_bits, 2 * keep_words))
        if not backward and not d.pregenerated:
            out.append(("rng", d.rng, 16))
    if backward:
        out += [rows("d_out", d.d_out, tq, d.do_row_stride, d.do_batch_stride), ("aux", d.aux, B * H * tq * 4 * 4),
                rows("dq", d.dq, tq, d.dq_row_stride, d.dq_batch_stride), rows("dk_out", d.dk_out, tk, d.dkv_row_stride, d.dkv_batch_stride),
                rows("dv_out", d.dv_out, tk, d.dkv_row_stride, d.dkv_batch_stride)]
        for nm in ("dbias_q", "dbias_k", "dbias_v"):
            if getattr(d, nm):
                out.append((nm, getattr(d, nm), H * d.dk * 4))
    if probs is not None:
        out.append(("probs", probs, ((B - 1) * abs(probs_batch) + H * tq * d.tkp) * 2))
    return out


def check_flash(d, what, backward=False, keep_words=0, probs=None, probs_batch=0):
    STATS["descriptors"] += 1
    assert d.B > 0 and d.H > 0 and d.tq > 0 and d.tk > 0 and d.dk in (64, 96, 128) and d.tkp == -(-d.tk // 8) * 8, \
        f"FS2_CHECK_BOUNDS: {what}: bad shape B={d.B} H={d.H} tq={d.tq} tk={d.tk} dk={d.dk} tkp={d.tkp}"
    check_ranges(flash_ranges(d, backward, keep_words, probs, probs_batch),
                 f"{what} B={d.B} H={d.H} tq={d.tq} tk={d.tk} dk={d.dk} causal={d.causal} p={d.p:.3g}")

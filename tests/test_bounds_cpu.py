"""Host logic of the FS2_CHECK_BOUNDS validator (transformer_tts_amd/bounds.py) and of the balanced-stream slice numbering it mirrors
(csrc/gemm_big_km.hip: km_body writes slice = workgroup + tile, wgrad_reduce_k reads the slices L + z of the workgroups whose unit
range meets tile z): pure index arithmetic, no GPU."""
import ctypes
import itertools

import pytest

from transformer_tts_amd import bounds
from transformer_tts_amd.ops import FS2FlashAttn, FS2WgradPart


def stream_writes(base, nstk, U):
    """(workgroup, tile) pairs km_body stores in the balanced-stream form: workgroup L owns units [L*U, (L+1)*U) of the tile-major list
    of base x nstk stage units"""
    total = base * nstk
    out = []
    for L in range(-(-total // U)):
        ubeg, uend = L * U, min((L + 1) * U, total)
        t0, s0 = divmod(ubeg, nstk)
        tiles = [t0] if s0 + (uend - ubeg) <= nstk else [t0 + 1, t0]      # (head of the next tile first, as the kernel runs them)
        out += [(L, z) for z in tiles]
    return out


@pytest.mark.parametrize("base,nstk", [(144, 48), (144, 8), (18, 64), (1, 16), (7, 5), (255, 33), (100, 100)])
def test_balanced_stream_slices_are_unique_and_the_reduce_reads_exactly_them(base, nstk):
    total = base * nstk
    U = -(-total // 256)
    if U > nstk:
        pytest.skip("a workgroup would span three tiles: the host does not choose the stream form")
    writes = stream_writes(base, nstk, U)
    slices = [L + z for L, z in writes]
    assert len(set(slices)) == len(slices)                      # no two partial tiles share a slice
    by_tile = {}
    for L, z in writes:
        by_tile.setdefault(z, set()).add(L + z)
    for z in range(base):                                       # wgrad_reduce_k's loop bounds
        L0, L1 = (z * nstk) // U, (z * nstk + nstk - 1) // U
        assert {L + z for L in range(L0, L1 + 1)} == by_tile[z], z
    # the extent the validator computes for the workspace = one past the highest slice
    p = FS2WgradPart()
    p.M = p.N = 128
    p.tilesM = p.tilesN = 1
    p.nbatch, p.n2, p.splits, p.reserved = base, 1, -U, nstk
    p.ws, p.dst, p.ldc = 1 << 20, 1 << 30, 128
    ws = bounds.part_ranges(p)[0]
    assert ws[2] // (4 * 128 * 128) >= max(slices) + 1
    assert ws[2] // (4 * 128 * 128) <= -(-total // U) + base


def test_flash_ranges_cover_exactly_the_rows_the_kernels_address():
    B, H, tq, tk, dk = 3, 2, 37, 53, 64
    d = FS2FlashAttn()
    # q: (B, tq, 3, H, dk) fused projection, the q part; k / v: separate (B, tk, H, dk) tensors; o: (B, tq, H, dk)
    d.q, d.k, d.v, d.o = 1 << 20, 1 << 24, 1 << 26, 1 << 28
    d.q_row_stride, d.q_batch_stride = 3 * H * dk, tq * 3 * H * dk
    d.kv_row_stride, d.kv_batch_stride = H * dk, tk * H * dk
    d.o_row_stride, d.o_batch_stride = H * dk, tq * H * dk
    d.head_stride, d.dk = dk, dk
    d.key_mask, d.stats, d.keep_bits, d.rng = 1 << 30, 1 << 31, 1 << 32, 1 << 33
    d.B, d.H, d.tq, d.tk, d.tkp, d.p = B, H, tq, tk, 56, 0.1
    r = {x[0]: x for x in bounds.flash_ranges(d, False, keep_words=1234)}

    def last(row_stride, batch_stride, n):
        return max(b * batch_stride + h * dk + i * row_stride + dk - 1 for b, h, i in itertools.product(range(B), range(H), range(n)))
    assert r["q"][2] == 2 * (last(d.q_row_stride, d.q_batch_stride, tq) + 1)
    assert r["k"][2] == r["v"][2] == 2 * (last(d.kv_row_stride, d.kv_batch_stride, tk) + 1)
    assert r["o"][2] == 2 * (last(d.o_row_stride, d.o_batch_stride, tq) + 1)
    assert r["key_mask"][2] == B * tk and r["stats"][2] == B * H * tq * 8 and r["keep_bits"][2] == 2468 and r["rng"][2] == 16
    assert "aux" not in r and "dq" not in r
    d.d_out, d.aux, d.dq, d.dk_out, d.dv_out, d.dbias_q = 1 << 34, 1 << 35, 1 << 36, 1 << 37, 1 << 38, 1 << 39
    d.do_row_stride, d.do_batch_stride = H * dk, tq * H * dk
    d.dq_row_stride, d.dq_batch_stride = 3 * H * dk, tq * 3 * H * dk
    d.dkv_row_stride, d.dkv_batch_stride = 3 * H * dk, tk * 3 * H * dk
    r = {x[0]: x for x in bounds.flash_ranges(d, True, keep_words=1234, probs=1 << 40, probs_batch=4 * H * tq * 56)}
    assert r["dk_out"][2] == r["dv_out"][2] == 2 * (last(d.dkv_row_stride, d.dkv_batch_stride, tk) + 1)
    assert r["dq"][2] == 2 * (last(d.dq_row_stride, d.dq_batch_stride, tq) + 1)
    assert r["aux"][2] == B * H * tq * 16 and r["dbias_q"][2] == H * dk * 4 and "dbias_k" not in r and "rng" not in r
    assert r["probs"][2] == 2 * ((B - 1) * 4 * H * tq * 56 + H * tq * 56)


def test_a_range_outside_every_live_block_is_reported(monkeypatch):
    monkeypatch.setattr(bounds.torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(bounds, "_refresh", lambda: None)
    monkeypatch.setitem(bounds._blocks, "starts", [1000, 5000])
    monkeypatch.setitem(bounds._blocks, "ends", [2000, 9000])
    monkeypatch.setitem(bounds._blocks, "age", 0)
    bounds.check_ranges([("a", 1000, 1000), ("b", 5500, 3500)], "unit")
    with pytest.raises(RuntimeError, match="not inside one live allocation"):
        bounds.check_ranges([("a", 1500, 501)], "unit")
    with pytest.raises(RuntimeError, match="null pointer"):
        bounds.check_ranges([("a", 0, 4)], "unit")
    # a batched operand whose items sit in separate blocks is accepted item by item
    bounds.check_ranges([("c", 1000, 6000, [("c", 1000, 500), ("c", 5000, 2000)])], "unit")
    assert ctypes.sizeof(FS2WgradPart) > 0

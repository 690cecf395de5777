"""The C-ABI library builds, loads on a machine without a GPU and exports every symbol that
include/fs2_hip.h declares; the ctypes table in ops.py covers the same set; the product has no CPU path."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fs2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs2_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from transformer_tts_amd import build, ops
    path = build.build_library(verbose=False)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fs2_hip.h but not exported by {path}"
    table = set(ops.SIGNATURES) | {"fs2_last_error", "fs2_abi_version"}
    assert table == set(names), f"ops.SIGNATURES out of sync with the header: {table ^ set(names)}"
    assert ops.lib().fs2_abi_version() == 9


def test_argument_errors_are_reported_not_crashed():
    """host-side validation runs without a GPU: bad descriptors return FS2_EINVAL with a message"""
    from transformer_tts_amd import ops
    lib = ops.lib()
    g = ops.FS2Gemm()
    g.M, g.N, g.K, g.dtype = 8, 8, 8, 7
    rc = lib.fs2_gemm(ctypes.byref(g), None)
    assert rc == -1 and b"dtype" in lib.fs2_last_error()
    # split-K without the fp32 accumulate epilogue / accumulate with a fused activation are contradictions
    g = ops.FS2Gemm()
    g.M, g.N, g.K, g.dtype, g.c_dtype, g.split_k = 128, 128, 256, 1, 1, 4
    g.batch1 = g.batch2 = 1
    g.lda = g.ldb = 256
    g.ldc = 128
    g.A, g.B, g.C = 0x10000, 0x20000, 0x30000     # never dereferenced: validation fails before any launch
    assert lib.fs2_gemm(ctypes.byref(g), None) < 0 and b"split_k" in lib.fs2_last_error()
    g.split_k, g.accumulate, g.c_dtype, g.relu = 1, 1, 0, 1
    assert lib.fs2_gemm(ctypes.byref(g), None) < 0 and b"accumulate" in lib.fs2_last_error()
    # the LDS-strip attention kernels refuse shapes whose 64 x tp strip does not fit 160 KiB or odd head sizes
    assert lib.fs2_attn_probs_lds_bytes(925, 128) > 0 and lib.fs2_attn_probs_lds_bytes(1016, 128) > 0
    assert lib.fs2_attn_probs_lds_bytes(1100, 128) == -1 and lib.fs2_attn_probs_lds_bytes(100, 48) == -1
    rc = lib.fs2_attn_probs_fwd(None, None, 768, 768 * 1100, 128, 128, None, None, None, 0, 1, 2, 1100, 1104, 0.1, 0.0, None, 0,
                                None, None, 0, 0, None)
    assert rc < 0 and b"does not fit" in lib.fs2_last_error()
    # softmax rows longer than the kernel's register tile, column sums with an unaligned width
    assert lib.fs2_softmax_fwd(None, None, 1, None, 1, 1, 3000, 3000, 0, 0.0, None, 0, None) < 0 and b"tp" in lib.fs2_last_error()
    assert lib.fs2_colsum(None, 1, 8, 6, 6, None, None) < 0 and b"multiples of 4" in lib.fs2_last_error()
    assert lib.fs2_colsum_segmented(None, 1, 8, 12, 12, None, 5, 100, None) < 0 and b"seg_cols" in lib.fs2_last_error()
    assert lib.fs2_splitk_finish(None, 8, 6, None, None, 0, 0, 0, None, 1, 6, None) < 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_product_has_no_cpu_fallback():
    from transformer_tts_amd import ops
    with pytest.raises(RuntimeError, match="GPU tensors"):
        ops.cast(torch.zeros(4), torch.float32)


def test_torch_library_registration_and_fake_tensors():
    """the fs2:: custom operators exist in the dispatcher with schemas and meta (fake-tensor) implementations: shape propagation
    works without a GPU (the CUDA implementations are exercised by tests/test_kernels_gpu.py::test_torch_library_ops)"""
    import transformer_tts_amd.torch_ops  # noqa: F401
    x, w, b = torch.empty(3, 7, 16, device="meta"), torch.empty(32, 16, device="meta"), torch.empty(32, device="meta")
    assert torch.ops.fs2.linear(x, w, b, True).shape == (3, 7, 32)
    assert torch.ops.fs2.conv1d_cl(x, torch.empty(8, 16, 5, device="meta"), None, 2, False).shape == (3, 7, 8)
    q = torch.empty(2, 4, 9, 64, device="meta")
    o, st = torch.ops.fs2.flash_attention(q, q, q, torch.empty(2, 9, dtype=torch.bool, device="meta"), True)
    assert o.shape == (2, 4, 9, 64) and st.shape == (2, 4, 9, 2)
    assert "fs2::linear" in str(torch.ops.fs2.linear.default._schema)


def test_ctypes_mirrors_have_the_layout_of_the_header(tmp_path):
    """every ctypes.Structure of ops.py against the struct of the same name in include/fs2_hip.h: a C program (gcc, the header as it
    is) prints sizeof and, per field, offsetof / sizeof; the ctypes mirror must agree field by field (FS2Gemm, FS2WgradPart,
    FS2FlashAttn, FS2CastDesc, FS2QuantDesc, FS2L1Item)"""
    import ctypes
    import subprocess
    from transformer_tts_amd import ops
    structs = [getattr(ops, n) for n in ("FS2Gemm", "FS2WgradPart", "FS2FlashAttn", "FS2CastDesc", "FS2QuantDesc", "FS2L1Item")]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "fs2_hip.h"', "int main(void) {"]
    for S in structs:
        lines.append(f'  printf("{S.__name__} %zu\\n", sizeof({S.__name__}));')
        for f, _ in S._fields_:
            lines.append(f'  printf("{S.__name__}.{f} %zu %zu\\n", offsetof({S.__name__}, {f}), sizeof((({S.__name__}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split(" ", 1) for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for S in structs:
        assert int(got[S.__name__]) == ctypes.sizeof(S), (S.__name__, got[S.__name__], ctypes.sizeof(S))
        for f, t in S._fields_:
            off, size = (int(x) for x in got[f"{S.__name__}.{f}"].split())
            assert (off, size) == (getattr(S, f).offset, ctypes.sizeof(t)), (S.__name__, f, off, size, getattr(S, f).offset)

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
    assert ops.lib().fs2_abi_version() == 1


def test_argument_errors_are_reported_not_crashed():
    """host-side validation runs without a GPU: bad descriptors return FS2_EINVAL with a message"""
    from transformer_tts_amd import ops
    g = ops.FS2Gemm()
    g.M, g.N, g.K, g.dtype = 8, 8, 8, 7
    rc = ops.lib().fs2_gemm(ctypes.byref(g), None)
    assert rc == -1 and b"dtype" in ops.lib().fs2_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_product_has_no_cpu_fallback():
    from transformer_tts_amd import ops
    with pytest.raises(RuntimeError, match="GPU tensors"):
        ops.cast(torch.zeros(4), torch.float32)

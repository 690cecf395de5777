"""pytest fixture: run transformer_tts_amd's host logic on CPU tensors by swapping every function of
``transformer_tts_amd.ops`` for its restatement in ``oracle.primitives`` (test seam only -- the product
never imports the oracle and has no CPU path)."""
import inspect

import pytest


@pytest.fixture
def fake_ops(monkeypatch):
    from oracle import primitives
    from transformer_tts_amd import ops
    for name, fn in inspect.getmembers(primitives, inspect.isfunction):
        if not name.startswith("_") and hasattr(ops, name):
            monkeypatch.setattr(ops, name, fn)
    monkeypatch.setattr(ops, "Rng", primitives.Rng)
    return ops

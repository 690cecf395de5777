"""The committed fixture recipe must keep working from HEAD: regenerate `tiny` (and the cheap `data` / `init` sets) from the
imported reference into a temp dir and compare with the committed files bit for bit.  Needs /root/reference (the build
container); skipped on the GPU box, where the reference never travels."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN

REF = "/root/reference"
SCRIPT = os.path.join(GOLDEN, "make_golden.py")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Models")), reason="reference not present (GPU box)")


def _regen(which, out):
    r = subprocess.run([sys.executable, SCRIPT, which, "--out", str(out)], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(GOLDEN))      # cwd = tests/: the repo-root CLI shims must not matter either way
    assert r.returncode == 0, r.stderr[-2000:]


def _same(a_path, b_path):
    a, b = np.load(a_path), np.load(b_path)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        x, y = a[k], b[k]
        assert x.dtype == y.dtype and x.shape == y.shape, k
        assert np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")), k


@pytest.mark.parametrize("which,fname", [("tiny", "tiny.npz"), ("data", "data.npz"), ("init", "init.npz")])
def test_recipe_reproduces_the_committed_fixture(tmp_path, which, fname):
    _regen(which, tmp_path)
    _same(tmp_path / fname, os.path.join(GOLDEN, fname))


def test_recipe_imports_the_reference_trainer_not_the_repo_shim():
    """regression: a repo-root train_fastspeech2.py (CLI shim) once shadowed the reference module on sys.path"""
    code = ("import sys; sys.argv=['x']; import runpy; ns = runpy.run_path(%r, run_name='recipe'); "
            "T = ns['ref_trainer'](); assert T.__file__.startswith(%r), T.__file__; "
            "assert hasattr(T, 'create_masks') and hasattr(T, 'train_loop')") % (SCRIPT, REF)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(GOLDEN)))   # cwd = repo root, where the shim lives
    assert r.returncode == 0, r.stderr[-2000:]

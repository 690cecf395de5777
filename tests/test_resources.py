"""Register / scratch budgets of the GEMM kernels, checked at build time without a GPU (tools/check_resources.py parses hipcc's
-Rpass-analysis=kernel-resource-usage): every instance of the 16-wave tiled kernel within 128 VGPRs and of the weights-stationary
stream within 256, none with scratch (the round-2 instances with a ReLU mask spilled up to 208 bytes per lane and ran at half speed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_kernels_fit_their_register_budget_without_scratch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_resources.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("ok ")]
    assert len(lines) >= 40, r.stdout      # 2 output types x 3 tile heights x 8 epilogues (minus the three not compiled at 256 rows) + 12 streams

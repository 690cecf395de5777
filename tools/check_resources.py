#!/usr/bin/env python3
"""Build-time register / scratch check of the GEMM kernels ("assert it in a build-time check that parses
-Rpass-analysis=kernel-resource-usage").  Compiles the given sources for gfx950 (device code only, no GPU needed) and applies the
budget of every kernel family:

    fs2_gemm_ring_kernel   <= 128 VGPRs (four waves per SIMD), 0 bytes of scratch -- every instance, bf16 and fp8
    fs2_gemm_ws_kernel     <= 256 VGPRs (two waves per SIMD),  0 bytes of scratch -- every instance
    fs2_gemm_big_km_kernel <= 128 VGPRs, 0 bytes of scratch -- every instance

    python tools/check_resources.py            # prints one line per instance, exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transformer_tts_amd", "csrc")
BUDGET = {            # kernel-name prefix -> (source file, max VGPRs, max scratch bytes per lane)
    "fs2_gemm_ring_kernel": ("gemm_ring.hip", 128, 0),
    "fs2_gemm_ring_kernel ": ("gemm_ring_f8.hip", 128, 0),      # (the one-byte-operand instances: same kernel template, own source file)
    "fs2_gemm_ws_kernel": ("gemm_ws.hip", 256, 0),
    "fs2_gemm_big_km_kernel": ("gemm_big_km.hip", 128, 0),
    "fs2_gemm_big_km_grouped_kernel": ("gemm_big_km.hip", 128, 0),
}


def resources(src):
    """[(mangled name, vgprs, agprs, sgprs, scratch bytes per lane, waves per SIMD)] of every kernel in one source file"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-S",
                            "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "out.s"),
                            os.path.join(CSRC, src)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    out = []
    for blk in re.split(r"remark: .*?Function Name: ", r.stderr)[1:]:
        g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))
        out.append((blk.split()[0], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
    return out


def main():
    bad = 0
    for prefix, (src, max_vgpr, max_scratch) in BUDGET.items():
        rows = [r for r in resources(src) if prefix.strip() in r[0]]
        assert rows, f"no {prefix} instance found in {src}"
        for name, vgpr, agpr, sgpr, scratch, occ in rows:
            ok = vgpr + agpr <= max_vgpr and scratch <= max_scratch
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} {name[:64]:64s} VGPR {vgpr:3d} AGPR {agpr:3d} SGPR {sgpr:3d} scratch {scratch:3d} B/lane  {occ} waves/SIMD")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

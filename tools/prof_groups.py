"""Per-step kernel time by family from the newest rocprofv3 kernel_stats.csv under gpurun_out/prof (12 steps in the trace)."""
import csv
import glob
import sys

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
import os
f = max(glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))


def grp(n):
    if "flash" in n or "attn_strip" in n or "softmax" in n:
        return "attention"
    if "gemm" in n.lower() or "splitk" in n:
        return "GEMM"
    if "ln_" in n or "layernorm" in n:
        return "LayerNorm family"
    if "bn_tanh" in n:
        return "postnet BN+tanh"
    return "other"


groups = {}
for r in rows:
    groups[grp(r["Name"])] = groups.get(grp(r["Name"]), 0.0) + float(r["TotalDurationNs"]) / steps / 1e3
print("kernel us/step", round(sum(groups.values()), 1), {k: round(v, 1) for k, v in groups.items()})
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    us = float(r["TotalDurationNs"]) / steps / 1e3
    if us > 18:
        print(f"{us:8.1f} {int(r['Calls']) / steps:6.1f} x {float(r['AverageNs']) / 1e3:7.1f}  {grp(r['Name']):18s} {r['Name'][:90]}")

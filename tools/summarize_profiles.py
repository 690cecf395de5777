"""Condense the rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/ (run locally
after a gpurun call):  python tools/summarize_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs("profiles", exist_ok=True)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")[:150]


stats = sorted(glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
if stats:
    rows = list(csv.DictReader(open(stats[-1])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    print("wrote", f"profiles/{tag}_kernel_stats.csv", "total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)


def pmc(dirname, counter):
    f = sorted(glob.glob(f"gpurun_out/{dirname}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]
    agg = collections.defaultdict(lambda: [0.0, 0])
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            a = agg[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


fetch, write = pmc("pmcb1", "FETCH_SIZE"), pmc("pmcb2", "WRITE_SIZE")


def variant(name):
    """bench.py's variant key (dtype/A layout/B layout/tile id) of a GEMM kernel name, mangled or demangled"""
    import re
    if "fs2_gemm_big_km_kernel" in name or "fs2_gemm_big_km_grouped_kernel" in name:       # (single and grouped launches: bench.py times both under one key)
        return "bf16/km/km/129"
    if "fs2_gemm_ws_kernel" in name:
        return "bf16/rm/rm/131"
    m = re.search(r"fs2_gemm_ring_kernelI(?:DF16b|f)Li(\d+)ELi\d+ELi(\d)E", name) or re.search(r"fs2_gemm_ring_kernel<[^,]*, (\d+), \d+, (\d)>", name)
    if m:       # (ES = 2: bf16 operands; 1: the fp8 instances, not in the configs[1] step)
        return {"32": "bf16/rm/rm/130", "48": "bf16/rm/rm/192", "64": "bf16/rm/rm/256"}.get(m.group(1)) if m.group(2) == "2" else None
    m = re.search(r"fs2_gemm_big_kernelI(?:DF16b|f)Li(\d+)E", name) or re.search(r"fs2_gemm_big_kernel<[^,]*, (\d+),", name)
    if m:
        return {"32": "bf16/rm/rm/130", "48": "bf16/rm/rm/192", "64": "bf16/rm/rm/256"}.get(m.group(1))
    m = re.search(r"gemm_kernelI(DF16b|f)(?:DF16b|f)Lb([01])ELb([01])ELi(\d+)E", name)
    if m:
        return f"{'bf16' if m.group(1) == 'DF16b' else 'f32'}/{'km' if m.group(2) == '1' else 'rm'}/{'km' if m.group(3) == '1' else 'rm'}/{m.group(4)}"
    m = re.search(r"gemm_kernel<[^,]*, (true|false), (true|false), (\d+),", name)
    if m:       # rocprofv3's demangler drops the element types of this one: bf16 in the timed configuration
        return f"bf16/{'km' if m.group(1) == 'true' else 'rm'}/{'km' if m.group(2) == 'true' else 'rm'}/{m.group(3)}"
    return None


if fetch and write:
    out, agg = {}, collections.defaultdict(lambda: [0.0, 0])
    for k in fetch:
        if "gemm" not in k or k not in write:
            continue
        n = fetch[k][1]
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
        # reads (MI355X_MICROARCH.md, HBM section) -> doubled
        rd = 2.0 * fetch[k][0] * 1024 / n
        wr = write[k][0] * 1024 / write[k][1]
        out[k] = dict(launches=n, hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr, variant=variant(k))
        if variant(k):
            agg[variant(k)][0] += (rd + wr) * n
            agg[variant(k)][1] += n
    fam = collections.defaultdict(lambda: [0.0, 0])           # bench.py's kernel families: dtype/{ring, ws, km, 4wave}
    for k, v in agg.items():
        dt, _, _, tile = k.split("/")
        f = fam[dt + "/" + ("km" if tile == "129" else "ws" if tile == "131" else "ring" if int(tile) >= 130 else "4wave")]
        f[0] += v[0]; f[1] += v[1]
    res = dict(per_kernel=out, by_variant={k: v[0] / v[1] for k, v in agg.items()}, by_family={k: v[0] / v[1] for k, v in fam.items()},
               note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); separate --pmc passes; "
                    "bench.py --no-graph --no-overlap --steps 4 --warmup 8; by_variant = launch-weighted mean over the kernel instances "
                    "(epilogue variants) that bench.py times under one variant key")
    json.dump(res, open("profiles/gemm_traffic.json", "w"), indent=1)
    print("wrote profiles/gemm_traffic.json", {k: round(v / 1e6, 1) for k, v in res["by_variant"].items()}, "MB per launch")

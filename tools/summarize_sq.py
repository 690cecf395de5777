"""Per-kernel SQ / TCC counters of one training step from the `tools/gpu_ci.sh pmcsq` passes (gpurun_out/pmcsq{1,2,3}) ->
profiles/<tag>_sq_pmc.txt.  Run locally after the gpurun call:  python tools/summarize_sq.py r04

Derived columns (MI355X_MICROARCH.md, rocprofv3 PMC slots / cycle constants): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count
quad-cycles of wave lifetime; SQ_VALU_MFMA_BUSY_CYCLES counts cycles of matrix-pipe occupancy summed over the SIMDs;
SQ_BUSY_CU_CYCLES sums, over the CUs, the cycles a CU had a wave.  mfma_busy = MFMA_BUSY / (4 SIMDs x BUSY_CU_CYCLES): the share of
the matrix pipes' time they were occupied while the kernel ran on their CU."""
import collections
import csv
import glob
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ("pmcsq1", "pmcsq2", "pmcsq3"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
want = ("gemm", "flash", "attn_strip", "ln_", "add_ln", "ffn_tail", "bn_tanh", "wgrad_reduce", "adam", "layernorm")
rows = []
for k, d in agg.items():
    if not any(w in k for w in want):
        continue
    g = lambda c: d[c][0] / max(1, d[c][1]) if c in d else float("nan")
    wave = g("SQ_WAVE_CYCLES")
    rows.append((g("SQ_BUSY_CYCLES") * max(1, d["SQ_BUSY_CYCLES"][1]) if "SQ_BUSY_CYCLES" in d else 0.0, k, d, g, wave))
rows.sort(key=lambda r: -r[0])
out = ["kernel | launches | per launch: wave quad-cycles | wait_any % | wait_inst_any % | active_inst_any % | VALU insts | MFMA insts | LDS insts | "
       "mfma_busy % (MFMA_BUSY / 4 / BUSY_CU) | valu_active % of wave cycles | LDS bank-conflict % of LDS-active | L2 hit %"]
for _, k, d, g, wave in rows:
    pct = lambda x: f"{100.0 * x / wave:5.1f}" if wave == wave and wave > 0 else "  n/a"
    busy_cu = g("SQ_BUSY_CU_CYCLES")
    mfma = g("SQ_VALU_MFMA_BUSY_CYCLES")
    lds_a, lds_c = g("SQ_LDS_IDX_ACTIVE"), g("SQ_LDS_BANK_CONFLICT")
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    out.append(f"{k[:110]} | {d['SQ_WAVE_CYCLES'][1] if 'SQ_WAVE_CYCLES' in d else 0} | {wave:.3g} | {pct(g('SQ_WAIT_ANY'))} | {pct(g('SQ_WAIT_INST_ANY'))} | "
               f"{pct(g('SQ_ACTIVE_INST_ANY'))} | {g('SQ_INSTS_VALU'):.3g} | {g('SQ_INSTS_MFMA'):.3g} | {g('SQ_INSTS_LDS'):.3g} | "
               f"{100.0 * mfma / (4.0 * busy_cu) if busy_cu == busy_cu and busy_cu > 0 else float('nan'):5.1f} | {pct(g('SQ_ACTIVE_INST_VALU'))} | "
               f"{100.0 * lds_c / lds_a if lds_a == lds_a and lds_a > 0 else float('nan'):5.1f} | "
               f"{100.0 * hit / (hit + miss) if hit == hit and hit + miss > 0 else float('nan'):5.1f}")
open(f"profiles/{tag}_sq_pmc.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:30]))

#!/usr/bin/env python3
"""Summarise a profiles/collect.sh run:  python profiles/summarize.py gpurun_out/prof_<tag> profiles/<name>

Writes <name>_kernel_stats.csv (copy of rocprofv3's --stats table), <name>_pmc.md (per kernel: mean
duration, PMC counters per launch, derived figures), <name>_pmc_traffic.json and profiles/pmc_traffic.json (HBM bytes
per launch per kernel, which bench.py reports as roofline.traffic).

Refuses ambiguous or stale input: exactly one *_kernel_stats.csv and one counter CSV per pass must be present
(collect.sh starts from an empty directory), and every kernel of the DEFAULT forward — the names
cid_launch_kernel() reports, listed in EXPECT below and checked against the library by tests/test_host.py —
must appear in the stats table and in every counter pass.  `--allow-other-kernels` lifts the second check for
profiles of a non-default configuration (fp16 storage, another conv algorithm).

HBM traffic per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes: FETCH_SIZE/WRITE_SIZE are in KiB and
on gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes for wide coalesced streams
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), hence the doubling of the read side.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


# kernel-name prefixes of the default fp32 forward (conv_algo = winograd42, fused last layer), launch order
EXPECT = ["k_conv_head<", "k_wino42_conv<64, 64, true,", "k_wino42_conv<64, 128, false,", "k_wino42_conv<128, 128, true,",
          "k_wino42_conv<128, 256, false,", "k_wino42_conv<256, 256, false,", "k_gemm_conv<256, 128, 2,",
          "k_wino42_conv<256, 128, false,", "k_wino42_conv<128, 128, false,", "k_convt_s32<128, 64>",
          "k_wino42_conv<128, 64, false,", "k_conv_tail_z<"]


def one(pattern, what):
    found = sorted(glob.glob(pattern, recursive=True))
    if len(found) != 1:
        sys.exit(f"summarize.py: expected exactly one {what}, found {len(found)}: {found} — re-run profiles/collect.sh (it starts from an empty directory)")
    return found[0]


def short(name):
    name = name.replace("void cid::", "").replace("cid::", "")
    return name.split("(")[0]


def load_counters(d):
    """kernel -> counter -> mean value per launch; kernel -> mean duration (ns)"""
    vals, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for f in [one(os.path.join(d, "**", "*_counter_collection.csv"), f"counter CSV under {d}")]:
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    mean = lambda v: sum(v) / len(v)  # noqa: E731
    return {k: {c: mean(v) for c, v in cs.items()} for k, cs in vals.items()}, {k: mean(v) for k, v in dur.items()}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    allow_other = "--allow-other-kernels" in sys.argv
    src, dst = args[0], args[1]
    stats = [one(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), "kernel-stats CSV")]
    shutil.copy(stats[0], dst + "_kernel_stats.csv")
    if os.path.exists(os.path.join(src, "bench_stats.json")):
        shutil.copy(os.path.join(src, "bench_stats.json"), dst + "_bench_under_rocprof.json")
    sq, sq_dur = load_counters(os.path.join(src, "pmc_sq"))
    fe, _ = load_counters(os.path.join(src, "pmc_fetch"))
    wr, _ = load_counters(os.path.join(src, "pmc_write"))
    avg_ns = {}
    if stats:
        for r in csv.DictReader(open(stats[0])):
            avg_ns[short(r["Name"])] = float(r["AverageNs"])
    if not allow_other:
        for table, label in ((avg_ns, "stats table"), (sq, "SQ counter pass"), (fe, "FETCH_SIZE pass"), (wr, "WRITE_SIZE pass")):
            missing = [e for e in EXPECT if not any(k.startswith(e) for k in table)]
            if missing:
                sys.exit(f"summarize.py: the {label} lacks default-forward kernels {missing}: this is not a profile of the default build "
                         f"(kernels found: {sorted(table)}); use --allow-other-kernels for a non-default configuration")
    # ---- ALU accounting (pass pmc_alu; optional for profiles collected before round 4) ----
    # SQ_INSTS_VALU counts every vector-ALU instruction INCLUDING the MFMAs (k_conv_tail_z, which has no MFMA, and the per-layer
    # differences agree with the static count: ~38 plain VALU per 48-MFMA unit + ~510 per item in k_wino42_conv), so the plain vector
    # instructions are the difference.  One of them holds the
    # SIMD's ALU for VALU_ISSUE_CYCLES when it runs beside an fp32 MFMA stream (wave64 on a SIMD-32: 2 passes x 2 cycles = 4 nominal;
    # tools/mix_bench measured 2.8 cycles of matrix-pipe time per instruction in round 1: both are printed).  SQ_VALU_MFMA_BUSY_CYCLES is
    # in cycles summed over the SIMDs; SQ_BUSY_CYCLES x 32 = SIMD cycles (see below).
    alu, alu_lines = {}, []
    if os.path.isdir(os.path.join(src, "pmc_alu")):
        al, _ = load_counters(os.path.join(src, "pmc_alu"))
        alu_lines = ["", "ALU accounting (separate pass): plain VALU = SQ_INSTS_VALU - SQ_INSTS_MFMA; alu_frac = (MFMA busy cycles + plain VALU x c) / SIMD cycles",
                     "", "| kernel | MFMA insts | plain VALU insts | VALU per MFMA | MFMA busy % | alu_frac (c = 4) | alu_frac (c = 2.8) | ACTIVE_INST_VALU % of SIMD cycles | MFMA+VALU co-exec % of MFMA busy | MFMA MOPS F32 / F16 |",
                     "|---|---|---|---|---|---|---|---|---|---|"]
        for k in sorted(avg_ns, key=lambda k: -avg_ns[k]):
            c = al.get(k)
            if not c or not k.startswith("k_"):
                continue
            simd = c.get("SQ_BUSY_CYCLES", float("nan")) * 32.0
            mf, va = c.get("SQ_INSTS_MFMA", 0.0), c.get("SQ_INSTS_VALU", 0.0)
            plain = va - mf
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
            f4, f28 = (busy + 4.0 * plain) / simd, (busy + 2.8 * plain) / simd
            alu[k] = {"mfma_busy_frac": busy / simd, "alu_frac": f4, "alu_frac_c2p8": f28, "valu_per_mfma": plain / mf if mf else None,
                      "insts_mfma": mf, "insts_valu_plain": plain}
            alu_lines.append(f"| `{k}` | {mf:.4g} | {plain:.4g} | {plain / mf if mf else float('nan'):.2f} | {100 * busy / simd:.1f} | {f4:.3f} | {f28:.3f} | "
                             f"{100 * 4 * c.get('SQ_ACTIVE_INST_VALU', float('nan')) / simd:.1f} | {100 * c.get('SQ_VALU_MFMA_COEXEC_CYCLES', float('nan')) / busy if busy else float('nan'):.1f} | "
                             f"{c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0):.3g} / {c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0):.3g} |")
    traffic = {}
    lines = ["| kernel | avg ms (stats pass) | HBM read MB (2xFETCH) | HBM write MB | MFMA busy % of SIMD cycles | WAIT_ANY % | WAIT_INST_ANY % | ACTIVE_INST % | LDS conflict % of LDS active | eff. clock GHz |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(avg_ns, key=lambda k: -avg_ns[k]):
        if not (k.startswith("k_")):
            continue
        c = sq.get(k, {})
        rd = 2 * fe.get(k, {}).get("FETCH_SIZE", float("nan")) * 1024
        wrb = wr.get(k, {}).get("WRITE_SIZE", float("nan")) * 1024
        if rd == rd and wrb == wrb:
            traffic[k] = rd + wrb
        wc = c.get("SQ_WAVE_CYCLES", float("nan"))
        pct = lambda x: 100.0 * c.get(x, float("nan")) / wc if wc else float("nan")  # noqa: E731
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the 1,024 SIMDs' matrix pipes; SQ_BUSY_CYCLES: busy cycles summed over the
        # 32 shader engines (measured: SQ_BUSY_CYCLES / (duration x effective clock) = 31.5-32) -> SIMD cycles = SQ_BUSY_CYCLES x 32
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
        busy = c.get("SQ_BUSY_CYCLES", float("nan")) * 32.0
        gui = fe.get(k, {}).get("GRBM_GUI_ACTIVE", float("nan"))
        clk = gui / 8.0 / avg_ns[k] if gui == gui else float("nan")
        lds = 100.0 * c.get("SQ_LDS_BANK_CONFLICT", float("nan")) / c.get("SQ_LDS_IDX_ACTIVE", float("nan")) if c.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        lines.append(f"| `{k}` | {avg_ns[k] / 1e6:.4f} | {rd / 1e6:.1f} | {wrb / 1e6:.1f} | {100.0 * mfma / busy if busy else float('nan'):.1f} (raw {mfma:.3g}/{busy:.3g}) | "
                     f"{pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | {pct('SQ_ACTIVE_INST_ANY'):.1f} | {lds:.2f} | {clk:.2f} |")
    lines += alu_lines
    open(dst + "_pmc.md", "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(dst + "_pmc_traffic.json", "w"), indent=1)
    if alu:
        json.dump(alu, open(dst + "_pmc_alu.json", "w"), indent=1)
    if not allow_other:   # the files bench.py reads: only ever a profile of the default forward
        json.dump(traffic, open(os.path.join(os.path.dirname(dst) or ".", "pmc_traffic.json"), "w"), indent=1)
        if alu:
            json.dump(alu, open(os.path.join(os.path.dirname(dst) or ".", "pmc_alu.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

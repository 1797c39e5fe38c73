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
    open(dst + "_pmc.md", "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(dst + "_pmc_traffic.json", "w"), indent=1)
    if not allow_other:   # the file bench.py reads: only ever a profile of the default forward
        json.dump(traffic, open(os.path.join(os.path.dirname(dst) or ".", "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise a profiles/collect.sh run:  python profiles/summarize.py gpurun_out/prof_<tag> profiles/<name>

Writes <name>_kernel_stats.csv (copy of rocprofv3's --stats table), <name>_pmc.md (per kernel: mean
duration, PMC counters per launch, derived figures) and profiles/pmc_traffic.json (HBM bytes per launch
per kernel, which bench.py reports as roofline.traffic).

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


def short(name):
    name = name.replace("void cid::", "").replace("cid::", "")
    return name.split("(")[0]


def load_counters(d):
    """kernel -> counter -> mean value per launch; kernel -> mean duration (ns)"""
    vals, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
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
    src, dst = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
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
    traffic = {}
    lines = ["| kernel | avg ms (stats pass) | HBM read MB (2xFETCH) | HBM write MB | MFMA busy % of SQ_BUSY | WAIT_ANY % | WAIT_INST_ANY % | ACTIVE_INST % | LDS conflict % of LDS active | eff. clock GHz |",
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
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs' matrix pipes; SQ_BUSY_CYCLES counts per-SE busy cycles
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
        busy = c.get("SQ_BUSY_CYCLES", float("nan"))
        gui = fe.get(k, {}).get("GRBM_GUI_ACTIVE", float("nan"))
        clk = gui / 8.0 / avg_ns[k] if gui == gui else float("nan")
        lds = 100.0 * c.get("SQ_LDS_BANK_CONFLICT", float("nan")) / c.get("SQ_LDS_IDX_ACTIVE", float("nan")) if c.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        lines.append(f"| `{k}` | {avg_ns[k] / 1e6:.4f} | {rd / 1e6:.1f} | {wrb / 1e6:.1f} | {100.0 * mfma / busy if busy else float('nan'):.1f} (raw {mfma:.3g}/{busy:.3g}) | "
                     f"{pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | {pct('SQ_ACTIVE_INST_ANY'):.1f} | {lds:.2f} | {clk:.2f} |")
    open(dst + "_pmc.md", "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(os.path.dirname(dst) or ".", "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

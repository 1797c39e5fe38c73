"""Summarise gpurun_out/convt_trace.csv (layer_bench <N> trace): per-workgroup phase durations and per-CU-slot gaps."""
import csv, sys, collections, statistics as st
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/convt_trace.csv")))
for r in rows:
    for k in r: r[k] = int(r[k])
t0 = min(r["t_start"] for r in rows)
pro = [r["t_main"] - r["t_start"] for r in rows]
main = [r["t_main_end"] - r["t_main"] for r in rows]
epi = [r["t_stores_issued"] - r["t_main_end"] for r in rows]
drain = [r["t_stores_done"] - r["t_stores_issued"] for r in rows]
tot = [r["t_stores_done"] - r["t_start"] for r in rows]
q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
for name, v in (("prologue", pro), ("main loop", main), ("epilogue (to stores issued)", epi), ("store write-back wait", drain), ("total", tot)):
    print(f"{name:30s} median {st.median(v):9.0f}  p10 {q(v, .1):9.0f}  p90 {q(v, .9):9.0f}  (counter ticks)")
print("span of the launch:", max(r["t_stores_done"] for r in rows) - t0, "ticks;", len(rows), "workgroups")
# per CU (xcc, se, sh?, cu) and wave slot: consecutive workgroups -> gap between end and next start
cu = collections.defaultdict(list)
for r in rows:
    hw = r["hw_id"]
    key = (r["xcc_id"] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)   # xcc, se, sh, cu
    cu[key].append(r)
print("distinct CUs seen:", len(cu), " workgroups per CU:", st.median(len(v) for v in cu.values()))
gaps, conc = [], []
for key, v in cu.items():
    v.sort(key=lambda r: r["t_start"])
    ends = sorted(r["t_stores_done"] for r in v)
    # gap: for each workgroup start after the first two, time since the most recent end before it
    import bisect
    for r in v[2:]:
        k = bisect.bisect_right(ends, r["t_start"]) - 1
        if k >= 0: gaps.append(r["t_start"] - ends[k])
    # busy accounting: fraction of the CU's span with 2 / 1 / 0 workgroups inside their MAIN loop
    ev = []
    for r in v: ev += [(r["t_main"], 1), (r["t_main_end"], -1)]
    ev.sort()
    cur, last, acc = 0, ev[0][0], [0, 0, 0, 0]
    for t, d in ev:
        acc[min(cur, 3)] += t - last; last = t; cur += d
    span = ev[-1][0] - ev[0][0]
    conc.append([a / span for a in acc])
print("start-after-previous-end gap: median", st.median(gaps), " p90", q(gaps, .9))
print("fraction of a CU's time with 0 / 1 / 2 workgroups in the main loop:", [round(st.mean(c[i] for c in conc), 3) for i in range(3)])
if "t_table" in rows[0]:
    seq = ["t_start", "t_table", "t_dma1", "t_dma0", "t_dma_issued", "t_dma_landed", "t_dma_all", "t_main", "t_main_end", "t_ep_b1", "t_ep_b2", "t_ep_b3", "t_stores_issued", "t_stores_done"]
    print("phase-by-phase medians (ticks):")
    for a_, b_ in zip(seq, seq[1:]):
        d = [r[b_] - r[a_] for r in rows if r[a_] and r[b_]]
        print(f"  {a_:16s} -> {b_:16s} median {st.median(d):8.0f}  p90 {q(d, .9):8.0f}")

#!/usr/bin/env python3
"""Same-box A/B of two builds of libcid.so: bench.py in alternation, one process each (box-to-box spread of one build is +-3 %,
within a box +-0.1-0.3 %, so only same-box alternation resolves changes of a percent).
    python profiles/ab_bench.py build_ab/libcid_A.so celebrity_image_denoiser_amd/libcid.so [rounds] [extra bench.py args...]
Prints images/s and per-layer ms of every run and the per-layer mean of B relative to A."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
res = {0: [], 1: []}
for r in range(rounds):
    for k in (0, 1):
        env = dict(os.environ, CID_LIB_PATH=libs[k])
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline", "--steps", "30"] + extra,
                             env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("run failed:", out.stderr[-500:]); sys.exit(1)
        d = json.loads(line[-1])
        res[k].append(d)
        print("AB"[k], r, d["value"], " ".join(f'{l["ms"]:.4f}' for l in d["layers"]), flush=True)
names = [l["layer"] for l in res[0][0]["layers"]]
mean = lambda k, f: sum(f(d) for d in res[k]) / len(res[k])
print("images/s  A %.0f  B %.0f  (B/A %.4f)" % (mean(0, lambda d: d["value"]), mean(1, lambda d: d["value"]), mean(1, lambda d: d["value"]) / mean(0, lambda d: d["value"])))
for i, nm in enumerate(names):
    a, b = mean(0, lambda d: d["layers"][i]["ms"]), mean(1, lambda d: d["layers"][i]["ms"])
    print("%-14s A %.4f  B %.4f  B/A %.4f" % (nm, a, b, b / a))

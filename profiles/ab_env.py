#!/usr/bin/env python3
"""Same-box comparison of several (library, environment) arms: bench.py in alternation, one process each.
    python profiles/ab_env.py ROUNDS name=LIBPATH[,VAR=VALUE...] name2=... [-- extra bench.py args]
Prints images/s and per-layer ms of every run, then per-arm means relative to the first arm."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
extra = []
if "--" in argv:
    i = argv.index("--"); extra = argv[i + 1:]; argv = argv[:i]
rounds = int(argv[0])
arms = []
for spec in argv[1:]:
    name, rest = spec.split("=", 1)
    parts = rest.split(",")
    env = {"CID_LIB_PATH": os.path.abspath(parts[0])}
    for kv in parts[1:]:
        k, v = kv.split("=", 1); env[k] = v
    arms.append((name, env))
res = {name: [] for name, _ in arms}
for r in range(rounds):
    for name, env in arms:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline", "--steps", "30"] + extra,
                             env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "run failed:", out.stderr[-800:]); sys.exit(1)
        d = json.loads(line[-1]); res[name].append(d)
        print(name, r, d["value"], " ".join(f'{l["ms"]:.4f}' for l in d["layers"]), "err", d.get("parity", {}).get("max_abs_err_vs_cpu_oracle"), flush=True)
names = [l["layer"] for l in res[arms[0][0]][0]["layers"]]
mean = lambda k, f: sum(f(d) for d in res[k]) / len(res[k])
base = arms[0][0]
print("images/s  " + "  ".join("%s %.0f (%.4f)" % (k, mean(k, lambda d: d["value"]), mean(k, lambda d: d["value"]) / mean(base, lambda d: d["value"])) for k, _ in arms))
for i, nm in enumerate(names):
    print("%-14s " % nm + "  ".join("%s %.4f (%.4f)" % (k, mean(k, lambda d: d["layers"][i]["ms"]), mean(k, lambda d: d["layers"][i]["ms"]) / mean(base, lambda d: d["layers"][i]["ms"])) for k, _ in arms))

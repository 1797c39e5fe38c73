#!/bin/bash
# Run on the MI355X box from the repo root:  bash profiles/power_samples.sh [extra bench.py args...]  > gpurun_out/power.txt
# Samples rocm-smi (socket power, sclk) every 0.5 s while bench.py runs a long timed region.
python bench.py --steps 500 --warmup 10 --no-extras --no-cpu-baseline "$@" > /tmp/power_bench.json 2> /tmp/power_bench.err &
BP=$!
for i in $(seq 1 36); do
    c=$(rocm-smi -c 2>/dev/null | grep -m1 "sclk" | sed 's/.*sclk clock level: //')
    p=$(rocm-smi -P 2>/dev/null | grep -m1 "Package Power" | sed 's/.*GPU\[[0-9]*\][ \t]*: //')
    echo "$c  : $p"
    kill -0 $BP 2>/dev/null || break
    sleep 0.5
done
wait $BP
python - <<'PY'
import json
d = json.load(open("/tmp/power_bench.json"))
print("# bench:", d["value"], "images/s,", d["ms_per_step"], "ms/step,", d["steps"], "steps, dtype", d["dtype"])
PY

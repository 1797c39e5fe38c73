#!/usr/bin/env python3
"""Summarise `make asm` output (resource_usage.txt): one line per kernel."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "resource_usage.txt").read()
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    print(f"{name:70s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('TotalSGPRs')} spill={r.get('VGPRs Spill')} "
          f"scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")

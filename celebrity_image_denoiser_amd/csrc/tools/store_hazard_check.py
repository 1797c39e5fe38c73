#!/usr/bin/env python3
"""Static check of csrc/cid_kernels.s (`make -C celebrity_image_denoiser_amd/csrc asm`) for the store-data hazard of round 3:

    buffer_store_dwordx4 v[A:A+3], vO, s[..], sN offen      (16 bytes of data, REGISTER soffset)
    <at most WAIT_STATES - 1 other instructions>
    v_* / ds_read* / buffer_load* ... writing a VGPR in A..A+3

hipcc (ROCm 7.2) does not pad this form (LLVM's hazard recognizer exempts MUBUF stores whose soffset is a register from the
">64-bit store data followed by a write of those VGPRs" rule) and gfx950 was observed to store the NEW value
(profiles/r03_store_hazard.txt).  The product's stores of this form go through store16() (wino42_kernels.h), which holds the data
registers across an s_nop 3; this script reports every site where fewer than WAIT_STATES wait states separate such a store from
the first VALU instruction that overwrites one of its data registers.  s_nop N counts N + 1 wait states, every other instruction 1.

    python tools/store_hazard_check.py [cid_kernels.s] [--wait-states 4]      exit status 1 if a site is found
"""
import re
import sys

WAIT = 4
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--wait-states" in sys.argv:
    WAIT = int(sys.argv[sys.argv.index("--wait-states") + 1])
path = args[0] if args else "cid_kernels.s"
store = re.compile(r"^\s*buffer_store_dwordx4\s+v\[(\d+):(\d+)\],\s*v\d+,\s*s\[\d+:\d+\],\s*s\d+\b")
vdst = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))\b")
kernel, sites, nstores = None, [], 0
lines = open(path).read().splitlines()
for i, line in enumerate(lines):
    if line.endswith(":") and (line.startswith("_Z") or line.startswith("k_")):
        kernel = line[:-1]
    m = store.match(line)
    if not m:
        continue
    nstores += 1
    lo, hi = int(m.group(1)), int(m.group(2))
    waited = 0
    for j in range(i + 1, min(i + 40, len(lines))):
        t = lines[j].strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            if t.endswith(":") or t.startswith(".LBB"):
                break                      # control flow joins: stop the linear scan (a branch target is at least one more state away)
            continue
        n = re.match(r"s_nop\s+(\d+)", t)
        if n:
            waited += int(n.group(1)) + 1
            continue
        w = vdst.match(lines[j])
        if w and not w.group(1).startswith("v_cmp") and not w.group(1).startswith("v_readfirstlane"):
            a, b = (int(w.group(3)), int(w.group(4))) if w.group(3) else (int(w.group(5)), int(w.group(5)))
            if a <= hi and b >= lo and waited < WAIT:
                sites.append((kernel, i + 1, line.strip(), j + 1, t, waited))
                break
        waited += 1
        if waited >= WAIT:
            break
print(f"{path}: {nstores} buffer_store_dwordx4 with a register soffset, {len(sites)} closer than {WAIT} wait states to a VALU overwrite of their data")
for k, i, s, j, t, w in sites:
    print(f"  {k}\n    line {i}: {s}\n    line {j}: {t}    ({w} wait state(s) in between)")
sys.exit(1 if sites else 0)

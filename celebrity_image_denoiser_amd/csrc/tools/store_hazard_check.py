#!/usr/bin/env python3
"""Static check of csrc/cid_kernels.s (`make -C celebrity_image_denoiser_amd/csrc asm`) for the store-data hazard of round 3:

    buffer_store_dwordx3 / buffer_store_dwordx4 v[A:B], <vaddr | off>, s[..], sN [offen] [offset:..] [nt] [sc0] [sc1]
                                                        (more than 64 bits of data, REGISTER soffset)
    <at most WAIT_STATES - 1 other instructions>
    an instruction writing a VGPR in A..B: v_* (VALU, v_accvgpr_read included), ds_read* / ds_load*, buffer_load* /
    global_load* / flat_load* / scratch_load* (memory returns are not what LLVM's rule names, but are reported too: a superset)

hipcc (ROCm 7.2) does not pad this form (LLVM's hazard recognizer exempts MUBUF stores whose soffset is a register from the
">64-bit store data followed by a write of those VGPRs" rule) and gfx950 was observed to store the NEW value
(profiles/r03_store_hazard.txt).  The product's stores of this form go through store16() (wino42_kernels.h), which holds the data
registers across an s_nop 3; this script reports every site where fewer than WAIT_STATES wait states separate such a store from
the first instruction that overwrites one of its data registers.  s_nop N counts N + 1 wait states, every other instruction 1.
The scan follows the fall-through path across labels and, at a branch, the taken path as well (to the label in this function).

    python tools/store_hazard_check.py [cid_kernels.s] [--wait-states 4] [--per-kernel]      exit status 1 if a site is found
"""
import collections
import re
import sys

WAIT = 4
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--wait-states" in sys.argv:
    WAIT = int(sys.argv[sys.argv.index("--wait-states") + 1])
    args = [a for a in args if a != str(WAIT)]
path = args[0] if args else "cid_kernels.s"
# data operand, then vaddr (a VGPR, a VGPR pair or `off`), the descriptor, and a REGISTER soffset (a literal 0 / inline constant
# soffset is the form LLVM's hazard table does cover)
store = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(?:v\d+|v\[\d+:\d+\]|off),\s*s\[\d+:\d+\],\s*(s\d+|m0|ttmp\d+)\b(.*)$")
vdst = re.compile(r"^\s*(v_\w+|ds_read\w*|ds_load\w*|buffer_load\w*|global_load\w*|flat_load\w*|scratch_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))\b")
branch = re.compile(r"^\s*s_c?branch\w*\s+(\.?\w+)")
lines = [l.split(";", 1)[0].rstrip() for l in open(path).read().splitlines()]    # comments dropped (labels carry "; @name" / "; %bb")
labels = {l.strip()[:-1]: i for i, l in enumerate(lines) if l.strip().endswith(":")}


def scan(start, lo, hi, waited, depth, seen):
    """First overwrite of v[lo:hi] closer than WAIT wait states after `start`; -> (line index, text, waited) or None."""
    for j in range(start, min(start + 60, len(lines))):
        if waited >= WAIT:
            return None
        t = lines[j].strip()
        if not t or (t.startswith(".") and not t.endswith(":")):
            continue
        if t.endswith(":"):
            continue                       # a label: the fall-through path goes on
        if t.startswith("s_endpgm"):
            return None
        n = re.match(r"s_nop\s+(\d+)", t)
        if n:
            waited += int(n.group(1)) + 1
            continue
        b = branch.match(t)
        if b:
            waited += 1
            tgt = labels.get(b.group(1))
            if tgt is not None and depth < 3 and tgt not in seen:
                hit = scan(tgt + 1, lo, hi, waited, depth + 1, seen | {tgt})
                if hit:
                    return hit
            if t.startswith("s_branch"):
                return None                # unconditional: no fall-through
            continue
        w = vdst.match(lines[j])
        if w and not w.group(1).startswith("v_cmp") and not w.group(1).startswith("v_readfirstlane") and " lds" not in t:
            a, b_ = (int(w.group(3)), int(w.group(4))) if w.group(3) else (int(w.group(5)), int(w.group(5)))
            if a <= hi and b_ >= lo:
                return (j, t, waited)
        waited += 1
    return None


kernel, sites = None, []
per_kernel = collections.OrderedDict()
for i, line in enumerate(lines):
    if line.endswith(":") and (line.startswith("_Z") or line.startswith("k_")):
        kernel = line[:-1]
    m = store.match(line)
    if not m:
        continue
    mods = " ".join(sorted(x for x in m.group(5).split() if x in ("nt", "sc0", "sc1", "offen", "idxen")))
    per_kernel.setdefault(kernel, collections.Counter())[f"dwordx{m.group(1)} {mods}".strip()] += 1
    hit = scan(i + 1, int(m.group(2)), int(m.group(3)), 0, 0, frozenset())
    if hit:
        sites.append((kernel, i + 1, line.strip(), hit[0] + 1, hit[1], hit[2]))
nstores = sum(sum(c.values()) for c in per_kernel.values())
print(f"{path}: {nstores} buffer stores of more than 64 bits with a register soffset in {len(per_kernel)} kernels, "
      f"{len(sites)} closer than {WAIT} wait states to an overwrite of their data")
if "--per-kernel" in sys.argv:
    for k, c in per_kernel.items():
        print(f"  {sum(c.values()):5d}  {k}  ({', '.join(f'{n} x {f}' for f, n in sorted(c.items()))})")
for k, i, s, j, t, w in sites:
    print(f"  {k}\n    line {i}: {s}\n    line {j}: {t}    ({w} wait state(s) in between)")
sys.exit(1 if sites else 0)

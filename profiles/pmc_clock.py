#!/usr/bin/env python3
"""Effective clock and MFMA-busy fraction per kernel from one rocprofv3 counter pass
(--pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES):  python profiles/pmc_clock.py <dir>
Means over the second half of each kernel's launches (the power controller has settled by then)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True)[0]
v = collections.defaultdict(lambda: collections.defaultdict(list)); d = collections.defaultdict(list); seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    v[k][r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); d[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k in v:
    n = len(d[k]); tail = slice(n // 2, n)
    m = lambda c: sum(v[k][c][tail]) / len(v[k][c][tail])
    dur = sum(d[k][tail]) / len(d[k][tail])
    print('%-28s launches %4d  avg %8.1f us  clock %.3f GHz  MFMA busy %.3f of SIMD cycles' % (k, n, dur / 1e3, m('GRBM_GUI_ACTIVE') / 8 / dur, m('SQ_VALU_MFMA_BUSY_CYCLES') / (m('SQ_BUSY_CYCLES') * 32)))

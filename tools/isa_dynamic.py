#!/usr/bin/env python3
"""Dynamic (loop-weighted) instruction counts of a stage kernel by source line bucket:
   python tools/isa_dynamic.py file_g.s kernel [trip=3]
hipcc -S -gline-tables-only output.  Every region closed by a backward branch that contains MFMAs is taken `trip` times
(the residual-block loops); other loops once."""
import collections, re, sys
s = open(sys.argv[1]).read()
kern = sys.argv[2]
trip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
files = dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s))
m = re.search(r'^(_ZN4lyra\d+%s\w*):[^\n]*\n(.*?)^\.Lfunc_end' % kern, s, re.S | re.M)
lines = m.group(2).split('\n')
labels = {}
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l.strip())
    if mm:
        labels[mm.group(1)] = i
weight = [1] * len(lines)
for i, l in enumerate(lines):
    mm = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if mm and labels.get(mm.group(1), 1 << 30) < i:
        lo = labels[mm.group(1)]
        if any('v_mfma' in x for x in lines[lo:i]):
            for j in range(lo, i + 1):
                weight[j] = trip
cur = ("?", 0)
cnt = collections.defaultdict(collections.Counter)
for i, l in enumerate(lines):
    l = l.strip()
    lm = re.match(r'\.loc\s+(\d+)\s+(\d+)', l)
    if lm:
        cur = (files.get(int(lm.group(1)), lm.group(1)).split('/')[-1], int(lm.group(2)))
        continue
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    op = l.split()[0]
    kind = 'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else \
        'vmem' if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) else 'salu' if op.startswith('s_') else 'other'
    cnt[(cur[0], cur[1] // 5 * 5)][kind] += weight[i]
tot = collections.Counter()
rows = sorted(cnt.items(), key=lambda kv: -kv[1]['valu'])
for k, c in rows[:28]:
    print("%-20s %5d  valu %5d lds %4d vmem %4d mfma %4d" % (k[0], k[1], c['valu'], c['lds'], c['vmem'], c['mfma']))
for k, c in cnt.items():
    tot.update(c)
print("total (per wave, dynamic)", dict(tot))

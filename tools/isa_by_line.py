#!/usr/bin/env python3
"""VALU / LDS / VMEM instruction counts of one kernel attributed to source lines (hipcc -S -gline-tables-only):
   python tools/isa_by_line.py /tmp/dec_kernels_g.s dec_s0_kernel [bucket-size]"""
import collections, re, sys
s = open(sys.argv[1]).read()
kern = sys.argv[2]
files = dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s))
files.update(dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"([^"]+)"\s*$', s, re.M)))
m = re.search(r'^(_ZN4lyra\d+%s\w*):[^\n]*\n(.*?)s_endpgm' % kern, s, re.S | re.M)
cnt = collections.defaultdict(collections.Counter)
cur = ("?", 0)
for line in m.group(2).split('\n'):
    line = line.strip()
    lm = re.match(r'\.loc\s+(\d+)\s+(\d+)', line)
    if lm:
        cur = (files.get(int(lm.group(1)), lm.group(1)).split('/')[-1], int(lm.group(2)))
        continue
    if not line or line.startswith(('.', ';', '//')) or line.endswith(':'):
        continue
    op = line.split()[0]
    kind = 'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else \
        'vmem' if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) else 'salu' if op.startswith('s_') else 'other'
    cnt[cur][kind] += 1
tot = collections.Counter()
for k in sorted(cnt):
    c = cnt[k]
    tot.update(c)
    print("%-22s %5d  valu %4d lds %3d vmem %3d mfma %3d salu %3d" % (k[0], k[1], c['valu'], c['lds'], c['vmem'], c['mfma'], c['salu']))
print("total", dict(tot))

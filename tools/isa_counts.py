#!/usr/bin/env python3
"""Static instruction mix of kernels in hipcc -S output:  python tools/isa_counts.py file.s kernel [kernel...]
(straight-line stage kernels: static count x waves = the SQ_INSTS_* counters)."""
import collections, re, sys
s = open(sys.argv[1]).read()
for kern in sys.argv[2:]:
    m = re.search(r'^(_ZN4lyra\d+%s\w*):[^\n]*\n(.*?)^\.Lfunc_end' % kern, s, re.S | re.M)
    c = collections.Counter()
    for line in m.group(2).split('\n'):
        line = line.strip()
        if not line or line.startswith(('.', ';', '//')) or line.endswith(':'):
            continue
        op = line.split()[0]
        k = 'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else \
            'scratch' if op.startswith('scratch_') else 'vmem' if op.startswith(('global_', 'flat_', 'buffer_')) else \
            'waitcnt' if op.startswith('s_waitcnt') else 'barrier' if op.startswith('s_barrier') else 'salu' if op.startswith('s_') else 'other'
        c[k] += 1
        if op.startswith('ds_'):
            c[op] += 1
    print(kern, {k: v for k, v in sorted(c.items()) if not k.startswith('ds_')})
    print("   lds ops:", {k: v for k, v in sorted(c.items()) if k.startswith('ds_')})

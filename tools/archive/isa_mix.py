#!/usr/bin/env python3
"""Static instruction mix per kernel of a hipcc -S dump:  python tools/isa_mix.py /tmp/enc_kernels.s"""
import collections, re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_ZN4lyra\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    c = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line.startswith(('.', ';', '//')) or line.endswith(':'):
            continue
        op = line.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): c[op.split('_')[0] + ('_st' if 'store' in op else '_ld')] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith(('s_cbranch', 's_branch')): c['branch'] += 1
        elif op.startswith('s_nop'): c['nop'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        else: c['other'] += 1
    print(re.sub(r'_ZN4lyra\d+(\w+?)E.*', r'\1', name), dict(sorted(c.items())))

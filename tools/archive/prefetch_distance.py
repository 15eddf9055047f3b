#!/usr/bin/env python3
"""How far ahead of their use are the weight-fragment loads issued?  For every global_load_dwordx4 of a kernel: the number
of MFMA instructions between its issue and the s_waitcnt that retires it (vmcnt is an in-order counter: `vmcnt(N)` retires
all but the newest N outstanding VMEM operations).  Straight-line estimate (loops are unrolled in these kernels).
   llvm-objdump -d kernels.co > k.s ; python tools/prefetch_distance.py k.s enc_s0_kernel [...]"""
import collections, re, sys
s = open(sys.argv[1]).read()
for kern in sys.argv[2:]:
    m = re.search(r'<_ZN4lyra\d+%s\w*>:(.*?)s_endpgm' % kern, s, re.S)
    lines = [l.split('//')[0].strip() for l in m.group(1).split('\n') if l.strip()]
    out = []          # [is_x4, mfma_count_at_issue]
    mf = 0
    dist = []
    for l in lines:
        op = l.split()[0]
        if op.startswith('v_mfma'):
            mf += 1
        elif op.startswith(('global_load', 'global_store', 'buffer_load', 'buffer_store', 'global_atomic', 'flat_')):
            out.append((op == 'global_load_dwordx4', mf))
        elif op == 's_waitcnt':
            v = re.search(r'vmcnt\((\d+)\)', l)
            if v:
                n = int(v.group(1))
                while len(out) > n:
                    x4, at = out.pop(0)
                    if x4:
                        dist.append(mf - at)
        elif op in ('s_barrier',):
            pass
    c = collections.Counter(min(d, 32) for d in dist)
    tot = sum(c.values())
    print(f"{kern}: {tot} weight loads; MFMAs between issue and the wait that retires them: " +
          " ".join(f"{k}:{v}" for k, v in sorted(c.items())) + f"   (share with none: {c.get(0, 0) / max(tot, 1):.0%}; mfma total {mf})")

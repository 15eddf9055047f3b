#!/usr/bin/env python3
"""Print the kernel timeline (start / end / queue) of a few steady-state steps from a rocprofv3 --kernel-trace database:
   python tools/timeline.py <results.db> [n_dispatches]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(db.execute("select name, start, end, queue_id from kernels where name like 'lyra::%' order by start"))
rows = rows[-(n + 20):-20]
t0 = rows[0][1]
busy_until = 0
for name, s, e, q in rows:
    nm = name.split('(')[0].replace('lyra::', '').replace('_kernel', '')
    idle = max(0, s - busy_until) if busy_until else 0
    print(f"{nm:12s} q{q} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f}" + (f"   <- chip idle {idle / 1e3:.1f} us before" if idle > 500 else ""))
    busy_until = max(busy_until, e)

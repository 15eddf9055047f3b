#!/usr/bin/env python3
"""Kernel timeline of the LAST n lyra dispatches of a rocprofv3 --kernel-trace database (the timed region of
`bench.py --steps K --no-kernel-table --latency-steps 0 --no-verify` is the last 7 K dispatches):
   python tools/region_timeline.py <results.db> [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 140
rows = list(db.execute("select name, start, end, queue_id from kernels where name like 'lyra::%' order by start"))
rows = rows[-n:]
t0 = rows[0][1]
busy_until = 0
for name, s, e, q in rows:
    nm = name.split('(')[0].replace('lyra::', '').replace('_kernel', '')
    idle = max(0, s - busy_until) if busy_until else 0
    print(f"{nm:12s} q{q} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f}" + (f"   <- chip idle {idle / 1e3:.1f} us before" if idle > 500 else ""))
    busy_until = max(busy_until, e)
print("span", (max(r[2] for r in rows) - t0) / 1e3, "us")

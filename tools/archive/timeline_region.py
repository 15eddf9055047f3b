#!/usr/bin/env python3
"""Kernel timeline of the LAST n dispatches of a rocprofv3 --kernel-trace database (the timed region of a short bench run):
   python tools/timeline_region.py <results.db> <n_dispatches>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
rows = list(db.execute("select name, start, end, queue_id from kernels where name like 'lyra::%' order by start"))
rows = rows[-n:]
t0 = rows[0][1]
busy_until = 0
idle_total = 0
for name, s, e, q in rows:
    nm = name.split('(')[0].replace('lyra::', '').replace('_kernel', '')
    idle = max(0, s - busy_until) if busy_until else 0
    idle_total += idle
    print(f"{nm:12s} q{q} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f}" + (f"   <- chip idle {idle / 1e3:.1f} us before" if idle > 500 else ""))
    busy_until = max(busy_until, e)
print(f"span {(busy_until - t0) / 1e3:.1f} us, chip idle inside {idle_total / 1e3:.1f} us")
enc0 = [r for r in rows if 'enc_s0' in r[0]]
print("enc_s0 start-to-start (us):", " ".join(f"{(b[1] - a[1]) / 1e3:.0f}" for a, b in zip(enc0, enc0[1:])))

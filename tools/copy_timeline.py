#!/usr/bin/env python3
"""Kernels and memory copies of a rocprofv3 --kernel-trace --memory-copy-trace database on one time line (last n events):
   python tools/copy_timeline.py <results.db> [n] [skip_from_end]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
kt = "kernels" if "kernels" in tabs else None
if kt:
    for name, s, e, q in db.execute(f"select name, start, end, queue_id from {kt}"):
        ev.append((s, e, "K q%s %s" % (q, name.split('(')[0].replace('lyra::', '').replace('_kernel', ''))))
mt = "memory_copies" if "memory_copies" in tabs else None
if mt:
    cols = [r[1] for r in db.execute(f"pragma table_info({mt})")]
    sz = "size" if "size" in cols else None
    nm = "name" if "name" in cols else None
    for row in db.execute(f"select start, end, {nm or 'NULL'}, {sz or 'NULL'} from {mt}"):
        ev.append((row[0], row[1], "C %s %s B" % (row[2], row[3])))
else:
    print("tables:", tabs)
ev.sort()
ev = ev[-(n + skip):len(ev) - skip] if skip else ev[-n:]
t0 = ev[0][0]
busy = 0
for s, e, what in ev:
    gap = (s - busy) / 1e3 if busy else 0
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {what}" + (f"    <- idle {gap:.1f}" if gap > 5 else ""))
    busy = max(busy, e)

#!/usr/bin/env python3
"""Is the two-chain schedule's steady state unique?  Per timed region of a rocprofv3 --kernel-trace run of
tools/region_spread.py: the period (enc_s0 start to enc_s0 start) and where in that period the decoder chain's kernels start.
   cd /tmp && rocprofv3 --kernel-trace -d out -o ps -- python $REPO/tools/region_spread.py 12 300
   python tools/phase_study.py $(find out -name '*.db' | head -1)"""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if t.startswith("kernels")][0]
rows = list(db.execute(f"select name, start, end from {kt} where name like 'lyra::%' order by start"))
def short(n): return n.split('(')[0].replace('lyra::', '').replace('_kernel', '').replace('_xn', '')
ev = [(short(n), s, e) for n, s, e in rows]
e0 = [(s, e) for n, s, e in ev if n == "enc_s0"]
# regions: runs of enc_s0 starts less than 2 ms apart, at least 100 long
regions, cur = [], [e0[0]]
for a, b in zip(e0, e0[1:]):
    if b[0] - a[0] < 2_000_000: cur.append(b)
    else:
        if len(cur) >= 100: regions.append(cur)
        cur = [b]
if len(cur) >= 100: regions.append(cur)
by = {}
for n, s, e in ev: by.setdefault(n, []).append((s, e))
import bisect
for ri, reg in enumerate(regions):
    starts = [s for s, _ in reg]
    mid = starts[len(starts) // 4: -len(starts) // 8]          # steady part
    period = statistics.median(b - a for a, b in zip(mid, mid[1:])) / 1e3
    line = f"region {ri:2d} steps {len(starts):4d} period {period:7.2f} us  span/step {(reg[-1][0] - reg[0][0]) / (len(reg) - 1) / 1e3:7.2f}"
    for k in ("enc_s1", "enc_s2", "rvq_encode", "dec_s0", "dec_s1", "dec_s2"):
        ks = [s for s, _ in by.get(k, [])]
        ph, du = [], []
        for s0 in mid[:-1]:
            i = bisect.bisect_left(ks, s0)
            if i < len(ks) and ks[i] - s0 < period * 1e3:
                ph.append((ks[i] - s0) / 1e3)
                du.append((by[k][i][1] - by[k][i][0]) / 1e3)
        if ph:
            line += f" | {k} +{statistics.median(ph):6.1f} ({statistics.median(du):5.1f})"
    d0 = [e - s for s, e in reg[len(reg) // 4: -len(reg) // 8]]
    line += f" | enc_s0 dur {statistics.median(d0) / 1e3:5.1f}"
    print(line)

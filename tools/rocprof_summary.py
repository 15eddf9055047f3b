#!/usr/bin/env python3
"""Condense rocprofv3 rocpd databases (gpurun_out/prof/*) into the small summaries kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof r01 [out_dir]

Writes profiles/<tag>_kernel_trace_stats.csv (per-kernel calls / total / average / share from the
--kernel-trace --stats run), profiles/<tag>_pmc_hbm.csv (per-kernel FETCH_SIZE / WRITE_SIZE per launch from the
separate --pmc passes) and profiles/traffic.json (HBM bytes per launch per kernel, FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes for wide coalesced reads on gfx950; WRITE_SIZE as reported).
"""
import csv
import json
import os
import sqlite3
import sys


def short(name):
    name = name.split("(")[0]
    return name.replace("lyra::", "").replace("void ", "")[:60]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    os.makedirs(out, exist_ok=True)
    rows = []
    db = os.path.join(src, "trace", f"{tag}_results.db")
    if os.path.exists(db):
        cur = sqlite3.connect(db).cursor()
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            rows.append((short(name), calls, round(total, 1), round(avg, 3), round(pct, 2)))
        with open(os.path.join(out, f"{tag}_kernel_trace_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
            w.writerows(rows)
        for r in rows:
            print(r)
    pmc = {}
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        db = os.path.join(src, sub, f"{tag}_results.db")
        if not os.path.exists(db):
            continue
        cur = sqlite3.connect(db).cursor()
        q = "select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=? group by kernel_name"
        for name, n, val, dur in cur.execute(q, (ctr,)):
            pmc.setdefault(short(name), {})[ctr] = (n, val, dur)
    if pmc:
        traffic = {}
        with open(os.path.join(out, f"{tag}_pmc_hbm.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch",
                        "hbm_bytes_per_launch(2*FETCH+WRITE)", "avg_us_under_pmc"])
            for k, v in sorted(pmc.items()):
                fe = v.get("FETCH_SIZE", (0, 0.0, 0))
                wr = v.get("WRITE_SIZE", (0, 0.0, 0))
                hbm = int((2.0 * fe[1] + wr[1]) * 1024)
                w.writerow([k, fe[0] or wr[0], round(fe[1], 1), round(wr[1], 1), hbm, round((fe[2] or wr[2]) / 1e3, 2)])
                if "kernel" in k and "lyra" not in k and "at::" not in k:
                    traffic[k] = {"hbm_bytes_per_launch": hbm, "fetch_kb": round(fe[1], 1), "write_kb": round(wr[1], 1)}
                print(k, round(fe[1], 1), round(wr[1], 1), hbm)
        traffic["_batch"] = 4096   # streams per launch of the profiled run (bench.py default config)
        json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

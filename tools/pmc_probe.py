#!/usr/bin/env python3
"""Run on the GPU box: collect SQ/GRBM counters for the lyra kernels (short bench run per pass) and print
per-kernel averages per launch.  usage: python tools/pmc_probe.py [pass ...]   (default: all passes)"""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PASSES = {
    "sq1": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
           "SQ_INSTS_VALU SQ_INSTS_MFMA",
    "sq2": "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU "
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM",
    "grbm": "GRBM_GUI_ACTIVE GRBM_COUNT",
}


def main():
    which = sys.argv[1:] or list(PASSES)
    os.chdir("/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    table = {}
    for p in which:
        out = f"/tmp/pmc_{p}"
        cmd = ["rocprofv3", "--pmc"] + PASSES[p].split() + ["--kernel-trace", "-d", out, "-o", "x", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-kernel-table", "--steps", "4", "--warmup", "2", "--latency-steps", "0"]
        subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        db = os.path.join(out, "x_results.db")
        if not os.path.exists(db):
            print("pass", p, "produced no db")
            continue
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, counter_name, avg(value), avg(duration), count(*) from counters_collection "
             "where kernel_name like 'lyra::%' group by kernel_name, counter_name")
        for k, c, v, d, n in cur.execute(q):
            k = k.split("(")[0].replace("lyra::", "")
            table.setdefault(k, {})[c] = v
            table[k]["_us_" + p] = d / 1e3
    for k in sorted(table):
        print(k)
        for c in sorted(table[k]):
            print(f"    {c:28s} {table[k][c]:16.1f}")


if __name__ == "__main__":
    main()

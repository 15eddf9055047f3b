#!/usr/bin/env python3
"""Run bench.py and print a compact summary (for quick A/B on the GPU box)."""
import json, subprocess, sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:],
                     capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-3000:]); sys.exit(1)
r = json.loads(line[-1])
print("frames/s", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"], "serial_sum_us", r.get("serial_sum_us"))
print("  ".join(f"{k.replace('_kernel','')}={v['avg_us']}" for k, v in r["kernels"].items()),
      " sum=", round(sum(v["avg_us"] for v in r["kernels"].values()), 1))

#!/usr/bin/env python3
"""GPU box: spread of the sustained step over regions and processes (is the pipeline's steady state unique?):
python tools/region_spread.py [regions] [steps per region]   -> ms per step of every region of ONE process"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
args = bench.parse(sys.argv[3:] + ["--no-cpu-baseline", "--no-verify", "--latency-steps", "0"])
import torch
torch.cuda.set_device(0)
wl = bench.resolve_workload(args, 1)
sh = bench.Shard(0, 0, wl, args)
kind = "encdec" if wl["mode"] == "encdec" else "generate"
sh.steps(kind, 0, 30); sh.sync()
cur = 30; ms = []
for _ in range(R):
    sh.steps(kind, cur, 64); cur += 64
    secs, _ = sh.timed(kind, cur, K, lambda: None); cur += K
    ms.append(round(secs / K * 1e3, 4))
print(json.dumps({"lib": os.environ.get("LYRA_HIP_LIB") or "default", "K": K, "ms_per_step": ms}))

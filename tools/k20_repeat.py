#!/usr/bin/env python3
"""GPU box: the driver's 20-step form and the sustained form of the bare step, R times each in ONE process
(LYRA_HIP_LIB selects the library build):  python tools/k20_repeat.py [R] [--config N]
Prints the median / min / max ms per step of the R regions of each form."""
import os, sys, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    argv = [a for a in sys.argv[1:] if not a.isdigit()]
    args = bench.parse(argv + ["--no-cpu-baseline", "--no-verify", "--latency-steps", "0", "--steps", "20", "--warmup", "5"])
    import torch
    torch.cuda.set_device(0)
    wl = bench.resolve_workload(args, 1)
    sh = bench.Shard(0, 0, wl, args)
    kind = "encdec" if wl["mode"] == "encdec" else "generate"
    sh.steps(kind, 0, 30); sh.sync()
    cur = 30
    out = {"lib": os.environ.get("LYRA_HIP_LIB") or "default", "B": wl["B"]}
    for K, name in ((20, "k20"), (400, "k400")):
        ms = []
        for _ in range(R if K == 20 else max(3, R // 3)):
            sh.steps(kind, cur, 64); cur += 64
            secs, _ = sh.timed(kind, cur, K, lambda: None); cur += K
            ms.append(secs / K * 1e3)
        out[name] = {"median_ms": round(statistics.median(ms), 4), "min_ms": round(min(ms), 4), "max_ms": round(max(ms), 4),
                     "Mframes_s_median": round(wl["B"] / statistics.median(ms) / 1e3, 3)}
    print(json.dumps(out))

main()

#!/usr/bin/env python3
"""GPU box: host <-> device copy time of one hop of 4096 streams (2.6 MB of PCM) from pageable vs pinned host memory."""
import time, torch
dev = torch.device("cuda", 0)
n = 4096 * 320
d = torch.zeros(n, dtype=torch.int16, device=dev)
for name, h in (("pageable", torch.zeros(n, dtype=torch.int16)), ("pinned", torch.zeros(n, dtype=torch.int16).pin_memory())):
    for direction in ("H2D", "D2H"):
        for _ in range(5):
            (d.copy_(h, non_blocking=True) if direction == "H2D" else h.copy_(d, non_blocking=True)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            (d.copy_(h, non_blocking=True) if direction == "H2D" else h.copy_(d, non_blocking=True)); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f"{name:9s} {direction}: {dt * 1e6:7.1f} us per 2.6 MB copy = {n * 2 / dt / 1e9:5.1f} GB/s")

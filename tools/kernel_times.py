#!/usr/bin/env python3
"""GPU box: steady-state per-kernel times (HIP events around every launch, 10 warm-up + N timed steps).
For A/B work: LYRA_HIP_LIB=<variant.so> python tools/kernel_times.py [steps]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import lyra_amd
B, bits = int(os.environ.get("B", 4096)), 184
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
ctx = lyra_amd.LyraHip(max_streams=B)
g = torch.Generator(device=dev); g.manual_seed(1)
pcm = torch.randint(-32768, 32768, (10 + N, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
ids = torch.arange(B, device=dev, dtype=torch.int32)
pks = [torch.empty((B, 23), device=dev, dtype=torch.uint8) for _ in range(2)]
outs = [torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
SERIAL = os.environ.get("SERIAL") == "1"


def step(i):
    ctx.encode_dev(ids, pcm[i], bits, pks[i & 1])
    ctx.decode_dev(ids, pks[i & 1], bits, outs[i & 1])
    if SERIAL:
        ctx.synchronize()
torch.cuda.synchronize()
import time
for i in range(10):
    step(i)
ctx.synchronize()
ctx.profile_enable(True); ctx.profile_read()
t0 = time.perf_counter()
for i in range(10, 10 + N):
    step(i)
p = ctx.profile_read()
wall = (time.perf_counter() - t0) / N * 1e6
tot = 0.0
s = []
for k, (ms, n) in p.items():
    if n:
        s.append(f"{k.replace('_kernel','')}={ms / n * 1e3:.1f}"); tot += ms / n * 1e3
print("  ".join(s), f" sum={tot:.1f}  wall/step={wall:.1f}us (events on)")
ctx.profile_enable(False)
t0 = time.perf_counter()
for i in range(10 + N - 30 if N >= 30 else 10, 10 + N):
    step(i)
ctx.synchronize()
n2 = min(N, 30)
print(f"wall/step without events = {(time.perf_counter() - t0) / n2 * 1e6:.1f} us")

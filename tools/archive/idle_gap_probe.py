#!/usr/bin/env python3
"""GPU box: does the idle gap before a short timed region matter (clock / power state)?  K = 20 steps timed right after
300 busy steps, and after 2 ms / 20 ms / 1 s of idling."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import lyra_amd
B, bits, K = 4096, 184, 20
ctx = lyra_amd.LyraHip(max_streams=B)
ctx.torch_order = False
dev = torch.device("cuda", 0)
pcm = torch.randint(-32768, 32768, (32, B, 320), device=dev, dtype=torch.int32).to(torch.int16)
ids = torch.arange(B, device=dev, dtype=torch.int32)
pk = [torch.zeros((B, 23), device=dev, dtype=torch.uint8) for _ in range(2)]
out = [torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
step = 0
def run(n):
    global step
    ctx.run_steps_dev(ids, bits, n, first_step=step, d_pcm_ring=pcm, d_packets=pk, d_pcm_out=out)
    step += n
for gap in (0.0, 0.002, 0.02, 1.0, 0.0, 0.02):
    res = []
    for rep in range(3):
        run(300); ctx.synchronize()
        time.sleep(gap)
        t0 = time.perf_counter()
        run(K); ctx.synchronize()
        res.append((time.perf_counter() - t0) / K * 1e6)
    print(f"idle {gap * 1e3:7.1f} ms before the region: {' / '.join(f'{r:.1f}' for r in res)} us per step", flush=True)
run(1000); ctx.synchronize()
t0 = time.perf_counter(); run(1000); ctx.synchronize()
print(f"sustained: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us per step")

#!/usr/bin/env python3
"""GPU box: do kernels launched from different HIP streams overlap?  Two contexts (independent stream pairs),
2048 streams each, vs one context with 4096."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lyra_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
N = 40


def run(ctxs, B):
    pcm = torch.randint(-32768, 32768, (N, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    pk = [[torch.empty((B, 23), device=dev, dtype=torch.uint8) for _ in range(2)] for _ in ctxs]
    out = [[torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)] for _ in ctxs]
    torch.cuda.synchronize()
    def step(i):
        for c, ctx in enumerate(ctxs):
            ctx.encode_dev(ids, pcm[i], 184, pk[c][i & 1])
            ctx.decode_dev(ids, pk[c][i & 1], 184, out[c][i & 1])
    for i in range(8):
        step(i)
    for ctx in ctxs:
        ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(8, N):
        step(i)
    for ctx in ctxs:
        ctx.synchronize()
    return (time.perf_counter() - t0) / (N - 8) * 1e6


one = lyra_amd.LyraHip(max_streams=4096)
print("1 ctx  x 4096 streams: %.1f us/step" % run([one], 4096))
print("1 ctx  x 2048 streams: %.1f us/step" % run([one], 2048))
two = [lyra_amd.LyraHip(max_streams=2048), lyra_amd.LyraHip(max_streams=2048)]
print("2 ctxs x 2048 streams: %.1f us/step (4096 total)" % run(two, 2048))
four = two + [lyra_amd.LyraHip(max_streams=1024), lyra_amd.LyraHip(max_streams=1024)]
print("4 ctxs x 1024 streams: %.1f us/step (4096 total)" % run(four, 1024))
print("1 ctx  x 1024 streams: %.1f us/step" % run([one], 1024))

#!/usr/bin/env python3
"""GPU box: per-kernel time as a function of batch size (looks for residency cliffs)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import lyra_amd
dev = torch.device("cuda", 0)
ctx = lyra_amd.LyraHip(max_streams=8192)
g = torch.Generator(device=dev); g.manual_seed(1)
for B in (1024, 2048, 3072, 4096, 5120, 6144, 8192):
    pcm = torch.randint(-32768, 32768, (16, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    pk = torch.empty((B, 23), device=dev, dtype=torch.uint8)
    out = torch.empty((B, 320), device=dev, dtype=torch.int16)
    torch.cuda.synchronize()
    for i in range(4):
        ctx.encode_dev(ids, pcm[i], 184, pk); ctx.decode_dev(ids, pk, 184, out); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_read()
    for i in range(4, 16):
        ctx.encode_dev(ids, pcm[i], 184, pk); ctx.decode_dev(ids, pk, 184, out); ctx.synchronize()
    p = ctx.profile_read(); ctx.profile_enable(False)
    print(B, "  ".join(f"{k.replace('_kernel','')}={ms / n * 1e3:.0f}" for k, (ms, n) in p.items() if n))

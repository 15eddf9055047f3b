#!/usr/bin/env python3
"""GPU box: rvq_encode kernel time, back to back (codebook warm in L2) vs with a cache-flushing kernel in between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lyra_amd
dev = torch.device("cuda", 0)
for B in (16, 1024, 4096):
    c = lyra_amd.LyraHip(max_streams=max(B, 2048))   # unmasked streams
    c.torch_order = False
    feat = torch.randn(B, 64, device=dev)
    idx = torch.empty(B, 46, device=dev, dtype=torch.int32)
    junk = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
    for bits in (64, 184):
        for flush in (False, True):
            ts = []
            for i in range(30):
                if flush:
                    junk.add_(1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                c.rvq_encode_dev(feat, bits, idx)
                c.synchronize()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            print(f"B={B:5d} bits={bits:3d} flush={flush!s:5s} host-timed launch+kernel+sync: median {ts[len(ts)//2]*1e6:6.1f} us  min {ts[0]*1e6:6.1f} us")

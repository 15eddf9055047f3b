#!/usr/bin/env python3
"""Timing variant: for single launches of enc_s2, compare the HIP-event duration with the span first-WG-start ->
last-WG-end recorded inside the kernel (100 MHz wall clock).  MODE=extract|encode|full"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", "timing.so")
import torch
import lyra_amd
B = int(os.environ.get("B", 4096))
MODE = os.environ.get("MODE", "full")
dev = torch.device("cuda", 0)
ctx = lyra_amd.LyraHip(max_streams=B)
g = torch.Generator(device=dev); g.manual_seed(1)
pcm = torch.randint(-32768, 32768, (16, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
ids = torch.arange(B, device=dev, dtype=torch.int32)
pk = torch.empty((B, 23), device=dev, dtype=torch.uint8)
out = torch.empty((B, 320), device=dev, dtype=torch.int16)
feats = torch.empty((B, 64), device=dev, dtype=torch.float32)
torch.cuda.synchronize()
def step(i):
    if MODE == "extract":
        ctx.extract_dev(ids, pcm[i], feats)
    else:
        ctx.encode_dev(ids, pcm[i], 184, pk)
        if MODE == "full":
            ctx.decode_dev(ids, pk, 184, out)
    ctx.synchronize()
for i in range(6):
    step(i)
ctx.profile_enable(True); ctx.profile_read()
buf = (ctypes.c_longlong * (2048 * 4))()
for it in range(6, 12):
    step(it)
    p = ctx.profile_read()
    ctx.L.lyra_hip_debug_wgtrace_s2(buf)
    t = np.array(buf[:]).reshape(2048, 4)
    t = t[t[:, 0] != 0]
    span = (t[:, 1].max() - t[:, 0].min()) / 100.0
    dur = (t[:, 1] - t[:, 0]) / 100.0
    ev = p["enc_s2_kernel"][0] / max(p["enc_s2_kernel"][1], 1) * 1e3
    print(f"{MODE} B={B} event {ev:6.1f} us   span {span:6.1f} us   WG dur median {np.median(dur):5.1f} max {dur.max():5.1f}   start spread {(t[:,0].max()-t[:,0].min())/100.0:.1f}")

#!/usr/bin/env python3
"""Per-workgroup timeline of enc_s2 / dec_s0 inside the sustained encode+decode pipeline (timing variant),
next to the HIP-event duration of the same kernels: does the kernel time come from the workgroups' own duration,
from late starts, or from a tail?   SERIAL=1 forces the two library streams in call order."""
import ctypes, os, sys
from collections import Counter
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", os.environ.get("VARIANT", "timing") + ".so")
import torch
import lyra_amd
B = int(os.environ.get("B", 4096))
dev = torch.device("cuda", 0)
ctx = lyra_amd.LyraHip(max_streams=B)
ctx.torch_order = False
g = torch.Generator(device=dev); g.manual_seed(1)
pcm = torch.randint(-32768, 32768, (8, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
ids = torch.arange(B, device=dev, dtype=torch.int32)
pk = [torch.empty((B, 23), device=dev, dtype=torch.uint8) for _ in range(2)]
out = [torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
torch.cuda.synchronize()
if os.environ.get("SERIAL") == "1":
    ctx.set_serial(True)
N = int(os.environ.get("STEPS", 40))
for i in range(N):
    if i == N - 10:
        ctx.synchronize(); ctx.profile_enable(True); ctx.profile_read()
    ctx.encode_dev(ids, pcm[i % 8], 184, pk[i & 1])
    ctx.decode_dev(ids, pk[i & 1], 184, out[i & 1])
p = ctx.profile_read()
print("event times:", "  ".join(f"{k.replace('_kernel','')}={ms / n * 1e3:.1f}" for k, (ms, n) in p.items() if n))
for kern in ("s2", "d0"):
    buf = (ctypes.c_longlong * (2048 * 4))()
    getattr(ctx.L, "lyra_hip_debug_wgtrace_" + kern)(buf)
    t = np.array(buf[:]).reshape(2048, 4)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0
    end = (t[:, 1] - t0) / 100.0
    dur = end - start
    hw = t[:, 2]
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = t[:, 3] & 15
    key = [(int(x), int(e), int(h), int(c)) for x, e, h, c in zip(xcc, se, sh, cu)]
    cnt = Counter(key)
    print(f"[{kern}] {len(t)} WGs on {len(cnt)} CUs; WGs/CU histogram {sorted(Counter(cnt.values()).items())}; "
          f"span {end.max():.1f} us; start median {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}; "
          f"dur min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}")
    hist = np.histogram(start, bins=[0, 1, 2, 5, 10, 20, 30, 40, 60, 100])[0]
    print(f"     start histogram (0,1,2,5,10,20,30,40,60,100 us): {hist.tolist()}")
    hist = np.histogram(end, bins=[0, 20, 30, 40, 50, 60, 70, 80, 100, 150])[0]
    print(f"     end histogram (0,20,30,40,50,60,70,80,100,150 us): {hist.tolist()}")

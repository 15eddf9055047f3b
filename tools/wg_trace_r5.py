#!/usr/bin/env python3
"""GPU box, timing variant: per-workgroup start / duration / placement of one stage kernel run ALONE at B streams.
KERNEL=s0|s2 (enc_s0 / enc_s2), d0 ... via lyra_hip_debug_wgtrace_<KERNEL>; MODE=extract|full."""
import ctypes, os, sys
from collections import Counter
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", "timing.so")
import lyra_amd
B = int(os.environ.get("B", 4096))
K = os.environ.get("KERNEL", "s0")
ctx = lyra_amd.LyraHip(max_streams=B)
pcm = np.random.default_rng(0).integers(-32768, 32768, size=(B, 320)).astype(np.int16)
for _ in range(3):
    pk = ctx.encode(pcm, 184)
    if K.startswith("d"):
        ctx.decode(pk, 184)
buf = (ctypes.c_longlong * (2048 * 4))()
getattr(ctx.L, "lyra_hip_debug_wgtrace_" + K)(buf)
t = np.array(buf[:]).reshape(2048, 4)
t = t[t[:, 0] != 0]
n = len(t)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0
end = (t[:, 1] - t0) / 100.0
dur = end - start
hw = t[:, 2]
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = t[:, 3] & 15
key = [(int(x), int(e), int(h), int(c)) for x, e, h, c in zip(xcc, se, sh, cu)]
cnt = Counter(key)
print(f"KERNEL={K} B={B}: {n} workgroups on {len(cnt)} distinct CUs; WGs/CU histogram: {sorted(Counter(cnt.values()).items())}")
print(f"span first start -> last end {end.max():.1f} us; start: p50 {np.median(start):.1f} p90 {np.percentile(start,90):.1f} max {start.max():.1f} us")
print(f"duration: min {dur.min():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f} us;  end: p10 {np.percentile(end,10):.1f} p50 {np.median(end):.1f} p90 {np.percentile(end,90):.1f}")
hist, edges = np.histogram(start, bins=12)
print("start histogram:", [(round(float(edges[i]), 1), int(hist[i])) for i in range(len(hist))])
hist, edges = np.histogram(end, bins=12)
print("end histogram:", [(round(float(edges[i]), 1), int(hist[i])) for i in range(len(hist))])

#!/usr/bin/env python3
"""Per-workgroup placement/timeline of one kernel (timing variant): which XCD/SE/CU each workgroup ran on,
when it started and how long it took.  KERNEL=s2 (enc_s2)."""
import ctypes, os, sys
from collections import Counter
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", "timing.so")
import lyra_amd
B = int(os.environ.get("B", 4096))
ctx = lyra_amd.LyraHip(max_streams=B)
pcm = np.random.default_rng(0).integers(-32768, 32768, size=(B, 320)).astype(np.int16)
for _ in range(3):
    ctx.extract(pcm)
buf = (ctypes.c_longlong * (2048 * 4))()
getattr(ctx.L, "lyra_hip_debug_wgtrace_" + os.environ.get("KERNEL", "s2"))(buf)
t = np.array(buf[:]).reshape(2048, 4)
t = t[t[:, 0] != 0]
n = len(t)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0
dur = (t[:, 1] - t[:, 0]) / 100.0
hw = t[:, 2]
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = t[:, 3] & 15
key = [(int(x), int(e), int(h), int(c)) for x, e, h, c in zip(xcc, se, sh, cu)]
cnt = Counter(key)
print(f"{n} workgroups on {len(cnt)} distinct CUs; WGs/CU histogram: {sorted(Counter(cnt.values()).items())}")
print(f"kernel span {((t[:,1]-t0).max())/100.0:.1f} us; start: median {np.median(start):.1f} max {start.max():.1f} us")
print(f"duration: min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f} us")
per = np.array([cnt[k] for k in key])
for c in sorted(set(per)):
    print(f"  WGs on CUs hosting {c}: n={int((per==c).sum())} median dur {np.median(dur[per==c]):.1f} us, median start {np.median(start[per==c]):.1f}")
print("per-XCD WG count:", sorted(Counter(int(x) for x in xcc).items()))
late = start > 5.0
print(f"workgroups starting >5us after the first: {int(late.sum())}")
for x in sorted(set(int(v) for v in xcc)):
    d = dur[xcc == x]
    print(f"  XCD {x}: dur min {d.min():.1f} median {np.median(d):.1f} max {d.max():.1f}")
order = np.argsort(-dur)[:12]
print("slowest WGs (index, xcc, se, cu, dur):", [(int(i), int(xcc[i]), int(se[i]), int(cu[i]), round(float(dur[i]), 1)) for i in order])

#!/usr/bin/env python3
"""Phase timing of workgroup 0 of logmel_kernel with the noise tail (timing variant: make EXTRA=-DLYRA_TIMING ...)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", "timing.so")
import lyra_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = lyra_amd.LyraHip(max_streams=B)
rng = np.random.default_rng(0)
for _ in range(4):
    pcm = (rng.normal(size=(B, 320)) * 3000).astype(np.int16)
    ctx.noise_receive(pcm)
buf = (ctypes.c_longlong * 128)()
ctx.L.lyra_hip_debug_timing_misc(buf)
t = np.array(buf[:])
names = {111: "window load + stage", 112: "5 radix-4 passes", 113: "separate + |X| (sqrt)", 114: "park mel weights",
         115: "band sums + log (thread 0)", 120: "... until every wave is through", 116: "tail: loads + ballot", 117: "tail: Average()", 118: "tail: recurrence", 119: "tail: header"}
print("logmel_noise WG0 total cycles", t[119] - t[110])
prev = 110
for i in (111, 112, 113, 114, 115, 120, 116, 117, 118, 119):
    if t[i]:
        print(f"  {names[i]:26s} {t[i] - t[prev]:7d}")
        prev = i

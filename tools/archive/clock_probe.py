#!/usr/bin/env python3
"""GPU box, timing variant: effective shader clock (s_memtime cycles per 100 MHz tick) in the phases of dec_s1 at
B = 4096 -- is an MFMA-dense phase slower because the clock drops?"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", "timing.so")
import lyra_amd
B = int(os.environ.get("B", 4096))
ctx = lyra_amd.LyraHip(max_streams=B)
feats = np.random.default_rng(0).normal(size=(B, 64)).astype(np.float32) * 3
for _ in range(4):
    ctx.generate(feats)
buf = (ctypes.c_longlong * 128)()
ctx.L.lyra_hip_debug_timing_d0(buf)
t = np.array(buf[:])
names = {51: "prologue end", 52: "resblocks end", 53: "lrelu end", 54: "tconv pass 1 end", 55: "tconv pass 2 end"}
prev = 51
for k in (52, 53, 54, 55):
    cyc = t[k] - t[prev]; wall = (t[100 + 10 + (k - 50)] - t[100 + 10 + (prev - 50)]) / 100.0
    print(f"{names[prev]:18s} -> {names[k]:18s}: {cyc:8d} cycles  {wall:7.2f} us  -> {cyc / max(wall, 1e-9) / 1e3:.2f} GHz")
    prev = k

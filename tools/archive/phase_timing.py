#!/usr/bin/env python3
"""GPU box: run the -DLYRA_TIMING build and print enc_s0 phase durations (cycles) of workgroup 0, wave 0."""
import ctypes, os, sys
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
buf = (ctypes.c_longlong * 128)()
ctx.L.lyra_hip_debug_timing.argtypes = [ctypes.c_void_p]
ctx.L.lyra_hip_debug_timing(buf)
t = np.array(buf[:])
names = {0: "start", 1: "pcm staged+state", 2: "first conv", 3: "resblocks done", 4: "lrelu+halo", 5: "k10s5 gemm", 6: "end"}
print("total", t[6] - t[0], "cycles;", (t[101] - t[100]) / 100.0, "us on the 100 MHz clock ->", (t[6] - t[0]) / max(t[101] - t[100], 1) / 10.0, "GHz effective")
for i in range(1, 7):
    print(f"  {names[i]:20s} {t[i] - t[i-1]:8d}")
ph = ["->top", "a write+bar", "dw", "bar+state wr+bar", "D write+bar", "pw gemm", "bar+P write+bar", "cv gemm+resid"]
for r in range(3):
    base = 10 + r * 8
    prev = t[2] if r == 0 else t[10 + (r - 1) * 8 + 7]
    row = []
    for k in range(8):
        row.append(f"{ph[k]}={t[base + k] - prev}")
        prev = t[base + k]
    print(f"  res{r}:", "  ".join(row))

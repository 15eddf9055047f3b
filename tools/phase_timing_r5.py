#!/usr/bin/env python3
"""GPU box: -DLYRA_TIMING build (lyra_amd/variants/timing.so): phase durations (shader cycles) of workgroup 0, wave 0 of
enc_s0 and enc_s1 at B streams (all tiles co-resident), MFMA cycles of each GEMM phase beside them."""
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
print("B", B, "enc_s0 total", t[6] - t[0], "cycles (880 MFMAs per wave = 28160 cycles of matrix pipe per wave, x tiles per CU);",
      (t[101] - t[100]) / 100.0, "us wall ->", (t[6] - t[0]) / max(t[101] - t[100], 1) / 10.0, "GHz")
for i in range(1, 7):
    print(f"  {names[i]:20s} {t[i] - t[i-1]:8d}")
ph = ["->top", "a write+bar", "dw", "bar+state wr+bar", "D write+bar", "pw gemm(80 mfma=2560)", "bar+P write+bar", "cv gemm+resid(2560)"]
for r in range(3):
    base = 10 + r * 8
    prev = t[2] if r == 0 else t[10 + (r - 1) * 8 + 7]
    row = []
    for k in range(8):
        row.append(f"{ph[k]}={t[base + k] - prev}")
        prev = t[base + k]
    print(f"  res{r}:", "  ".join(row))
print("enc_s1 total", t[73] - t[70], " load", t[71] - t[70], " resblocks", t[72] - t[71], " lrelu+down conv", t[73] - t[72])
ph = {2: "dw+state+bar", 3: "pw gemm(64 mfma=2048)", 4: "bar+P write+bar", 5: "hist prefetch+cv gemm(32 mfma=1024)", 6: "X update+bar"}
for r in range(3):
    base = 40 + r * 8
    print(f"  res{r}:", "  ".join(f"{ph[k]}={t[base+k]-t[base+(0 if k == 2 else k-1)]}" for k in range(2, 7)))
wall = (t[103] - t[102]) / 100.0
print(f"  WG0 wall {wall:.1f} us -> {(t[73]-t[70])/max(wall,1e-9)/1e3:.2f} GHz")

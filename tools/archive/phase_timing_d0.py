#!/usr/bin/env python3
import ctypes, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", "timing.so")
import lyra_amd
ctx = lyra_amd.LyraHip(max_streams=4096)
feats = np.random.default_rng(0).normal(size=(4096, 64)).astype(np.float32) * 3
for _ in range(3):
    ctx.generate(feats)
buf = (ctypes.c_longlong * 128)()
ctx.L.lyra_hip_debug_timing_d0(buf)
t = np.array(buf[:])
names = ["", "ids+luts+feat window", "head conv+q", "up0 tconv", "a0 quantize", "resblock0", "resblock1+2", "lrelu", "up1 tconv+out"]
print("dec_s0 total", t[88] - t[80])
for i in range(81, 89):
    print(f"  {names[i-80]:22s} {t[i] - t[i-1]:7d}")
for base in (90, 94):
    print("  resblock@", base, " lrelu+dw+state", t[base+1]-t[base], " pw gemm+epi", t[base+2]-t[base+1], " cv gemm+add", t[base+3]-t[base+2])
print("  resblock@94 split: pw phase = gemm (LDS A + L2 weight fragments + 8 MFMA)", t[122]-t[95], "+ epilogue", t[96]-t[122], "; cv phase = gemm", t[123]-t[96], "+ epilogue (requant + ADD tables)", t[97]-t[123])
wall = (t[121] - t[120]) / 100.0
print(f"  WG0 wall {wall:.1f} us  -> shader clock {(t[88]-t[80])/wall/1e3:.2f} GHz")
idx = [i for i in range(128) if t[i] != 0 and i < 100]
idx.sort(key=lambda i: t[i])
print("  stamps in time order (id:+delta):", " ".join(f"{i}:+{t[i]-t[idx[max(0,k-1)]]}" for k, i in enumerate(idx)))

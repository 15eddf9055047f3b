#!/usr/bin/env python3
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
ctx.L.lyra_hip_debug_timing(buf)
t = np.array(buf[:])
print("enc_s1 total", t[73] - t[70], " load", t[71] - t[70], " resblocks", t[72] - t[71], " lrelu+down conv", t[73] - t[72])
ph = {2: "dw+state+bar", 3: "pw gemm", 4: "bar+P write+bar", 5: "hist prefetch+cv gemm", 6: "X update+bar"}
for r in range(3):
    base = 40 + r * 8
    print(f"  res{r}:", "  ".join(f"{ph[k]}={t[base+k]-t[base+(0 if k == 2 else k-1)]}" for k in range(2, 7)))
wall = (t[103] - t[102]) / 100.0
print(f"  WG0 wall {wall:.1f} us -> {(t[73]-t[70])/wall/1e3:.2f} GHz")

#!/usr/bin/env python3
import ctypes, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["LYRA_HIP_LIB"] = os.path.join(ROOT, "lyra_amd", "variants", "timing.so")
import lyra_amd
ctx = lyra_amd.LyraHip(max_streams=4096)
pcm = np.random.default_rng(0).integers(-32768, 32768, size=(4096, 320)).astype(np.int16)
if os.environ.get("MODE") == "full":   # sustained encode+decode pipeline, as benchmarked
    import torch
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    ids = torch.arange(4096, device=dev, dtype=torch.int32)
    pk = [torch.empty((4096, 23), device=dev, dtype=torch.uint8) for _ in range(2)]
    out = [torch.empty((4096, 320), device=dev, dtype=torch.int16) for _ in range(2)]
    for i in range(int(os.environ.get("STEPS", 200))):
        ctx.encode_dev(ids, d_pcm, 184, pk[i & 1])
        ctx.decode_dev(ids, pk[i & 1], 184, out[i & 1])
        if os.environ.get("SERIAL") == "1":
            ctx.synchronize()
    ctx.synchronize()
else:
    for _ in range(3):
        ctx.extract(pcm)
buf = (ctypes.c_longlong * 128)()
ctx.L.lyra_hip_debug_timing_s2(buf)
t = np.array(buf[:])
names = ["ids+phase", "load XF", "dw0+state", "pw0 gemm+q", "r0b gemm+X1", "resblock1", "resblock2", "lrelu+d2 hist", "down2+bott hist", "bott+feats"]
print("enc_s2 total", t[9] - t[0])
for i in range(1, 10):
    print(f"  {names[i]:18s} {t[i] - t[i-1]:7d}")
for base in (20, 30):
    print("  resblock@", base, " lrelu+dw+state", t[base+1]-t[base], " pw gemm+epi", t[base+2]-t[base+1], " cv gemm+add", t[base+3]-t[base+2])
wall = (t[101] - t[100]) / 100.0  # wall_clock64 ticks at 100 MHz
print(f"  WG0 wall {wall:.1f} us  -> shader clock {(t[9]-t[0])/wall/1e3:.2f} GHz")

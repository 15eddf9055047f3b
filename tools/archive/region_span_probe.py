#!/usr/bin/env python3
"""GPU box: where does a short timed region's constant overhead go?  Host clock around lyra_hip_run_steps_dev(K) +
synchronise vs the GPU's own span (first enc_s0 start -> last dec_s2 end, HIP events on the library's streams)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import lyra_amd
B, bits = 4096, 184
ctx = lyra_amd.LyraHip(max_streams=B)
ctx.torch_order = False
dev = torch.device("cuda", 0)
pcm = torch.randint(-32768, 32768, (32, B, 320), device=dev, dtype=torch.int32).to(torch.int16)
ids = torch.arange(B, device=dev, dtype=torch.int32)
pk = [torch.zeros((B, 23), device=dev, dtype=torch.uint8) for _ in range(2)]
out = [torch.zeros((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
names = ctx.profile_kernel_names()
ctx.L.lyra_hip_profile_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
def run(first, K, prof):
    if prof:
        ctx.profile_enable(True, only=None, every=1)
        ctx.L.lyra_hip_profile_enable(ctx.h, (1 << names.index("enc_s0_kernel")) | (1 << names.index("dec_s2_kernel")))
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.run_steps_dev(ids, bits, K, first_step=first, d_pcm_ring=pcm, d_packets=pk, d_pcm_out=out)
    t_enq = time.perf_counter()
    ctx.synchronize(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    span = None
    if prof:
        cap = 4 * K + 8
        kid = np.zeros(cap, np.int32); a = np.zeros(cap, np.float32); b = np.zeros(cap, np.float32)
        n = ctx.L.lyra_hip_profile_timeline(ctx.h, cap, kid.ctypes.data, a.ctypes.data, b.ctypes.data)
        span = float(b[:n].max())
        e0 = [(a[i], b[i]) for i in range(n) if names[kid[i]] == "enc_s0_kernel"]
        d2 = [(a[i], b[i]) for i in range(n) if names[kid[i]] == "dec_s2_kernel"]
        ctx.profile_read(); ctx.profile_enable(False)
        return (t1 - t0) * 1e3, (t_enq - t0) * 1e3, span, e0, d2
    return (t1 - t0) * 1e3, (t_enq - t0) * 1e3, None, None, None
run(0, 40, False)
step = 40
for K in (20, 20, 100):
    h, enq, _, _, _ = run(step, K, False); step += K
    hp, enqp, span, e0, d2 = run(step, K, True); step += K
    print(f"K={K}: host {h:.3f} ms ({h / K * 1e3:.1f} us/step), enqueue returned after {enq:.3f} ms | with events: host {hp:.3f} ms, GPU span {span:.3f} ms, host - GPU {hp - span:.3f} ms")
    print("   enc_s0 starts (us):", " ".join(f"{x[0] * 1e3:.0f}" for x in e0[:8]), "...  dec_s2 ends (us):", " ".join(f"{x[1] * 1e3:.0f}" for x in d2[:4]), "...", " ".join(f"{x[1] * 1e3:.0f}" for x in d2[-3:]))

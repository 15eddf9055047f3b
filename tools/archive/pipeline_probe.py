#!/usr/bin/env python3
"""Why do enc_s2 / dec_s0 run 1.5-2x slower inside the sustained encode+decode pipeline on some boxes?
Per-kernel HIP-event times (serialised library streams) under different surrounding load:
  full      encode+decode back to back (as benchmarked, serial)
  enc       encoder side only, back to back (extract_dev + rvq_encode_dev)
  dec       decoder side only, back to back (generate_dev)
  full+gap  encode+decode with a host sync and an idle gap of GAP_US after every step
  full+flush  encode+decode, plus a 512 MB device memset between steps (cache flush, no idle)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import lyra_amd
B = int(os.environ.get("B", 4096))
dev = torch.device("cuda", 0)
ctx = lyra_amd.LyraHip(max_streams=B)
ctx.torch_order = False
ctx.set_serial(True)
g = torch.Generator(device=dev); g.manual_seed(1)
pcm = torch.randint(-32768, 32768, (8, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
ids = torch.arange(B, device=dev, dtype=torch.int32)
pk = [torch.empty((B, 23), device=dev, dtype=torch.uint8) for _ in range(2)]
out = [torch.empty((B, 320), device=dev, dtype=torch.int16) for _ in range(2)]
feat = torch.randn((B, 64), device=dev, dtype=torch.float32) * 3
idx = torch.empty((B, 46), device=dev, dtype=torch.int32)
junk = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
torch.cuda.synchronize()
GAP = float(os.environ.get("GAP_US", 300)) * 1e-6


def run(mode, n=60):
    def step(i):
        if mode in ("full", "full+gap", "full+flush"):
            ctx.encode_dev(ids, pcm[i % 8], 184, pk[i & 1])
            ctx.decode_dev(ids, pk[i & 1], 184, out[i & 1])
        elif mode == "enc":
            ctx.extract_dev(ids, pcm[i % 8], feat)
            ctx.rvq_encode_dev(feat, 184, idx)
        elif mode == "dec":
            ctx.generate_dev(ids, feat, out[i & 1])
        if mode == "full+gap":
            ctx.synchronize(); time.sleep(GAP)
        if mode == "full+flush":
            ctx.synchronize(); junk.fill_(i & 255); torch.cuda.synchronize()
    for i in range(20):
        step(i)
    ctx.synchronize(); ctx.profile_enable(True); ctx.profile_read()
    t0 = time.perf_counter()
    for i in range(20, 20 + n):
        step(i)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e6
    p = ctx.profile_read(); ctx.profile_enable(False)
    print(f"{mode:10s} wall/step {wall:7.1f} us | " + "  ".join(f"{k.replace('_kernel','')}={ms / c * 1e3:.1f}" for k, (ms, c) in p.items() if c), flush=True)


# EXIT_AT="d0:43,44,.." (timing variant only): every workgroup of that translation unit's kernels returns at the stamp --
# the kernel's duration is the cumulative cost of the phases before it (outputs are garbage)
if os.environ.get("EXIT_AT"):
    tu, stamps = os.environ["EXIT_AT"].split(":")
    fn = getattr(ctx.L, "lyra_hip_debug_exit_at_" + tu)
    for st_ in stamps.split(","):
        fn(int(st_))
        print("exit_at", tu, st_, end=" | ")
        run("full", 30)
    fn(-1)
    sys.exit(0)
for m in os.environ.get("MODES", "full,enc,dec,full+gap,full+flush,full").split(","):
    run(m)

import os, sys
sys.path.insert(0, "/root/repo")
import torch, lyra_amd
dev = torch.device("cuda", 0)
ctx = lyra_amd.LyraHip(max_streams=8192)
g = torch.Generator(device=dev); g.manual_seed(1)
res=[]
for B in (1024, 2048, 3072, 4096, 6144):
    pcm = torch.randint(-32768, 32768, (12, B, 320), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    ids = torch.arange(B, device=dev, dtype=torch.int32)
    feat = torch.empty((B, 64), device=dev, dtype=torch.float32)
    torch.cuda.synchronize()
    for i in range(4):
        ctx.extract_dev(ids, pcm[i], feat); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_read()
    for i in range(4, 12):
        ctx.extract_dev(ids, pcm[i], feat); ctx.synchronize()
    p = ctx.profile_read(); ctx.profile_enable(False)
    res.append(f"{B}:{p['enc_s0_kernel'][0]/p['enc_s0_kernel'][1]*1e3:.0f}")
print(" ".join(res))

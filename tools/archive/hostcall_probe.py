import sys, time, numpy as np
sys.path.insert(0, '.')
import lyra_amd
for ms in (768, 3072):
    ctx = lyra_amd.LyraHip(max_streams=ms)
    for B in (16, 52, 150, 512, 1024):
        if B > ms: continue
        pcm = np.random.default_rng(0).integers(-32768, 32768, size=(B, 320)).astype(np.int16)
        ids = np.arange(B, dtype=np.int32)
        for _ in range(5): f = ctx.extract(pcm, ids)
        t0 = time.perf_counter()
        for _ in range(50): f = ctx.extract(pcm, ids)
        t1 = time.perf_counter()
        for _ in range(5): pk = ctx.rvq_encode(f, 184) if hasattr(ctx, 'rvq_encode') else None
        print(f"max_streams {ms} B {B}: extract {1e6*(t1-t0)/50:.0f} us per call")
    ctx.close()

#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle comparison (run on the GPU box): localises the first diverging stage."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import lyra_amd  # noqa: E402
from oracle import lyra_oracle  # noqa: E402


def at16(k):
    return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3)


def unperm(a, C):
    """[..., C] in AT16 order -> logical order."""
    idx = np.array([at16(c) for c in range(C)])
    return a[..., idx]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
    g = np.load(os.path.join(ROOT, "tests", "golden", "speech_sample1.npz"))
    B, T = 5, 12
    o = lyra_oracle.Oracle(mode=mode)
    streams = [lyra_oracle.Stream(o, trace_cap=200000) for _ in range(B)]
    ctx = lyra_amd.LyraHip(max_streams=64, requant=mode)
    ids = np.array([3, 17, 1, 40, 9], np.int32)
    ok = True
    for t in range(T):
        pcm = np.stack([g["pcm_in"][(t + 7 * b) % 50] for b in range(B)])
        feat = ctx.extract(pcm, ids)
        e0 = unperm(ctx.debug_read(0, B * 512).reshape(B, 4, 128), 128)
        e1 = unperm(ctx.debug_read(1, B * 512).reshape(B, 2, 256), 256)
        codes = ctx.debug_read(2, B * 64).reshape(B, 64)
        of, taps = [], []
        for b in range(B):
            of.append(streams[b].encode(pcm[b]))
            taps.append(streams[b].taps())
        of = np.stack(of)
        # oracle taps: 0 first conv, 1-3 resblocks@64, 4 down0 out, 5-7 resblocks@128, 8 down1 out, 9 pw0, 10 X1,
        #              11 X2, 12 X3, 13 codes
        d0 = max(np.abs(e0[b].reshape(-1) - taps[b][4]).max() for b in range(B))
        d1 = max(np.abs(e1[b].reshape(-1) - taps[b][8]).max() for b in range(B))
        dc = max(np.abs(codes[b] - taps[b][13]).max() for b in range(B))
        df = np.abs(feat - of).max()
        idx_g = ctx.rvq_encode(feat, 184)
        idx_o = o.rvq_encode(of, 46)
        lossy_g = ctx.rvq_decode(idx_g)
        lossy_o = o.rvq_decode(idx_o)
        pg = ctx.generate(lossy_o, ids)
        x0 = unperm(ctx.debug_read(3, B * 512).reshape(B, 4, 128), 128)
        x1 = unperm(ctx.debug_read(4, B * 1280).reshape(B, 20, 64), 64)
        po, dtaps = [], []
        for b in range(B):
            po.append(streams[b].decode(lossy_o[b]))
            dtaps.append(streams[b].taps())
        po = np.stack(po)
        # decoder taps: 0 head, 1 x164, 2 X1, 3 X2, 4 X3, 5 x231, 6-8 resblocks@128, 9 y20, 10-12 resblocks@64
        dd0 = max(np.abs(x0[b].reshape(-1) - dtaps[b][5]).max() for b in range(B))
        dd1 = max(np.abs(x1[b].reshape(-1) - dtaps[b][9]).max() for b in range(B))
        dp = np.abs(pg.astype(int) - po.astype(int)).max()
        print(f"t={t:2d} enc: s0 {d0:.3g} s1 {d1:.3g} codes {dc:.3g} feat {df:.3g} | rvq idx mism "
              f"{(idx_g != idx_o).sum()} dec {np.abs(lossy_g - lossy_o).max():.3g} | dec: s0 {dd0:.3g} s1 {dd1:.3g} "
              f"pcm {dp}")
        ok &= d0 == 0 and d1 == 0 and dc == 0 and df == 0 and dd0 == 0 and dd1 == 0 and dp == 0
    print("ALL BIT-EXACT" if ok else "MISMATCH")


if __name__ == "__main__":
    main()

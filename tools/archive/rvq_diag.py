#!/usr/bin/env python3
"""GPU box: the screened quantizer against the oracle's all-exact chain -- indices of speech features and of random
vectors at several batch sizes, first mismatches, and how often the screen fell back to the exact chain."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lyra_amd
from oracle import lyra_oracle as lo

O = lo.Oracle(mode="xnnpack")
c = lyra_amd.LyraHip(max_streams=4096)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "speech_sample1.npz"))
rng = np.random.default_rng(1)
sets = {"speech": g["feats_xnnpack"].astype(np.float32), "randn": rng.standard_normal((4096, 64)).astype(np.float32),
        "randn x8": (8 * rng.standard_normal((1024, 64))).astype(np.float32)}
for name, feats in sets.items():
    for B in (1, 5, 16, 48, len(feats)):
        B = min(B, len(feats))
        x = np.ascontiguousarray(feats[:B])
        idx = c.rvq_encode(x, 184)
        ref = O.rvq_encode_batch(x, 46)
        bad = np.argwhere(idx != ref)
        print(f"{name:9s} B={B:5d} mismatching (frame, stage) pairs: {len(bad)}", bad[:6].tolist())
        if len(bad):
            i = bad[0][0]
            print("   gpu", idx[i].tolist()); print("   ref", ref[i].tolist())
    s = c.debug_read(5, 2)
    print(f"   exact-chain frame-stages so far {int(s[0])}, wavefront-stages {int(s[1])}")

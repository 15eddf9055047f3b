"""RVQ index parity at scale ("100 % index match on >= 10^6 random + real vectors") and
ARG_MIN tie behaviour (first minimum: quantizer.tflite `encode` ARG_MIN, lyra/residual_vector_quantizer.cc:95-100).

Ties cannot be provoked with the shipped codebooks short of bit-level search, so the tie tests load a weight container
whose codebooks contain DUPLICATED codewords (exactly equal distances by construction) through
lyra_hip_create_from_image, and the same container into the oracle.
"""
import os
import struct

import numpy as np
import pytest

from oracle import lyra_oracle

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PACK = os.path.join(ROOT, "lyra_amd", "assets", "lyra_v1.lyrapack")


def pack_with_duplicate_codewords(path):
    """Shipped container with ties built in: stage 0 codeword 9 := codeword 3; stage 1: 15 := 14 := 2; stage 5: all
    16 codewords equal; stage 45: codeword 1 := codeword 0."""
    blob = bytearray(open(PACK, "rb").read())
    n = struct.unpack_from("<I", blob, 8)[0]
    off = None
    for i in range(n):
        name, dtype, ndim, s0, s1, s2, s3, o, nb = struct.unpack_from("<56sII4IQQ", blob, 16 + 96 * i)
        if name.rstrip(b"\0") == b"rvq.codebooks":
            off = o
    cb = np.frombuffer(blob, np.float32, 46 * 16 * 64, off).reshape(46, 16, 64).copy()
    cb[0, 9] = cb[0, 3]
    cb[1, 14] = cb[1, 2]
    cb[1, 15] = cb[1, 2]
    cb[5, :] = cb[5, 7]
    cb[45, 1] = cb[45, 0]
    blob[off:off + cb.nbytes] = cb.tobytes()
    open(path, "wb").write(bytes(blob))
    return bytes(blob), cb


def near_codeword_features(cb, n, rng):
    """Sums of one codeword from each of the first two stages plus small noise: every stage-0 / stage-1 codeword
    (the duplicated ones included) is the nearest one for many of these vectors."""
    j0, j1 = rng.integers(0, 16, n), rng.integers(0, 16, n)
    return (cb[0, j0] + cb[1, j1] + rng.normal(0, 0.05, size=(n, 64))).astype(np.float32)


def test_oracle_first_minimum_on_ties(tmp_path):
    p = str(tmp_path / "ties.lyrapack")
    _, cb = pack_with_duplicate_codewords(p)
    o = lyra_oracle.Oracle(pack=p)
    rng = np.random.default_rng(3)
    feats = np.concatenate([rng.normal(0, 3, size=(2000, 64)).astype(np.float32), near_codeword_features(cb, 4000, rng)])
    idx = o.rvq_encode_batch(feats, 46)
    assert not np.any(idx[:, 0] == 9)            # codeword 3 always wins the tie against its copy
    assert not np.any((idx[:, 1] == 14) | (idx[:, 1] == 15))
    assert np.all(idx[:, 5] == 0)                # sixteen equal distances -> index 0
    assert not np.any(idx[:, 45] == 1)
    assert np.any(idx[:, 0] == 3) and np.any(idx[:, 1] == 2)   # the ties do occur


@pytest.mark.gpu
def test_gpu_rvq_ties_first_minimum(tmp_path):
    import lyra_amd
    p = str(tmp_path / "ties.lyrapack")
    image, cb = pack_with_duplicate_codewords(p)
    o = lyra_oracle.Oracle(pack=p)
    ctx = lyra_amd.LyraHip(max_streams=64, weights_image=image)
    rng = np.random.default_rng(4)
    feats = np.concatenate([rng.normal(0, 3, size=(30000, 64)),
                            (rng.integers(-128, 128, size=(30000, 64)) - 20) * 0.26349151134490967,
                            near_codeword_features(cb, 40000, rng)]).astype(np.float32)
    want = o.rvq_encode_batch(feats, 46)
    got = ctx.rvq_encode(feats, 184)
    assert np.array_equal(got, want)
    assert not np.any(got[:, 0] == 9) and np.all(got[:, 5] == 0) and not np.any(got[:, 45] == 1)
    assert np.any(got[:, 0] == 3) and np.any(got[:, 1] == 2)
    ctx.close()


@pytest.mark.gpu
def test_gpu_rvq_million_vectors(golden_dir, oracle_exact):
    """1,048,576 vectors: random normal, on-grid (encoder output lattice 0.26349 * (q - 20)), heavy-tailed, tiny, and
    real encoder features of the reference's speech sample; every index of every stage equal to the oracle's."""
    import lyra_amd
    ctx = lyra_amd.LyraHip(max_streams=64)
    rng = np.random.default_rng(2024)
    n = 1 << 18
    g = np.load(os.path.join(golden_dir, "speech_sample1.npz"))
    speech = np.tile(g["feats_exact"], (n // 50 // 8 + 1, 1))[:n // 8]
    sets = [rng.normal(0, 3, size=(n, 64)),
            (rng.integers(-128, 128, size=(n, 64)) - 20) * 0.26349151134490967,
            rng.standard_t(2, size=(n, 64)) * 2,
            rng.normal(0, 1e-3, size=(n - n // 8, 64)),
            speech + rng.normal(0, 0.05, size=speech.shape)]
    feats = np.concatenate(sets).astype(np.float32)
    assert feats.shape[0] >= 1 << 20
    want = oracle_exact.rvq_encode_batch(feats, 46, threads=16)
    for lo in range(0, feats.shape[0], 1 << 17):
        got = ctx.rvq_encode(feats[lo:lo + (1 << 17)], 184)
        assert np.array_equal(got, want[lo:lo + (1 << 17)]), f"index mismatch in block at {lo}"
    # fewer stages: a prefix, -1 beyond (lyra/residual_vector_quantizer.cc:143-157 convention)
    got = ctx.rvq_encode(feats[:4096], 64)
    assert np.array_equal(got[:, :16], want[:4096, :16]) and np.all(got[:, 16:] == -1)
    ctx.close()

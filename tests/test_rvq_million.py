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
    # exact ties can only be resolved by the exact chain: the screen (csrc/misc_kernels.hip rvq_encode_kernel) must have
    # handed at least every stage-5 frame (sixteen equal codewords) to it
    frames, waves = ctx.debug_read(5, 2)
    assert frames >= feats.shape[0] and waves >= feats.shape[0] / 16
    ctx.close()


@pytest.mark.gpu
def test_gpu_rvq_screen_adversarial_inputs(oracle_exact):
    """The quantizer certifies most indices from MFMA dot products and a rigorous error margin; whatever the margin cannot
    decide -- and whatever is not an ordinary float -- goes to the exact chain.  Inputs chosen to sit on those edges: zeros,
    denormals, 1e-30 .. 1e18 magnitudes (squares overflow), Inf / NaN rows, vectors equidistant from two codewords up to
    the last bits.  Every index equal to the oracle's all-exact chain."""
    import lyra_amd
    ctx = lyra_amd.LyraHip(max_streams=64)
    rng = np.random.default_rng(77)
    base = rng.normal(0, 3, size=(4096, 64)).astype(np.float32)
    sets = [np.zeros((64, 64), np.float32), np.full((64, 64), 1e-40, np.float32)]
    for scale in (1e-30, 1e-12, 1e-6, 1.0, 1e3, 1e9, 1e15, 1e18, 3e19, 1e30):
        sets.append((base[:512] * np.float32(scale)).astype(np.float32))
    weird = base[:256].copy()
    weird[0::4, 5] = np.inf
    weird[1::4, 9] = -np.inf
    weird[2::4, 63] = np.nan
    sets.append(weird)
    # midpoints of two codewords of stage 0 (+ a few ulps): the two best distances agree to the last bits
    import struct as _s
    blob = open(PACK, "rb").read()
    n = _s.unpack_from("<I", blob, 8)[0]
    for i in range(n):
        name, dtype, ndim, s0, s1, s2, s3, o, nb = _s.unpack_from("<56sII4IQQ", blob, 16 + 96 * i)
        if name.rstrip(b"\0") == b"rvq.codebooks":
            cb = np.frombuffer(blob, np.float32, 46 * 16 * 64, o).reshape(46, 16, 64)
    a, b = rng.integers(0, 16, 4096), rng.integers(0, 16, 4096)
    mid = ((cb[0, a].astype(np.float64) + cb[0, b]) / 2).astype(np.float32)
    mid[1::2] = np.nextafter(mid[1::2], np.float32(np.inf))
    sets.append(mid)
    feats = np.concatenate(sets)
    with np.errstate(all="ignore"):
        want = oracle_exact.rvq_encode_batch(feats, 46, threads=16)
    got = ctx.rvq_encode(feats, 184)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"{len(bad)} mismatching (row, stage) pairs, first {bad[:5].tolist()}"
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
    # how often the screen had to fall back to the exact chain: a property of the data (near-ties), not of correctness --
    # but if it were the rule rather than the exception the kernel would be the old one with extra steps
    frames, waves = ctx.debug_read(5, 2)
    assert frames / (46.0 * feats.shape[0]) < 0.02, "exact-chain rate above 2 % of frame-stages"
    ctx.close()

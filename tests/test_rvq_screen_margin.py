"""CPU model of the quantizer's certified screen (lyra_amd/csrc/misc_kernels.hip rvq_encode_kernel).

The GPU kernel ranks a stage's 16 codewords by A_k = N_k - 2 P_k (P_k an fp32 fma chain: the MFMA dot product) and declares the
index CERTIFIED when exactly one integer key lies within the margin M = 2^-14 (Rb + C^2) of the smallest one; everything else
runs the graph's exact chain.  This test restates that decision rule in numpy -- same operations, same roundings -- and checks
its one obligation on inputs built to sit on the edge: whenever the rule certifies an index, it IS the first minimum of the
reference's sequentially rounded sums S_k (three fp32 operations per term, lyra/residual_vector_quantizer.cc:77-110 through
quantizer.tflite `encode`).  It also re-derives the two error bounds the margin rests on against float128 sums.
"""
import os
import struct

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PACK = os.path.join(ROOT, "lyra_amd", "assets", "lyra_v1.lyrapack")
U = np.float64(2.0) ** -24


def _codebooks():
    blob = open(PACK, "rb").read()
    n = struct.unpack_from("<I", blob, 8)[0]
    for i in range(n):
        name, dtype, ndim, s0, s1, s2, s3, o, nb = struct.unpack_from("<56sII4IQQ", blob, 16 + 96 * i)
        if name.rstrip(b"\0") == b"rvq.codebooks":
            return np.frombuffer(blob, np.float32, 46 * 16 * 64, o).reshape(46, 16, 64).copy()
    raise AssertionError("rvq.codebooks not in the container")


def _exact_sums(r, c):
    """S[f, k]: df = r - c, sq = df * df, sum = sum + sq in ascending d, every operation rounded to fp32."""
    s = np.zeros((r.shape[0], c.shape[0]), np.float32)
    for d in range(64):
        df = (r[:, None, d] - c[None, :, d]).astype(np.float32)
        s = (s + (df * df).astype(np.float32)).astype(np.float32)
    return s


def _fma_dot(r, c):
    """P[f, k] = fp32 fma chain over d (what 16 chained v_mfma_f32_16x16x4_f32 compute; the product of two floats is exact in a
    double, one rounding per step)."""
    p = np.zeros((r.shape[0], c.shape[0]), np.float32)
    for d in range(64):
        p = (p.astype(np.float64) + r[:, None, d].astype(np.float64) * c[None, :, d].astype(np.float64)).astype(np.float32)
    return p


def _screen(r, c, rb_slack=1.0009765625):
    """-> (certified [F] bool, index [F]) by the kernel's rule."""
    n2 = (c.astype(np.float64) ** 2).sum(axis=1)
    nn = n2.astype(np.float32)
    c2s = np.nextafter(np.float32(n2.max() * (1.0 + 1e-6) / 16384.0), np.float32(np.inf))
    rb = ((r.astype(np.float64) ** 2).sum(axis=1).astype(np.float32) * np.float32(rb_slack)).astype(np.float32)
    p = _fma_dot(r, c)
    m = (rb.astype(np.float64) * 2.0 ** -14 + np.float64(c2s)).astype(np.float32)          # fma(Rb, 2^-14, c2s)
    off = (rb + m).astype(np.float32)
    a = (np.float64(-2.0) * p.astype(np.float64) + nn[None, :].astype(np.float64)).astype(np.float32)   # fma(-2, P, N)
    a = (a + off[:, None]).astype(np.float32)
    key = (a.view(np.uint32) & np.uint32(0xFFFFFFF0)) | np.arange(16, dtype=np.uint32)[None, :]
    kmin = key.min(axis=1)
    thr = (kmin.view(np.float32) + m).astype(np.float32)
    cnt = (key.view(np.float32) <= thr[:, None]).sum(axis=1)
    ok = (cnt == 1) & (kmin.view(np.float32) < np.float32(2.0 ** 126)) & (kmin.view(np.int32) >= 0)
    return ok, (kmin & np.uint32(15)).astype(np.int64)


def _inputs(cb, rng):
    sets = []
    for stage in (0, 1, 7, 20, 45):
        c = cb[stage]
        scale = float(np.sqrt((c.astype(np.float64) ** 2).sum(axis=1).mean()))
        base = rng.normal(0, scale / 8, size=(3000, 64)).astype(np.float32)
        a, b = rng.integers(0, 16, 3000), rng.integers(0, 16, 3000)
        mid = ((c[a].astype(np.float64) + c[b]) / 2).astype(np.float32)           # equidistant from two codewords ...
        mid_eps = mid + rng.normal(0, scale * 1e-7, size=mid.shape).astype(np.float32)   # ... up to the last bits
        near = (c[a] + rng.normal(0, scale * 1e-3, size=mid.shape)).astype(np.float32)
        sets.append((stage, np.concatenate([base, mid, mid_eps, near, base * np.float32(30), base * np.float32(1e-4)])))
    return sets


def test_certified_index_is_the_reference_argmin():
    cb = _codebooks()
    rng = np.random.default_rng(11)
    n_cert = n_all = 0
    for stage, r in _inputs(cb, rng):
        c = cb[stage]
        ok, idx = _screen(r, c)
        want = _exact_sums(r, c).argmin(axis=1)          # numpy argmin = first minimum, like ARG_MIN
        bad = np.flatnonzero(ok & (idx != want))
        assert bad.size == 0, f"stage {stage}: certified a wrong index for rows {bad[:5].tolist()}"
        n_cert += int(ok.sum()); n_all += ok.size
    assert n_cert > 0.5 * n_all        # the rule does certify (the midpoints are built not to be)


def test_duplicate_codewords_are_never_certified():
    cb = _codebooks()
    c = cb[3].copy()
    c[9] = c[2]
    rng = np.random.default_rng(12)
    r = (c[2] + rng.normal(0, 0.02, size=(500, 64))).astype(np.float32)      # codewords 2 and 9 tie exactly for every row
    ok, _ = _screen(r, c)
    s = _exact_sums(r, c)
    assert np.all(s[:, 2] == s[:, 9])
    tied_best = s.argmin(axis=1) == 2
    assert not np.any(ok & tied_best)        # an exact tie for the minimum is always left to the exact chain


def test_the_two_error_bounds_behind_the_margin():
    """|S - T| <= 67 u T and |A - (T - |r|^2)| <= 2 u |c|^2 + 142 u |r||c|, with T in float128."""
    cb = _codebooks()
    rng = np.random.default_rng(13)
    for stage in (0, 5, 30):
        c = cb[stage]
        scale = float(np.sqrt((c.astype(np.float64) ** 2).sum(axis=1).mean()))
        r = rng.normal(0, scale / 6, size=(4000, 64)).astype(np.float32)
        rl, cl = r.astype(np.longdouble), c.astype(np.longdouble)
        t = ((rl[:, None, :] - cl[None, :, :]) ** 2).sum(axis=2)
        s = _exact_sums(r, c).astype(np.longdouble)
        assert np.all(np.abs(s - t) <= 67 * U * t)
        nn = (cl ** 2).sum(axis=1)
        a = (np.float64(-2.0) * _fma_dot(r, c).astype(np.float64) + nn.astype(np.float32)[None, :].astype(np.float64)).astype(np.float32)
        rr = (rl ** 2).sum(axis=1)
        bound = 2 * U * nn[None, :] + 142 * U * np.sqrt(rr)[:, None] * np.sqrt(nn)[None, :]
        assert np.all(np.abs(a.astype(np.longdouble) - (t - rr[:, None])) <= bound)

"""Property tests (hypothesis) of the oracle's small pieces -- invariants the domain offers that hold for ANY input, next
to the golden vectors of test_oracle_golden.py: packet layout round trips (lyra/packet.h:91-122), the RVQ decoder's
left-to-right sum, the quantizer's first stage being a true nearest-codeword search, the resampler's length rule."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import lyra_oracle

STAGES = {64: 16, 120: 30, 184: 46}


@pytest.fixture(scope="module")
def o():
    return lyra_oracle.Oracle(mode="xnnpack")


@settings(max_examples=60, deadline=None)
@given(bits=st.sampled_from([64, 120, 184]), data=st.data())
def test_pack_unpack_round_trip_and_layout(o, bits, data):
    ns = STAGES[bits]
    idx = np.full((1, 46), -1, np.int32)
    idx[0, :ns] = data.draw(st.lists(st.integers(0, 15), min_size=ns, max_size=ns))
    pk = o.pack(idx, ns)
    assert pk.shape == (1, bits // 8)
    for j in range(ns // 2):                      # first quantizer in the most significant nibble
        assert pk[0, j] == (idx[0, 2 * j] << 4 | idx[0, 2 * j + 1])
    assert np.array_equal(o.unpack(pk, ns), idx)  # unused stages come back as -1


def _codeword(o, k, j):
    """Codeword j of stage k, read through the decoder: every stage contributes v * mask, so an index vector with only
    stage k present decodes to ((0 + .. + 0) + c) + 0 .. = c exactly."""
    one = np.full((1, 46), -1, np.int32)
    one[0, k] = j
    return o.rvq_decode(one)[0]


@settings(max_examples=40, deadline=None)
@given(bits=st.sampled_from([64, 120, 184]), data=st.data())
def test_rvq_decode_is_the_left_to_right_sum_of_codewords(o, bits, data):
    """DecodeToLossyFeatures (residual_vector_quantizer.cc:112-168 over the `decode` subgraph): ((c0 + c1) + c2) + ...
    in float32, strictly left to right, unused stages adding an exact zero."""
    ns = STAGES[bits]
    idx = np.full((1, 46), -1, np.int32)
    idx[0, :ns] = data.draw(st.lists(st.integers(0, 15), min_size=ns, max_size=ns))
    acc = _codeword(o, 0, idx[0, 0]).copy()
    for k in range(1, ns):
        acc = (acc + _codeword(o, k, idx[0, k])).astype(np.float32)
    assert np.array_equal(o.rvq_decode(idx)[0], acc)


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), scale=st.sampled_from([0.01, 1.0, 8.0, 40.0]))
def test_first_rvq_stage_is_the_nearest_codeword(o, seed, scale):
    """Stage 0 of the quantizer must pick a codeword at least as close (float32 sum of squared differences, the
    graph's arithmetic) as every other one; ARG_MIN's tie rule (first minimum) decides among equals."""
    rng = np.random.default_rng(seed)
    f = (rng.normal(size=64) * scale).astype(np.float32)
    best = int(o.rvq_encode(f, 16)[0, 0])
    d = []
    for j in range(16):
        c = _codeword(o, 0, j)
        s = np.float32(0)
        for x, y in zip(f, c):
            df = np.float32(x - y)
            s = np.float32(s + np.float32(df * df))
        d.append(s)
    assert d[best] == min(d) and best == d.index(min(d))


@settings(max_examples=30, deadline=None)
@given(rate=st.sampled_from([8000, 32000, 48000]), hops=st.integers(1, 3), direction=st.booleans(),
       seed=st.integers(0, 10**6))
def test_resampler_length_rule_and_determinism(rate, hops, direction, seed):
    """Resample(n samples at in_rate) -> exactly n * out_rate / in_rate samples (resampler.cc:57-62), independent of
    how the input is cut into hops (the filter state carries over), and zero in -> zero out."""
    i, out = (rate, 16000) if direction else (16000, rate)
    n = hops * (i // 50)
    rng = np.random.default_rng(seed)
    x = rng.integers(-20000, 20000, n).astype(np.int16)
    whole = lyra_oracle.Resampler(i, out).Resample(x)
    assert whole.size == n * out // i
    r = lyra_oracle.Resampler(i, out)
    pieces = np.concatenate([r.Resample(x[k:k + i // 50]) for k in range(0, n, i // 50)])
    assert np.array_equal(pieces, whole)
    assert not lyra_oracle.Resampler(i, out).Resample(np.zeros(n, np.int16)).any()

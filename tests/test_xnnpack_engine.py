"""oracle/_xnn/liblyra_oracle_xnn.so -- the second build of oracle/lyra_oracle.c (-DLO_XNNPACK): the oracle's graph plumbing
with every arithmetic operator (CONV_2D, DEPTHWISE_CONV_2D, TRANSPOSE_CONV, LEAKY_RELU, ADD, QUANTIZE, DEQUANTIZE) executed
by an XNNPACK operator object of torch's libtorch_cpu.so, the way TfLiteModelWrapper with use_xnn = true runs the two graphs
(tflite_model_wrapper.cc:63-85).  It is bench.py's `cpu_baseline_xnnpack`; here it is held to the scalar restatement and to
the committed fixtures (which the flatbuffer interpreter produced): whole graphs chained through real XNNPACK code reproduce
them bit for bit, state carried over 50 hops of speech and over random full-scale PCM.  TEST INFRASTRUCTURE."""
import os

import numpy as np
import pytest

from oracle import lyra_oracle as lo

pytestmark = pytest.mark.skipif(not lo.xnn_engine_available(), reason="oracle/_xnn/liblyra_oracle_xnn.so not built (needs torch's XNNPACK)")


@pytest.fixture(scope="module")
def engines():
    return lo.Oracle(mode="xnnpack"), lo.Oracle(mode="xnnpack", engine="xnnpack")


def test_engine_is_the_xnnpack_build(engines):
    scalar, xnn = engines
    assert xnn.engine == "xnnpack" and xnn.L is not scalar.L
    with pytest.raises(RuntimeError):
        lo.Oracle(mode="exact", engine="xnnpack")     # XNNPACK computes XNNPACK's arithmetic only


def test_whole_graphs_on_xnnpack_operators_reproduce_the_fixtures(engines, golden_dir):
    scalar, xnn = engines
    z = np.load(os.path.join(golden_dir, "speech_sample1.npz"))
    pcm = z["pcm_in"][:, None, :]
    xnn.L.lo_set_canonical_last(1)          # the one-channel last layer on the canonical fma chain (see below)
    r = lo.run_batch(xnn, pcm, 46, do_decode=True, threads=1, want_feats=True)
    assert np.array_equal(r["feats"][:, 0], z["feats_xnnpack"])          # 50 hops x 64 features, bit for bit
    want_pk = scalar.pack(z["idx_xnnpack"], 46)
    assert np.array_equal(r["packets"][:, 0], want_pk)
    assert np.array_equal(r["pcm"][:, 0], z["pcm_xnnpack"])
    # XNNPACK's own choice for the last layer (TRANSPOSE_CONV k64 / s16, ONE output channel: the x86 build's nr2 kernel,
    # four unfused lane sums -- oracle/lyra_oracle.c header): the float output moves by < 1e-8, a rare 1-LSB PCM flip
    xnn.L.lo_set_canonical_last(0)
    r2 = lo.run_batch(xnn, pcm, 46, do_decode=True, threads=1, want_feats=True)
    assert np.array_equal(r2["packets"], r["packets"])
    d = np.abs(r2["pcm"].astype(int) - r["pcm"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.005


@pytest.mark.parametrize("bits", [64, 120, 184])
def test_xnnpack_engine_equals_the_scalar_restatement_on_random_pcm(engines, bits):
    """Full-scale uniform PCM (the bench's workload), several streams on several threads, state carried."""
    scalar, xnn = engines
    rng = np.random.Generator(np.random.PCG64(bits))
    pcm = rng.integers(-32768, 32768, size=(12, 6, 320)).astype(np.int16)
    xnn.L.lo_set_canonical_last(1)
    a = lo.run_batch(scalar, pcm, bits // 4, do_decode=True, threads=3, want_feats=True)
    b = lo.run_batch(xnn, pcm, bits // 4, do_decode=True, threads=3, want_feats=True)
    xnn.L.lo_set_canonical_last(0)
    assert np.array_equal(a["feats"], b["feats"]) and np.array_equal(a["packets"], b["packets"])
    assert np.array_equal(a["pcm"], b["pcm"])

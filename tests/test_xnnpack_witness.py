"""The graph interpreter's "xnnpack" arithmetic (oracle/tflite_interp.py) held against REAL XNNPACK code run here.

The reference executes soundstream_encoder.tflite and lyragan.tflite through TFLite's XNNPACK delegate (use_xnn=true:
soundstream_encoder.cc:39-40, lyra_gan_model.cc:39-40; tflite_model_wrapper.cc:63-85).  TFLite 2.11 itself cannot be built
offline, but an XNNPACK can be run: torch's libtorch_cpu.so in this image exports the library's operator and subgraph API
(header torch/include/xnnpack.h).  oracle/xnn_witness.c drives it exactly as the delegate would -- values defined with the
flatbuffers' own quantisation parameters, one XNNPACK operator per TFLite operator -- and these tests compare it, operator by
operator, with the formulas the oracle (and, through the golden fixtures, the GPU kernels) use in mode "xnnpack".

It is a WITNESS, not the binary of record: newer than the XNNPACK commit TensorFlow 2.11 pins, x86 micro-kernels.  What it
settles (profiles/history/r04_xnnpack_witness.txt has the table):
  * QS8 convolutions requantise in fp32, RNE(float(acc) * scale) -- not the Q31 single rounding ("exact") the survey took for
    XNNPACK's, and not gemmlowp's double rounding;
  * int8 LEAKY_RELU, ADD and QUANTIZE are XNNPACK's own kernels with their own arithmetic, different from the TFLite
    builtin forms rounds 1-3 used for them in every mode;
  * the f32 GEMM / IGEMM / DWCONV / deconvolution micro-kernels start their accumulators from the bias and run one fused
    chain, taps ascending -- now the canonical fp32 order;
  * the last layer (one output channel) is routed to a different x86 kernel (4x2c4 SSE: four lane sums, unfused), which no
    ARM build shares: the canonical order stays the fused chain there, and the difference is bounded below.
Needs the reference flatbuffers and torch: this container only (skipped on the GPU box).
"""
import os

import numpy as np
import pytest

from oracle import xnn_witness

pytestmark = pytest.mark.skipif(not (os.path.isdir("/root/reference/lyra/model_coeffs") and xnn_witness.available()),
                                reason="needs /root/reference and oracle/_xnn/libxnn_witness.so")


def _cmp():
    from oracle import xnn_compare
    return xnn_compare


def test_every_arithmetic_op_of_the_encoder_equals_real_xnnpack():
    xc = _cmp()
    res = xc.per_op_on_traced(xc.ENC, "input_audio", xc.speech_frames(4))
    assert sum(1 for k in res if k[2] == "q") == 34 and sum(1 for k in res if k[2] == "f") == 45   # every arithmetic op visited
    for k, r in sorted(res.items()):
        assert r["refused"] == 0, ("XNNPACK refused to create the operator", k)
        assert r["differ"] == 0, (k, r)


def test_every_arithmetic_op_of_the_decoder_equals_real_xnnpack(golden_dir):
    xc = _cmp()
    g = np.load(os.path.join(golden_dir, "speech_sample1.npz"))
    key = "lossy_xnnpack" if "lossy_xnnpack" in g.files else "lossy_exact"
    frames = [f.reshape(1, 1, 64) for f in g[key][20:24]]
    res = xc.per_op_on_traced(xc.DEC, "input_audio", frames)
    assert sum(1 for k in res if k[2] == "q") == 42 and sum(1 for k in res if k[2] == "f") == 52
    n_last = 0
    for k, r in sorted(res.items()):
        assert r["refused"] == 0, ("XNNPACK refused to create the operator", k)
        assert r["differ"] == 0, (k, r)          # the last layer is compared with the x86 nr2-kernel model (chain_f32.c)
        if r["last"]:
            n_last += 1
            # ... and the canonical fused chain differs from that kernel by rounding noise only: < 2e-6 absolute on |y| <= 1
            assert r["canon_differ"] > 0 and r["maxabs"] < 2e-6, r
    assert n_last == 1


@pytest.mark.parametrize("which", ["ENC", "DEC"])
def test_int8_elementwise_ops_exhaustively(which):
    xc = _cmp()
    rows = xc.elementwise_exhaustive(getattr(xc, which))
    names = [r["name"] for r in rows]
    assert names.count("LEAKY_RELU") == (7 if which == "ENC" else 6) and names.count("ADD") == 2
    for r in rows:
        assert r["vs_model"] == 0, r                       # the "xnnpack" formulas ARE what XNNPACK computes
    # and they are a third flavour: XNNPACK's int8 LeakyReLU differs from the TFLite builtin kernel in several of the 256
    # codes of EVERY instance; QUANTIZE differs on rounding boundaries (ties to even vs away, reciprocal multiply)
    assert all(r["vs_tflite"] > 0 for r in rows if r["name"] == "LEAKY_RELU")
    assert all(r["vs_tflite"] > 0 for r in rows if r["name"] == "QUANTIZE")
    assert all(r["vs_tflite"] == 0 for r in rows if r["name"] == "DEQUANTIZE")


@pytest.mark.parametrize("which", ["ENC", "DEC"])
def test_int8_convolutions_requantise_in_fp32(which):
    xc = _cmp()
    rows = xc.conv_requant_random(getattr(xc, which), rows_per_layer=6000)
    assert len(rows) == (9 if which == "ENC" else 15)
    for r in rows:
        assert r["vs_model"] == 0, r
        assert r["max_acc"] < (1 << 24)                     # float(acc) is exact for every reachable accumulator
    # neither of the two older modes is XNNPACK's arithmetic (gemmlowp: every layer; Q31 single rounding: ~2e-6 of outputs)
    assert all(r["vs_gemmlowp"] > 0 for r in rows)
    assert sum(r["vs_exact"] for r in rows) > 0


@pytest.mark.parametrize("which", ["ENC", "DEC"])
def test_fp32_layers_are_bias_first_fused_chains(which):
    xc = _cmp()
    rows = xc.fp32_layers_random(getattr(xc, which))
    assert len(rows) == (23 if which == "ENC" else 21)
    for r in rows:
        if r["last"]:
            assert r["vs_c4"] == 0 and r["vs_bias_first"] > 0 and r["max_rel_vs_bias_first"] < 1e-5, r
        else:
            assert r["vs_bias_first"] == 0, r
    # the order rounds 1-3 called canonical (chain from 0, bias added last) is NOT what XNNPACK computes
    assert sum(r["vs_round3_order"] for r in rows if r["name"] != "DEPTHWISE_CONV_2D") > 0

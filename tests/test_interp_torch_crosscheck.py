"""A second, independently structured check of oracle/tflite_interp.py (the graph interpreter that produced the golden
fixtures): every CONV_2D / DEPTHWISE_CONV_2D / TRANSPOSE_CONV of soundstream_encoder.tflite and lyragan.tflite is
recomputed from the interpreter's own traced operand values with torch's float64 conv1d / conv_transpose1d (different
code path: no patch extraction, no per-tap matmul loops) and, for the int8 layers, a requantisation written with Python
integers (no numpy int64 vector code shared with the interpreter).

It validates the interpreter's convolution arithmetic layer by layer on real speech with live state; it does NOT turn the
parity claim into "pinned to TFLite" -- no TFLite binary exists offline (README "Parity").
Needs the reference's flatbuffers: skipped on the GPU box.
"""
import math
import os
import wave

import numpy as np
import pytest

REF = "/root/reference/lyra"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF + "/model_coeffs"), reason="reference flatbuffers not present")


def _qm(m):
    """TFLite QuantizeMultiplier with Python numbers only."""
    if m == 0.0:
        return 0, 0
    q, shift = math.frexp(m)
    M = int(math.floor(q * (1 << 31) + 0.5))
    if M == 1 << 31:
        M //= 2
        shift += 1
    return (0, 0) if shift < -31 else (M, shift)


def _requant_py(acc, M, shift, mode):
    acc = int(acc)
    if mode == "exact":                      # single rounding of the exact Q31 product (ruy / XNNPACK)
        total = 31 - shift
        return (acc * M + (1 << (total - 1))) >> total
    left, right = max(shift, 0), max(-shift, 0)   # gemmlowp: SRDHM then rounding divide by POT
    a = acc << left
    ab = a * M
    nudge = (1 << 30) if ab >= 0 else 1 - (1 << 30)
    t = ab + nudge
    x = t >> 31 if t >= 0 else -((-t) >> 31)     # C++ truncating division
    if a == -(1 << 31) and M == -(1 << 31):
        x = (1 << 31) - 1
    mask = (1 << right) - 1
    rem = x & mask
    thr = (mask >> 1) + (1 if x < 0 else 0)
    return (x >> right) + (1 if rem > thr else 0)


def _check_graph(path, sig_input, frames, mode):
    import torch
    import torch.nn.functional as F
    from oracle.tflite_interp import Interpreter
    it = Interpreter(path, requant=mode, acc64=True)
    sg = it.model.subgraphs[0]
    T = sg.tensors
    n_checked = {"CONV_2D": 0, "DEPTHWISE_CONV_2D": 0, "TRANSPOSE_CONV": 0}
    for x in frames:
        it.trace = {}
        it.run_signature("serving_default", {sig_input: x})
        tr = it.trace
        for op in sg.ops:
            if op.name not in n_checked:
                continue
            if op.name == "TRANSPOSE_CONV":
                w_i, x_i, b_i = op.inputs[1], op.inputs[2], op.inputs[3]
            else:
                x_i, w_i, b_i = op.inputs[0], op.inputs[1], op.inputs[2]
            xin, w, y = tr[x_i], tr[w_i], tr[op.outputs[0]]
            b = tr[b_i] if b_i >= 0 else None
            is_q = xin.dtype == np.int8
            xt = torch.from_numpy((xin.astype(np.float64) - (int(T[x_i].zero_point[0]) if is_q else 0))
                                  .reshape(xin.shape[1], xin.shape[3]).T.copy()).unsqueeze(0)   # [1, Cin, H]
            wt = torch.from_numpy(w.astype(np.float64))
            if op.name == "CONV_2D":            # w [Cout, K, 1, Cin/g]
                groups = xin.shape[3] // w.shape[3]
                acc = F.conv1d(xt, wt[:, :, 0, :].permute(0, 2, 1).contiguous(), stride=op.opts["stride_h"], groups=groups)
            elif op.name == "DEPTHWISE_CONV_2D":  # w [1, K, 1, C]
                C = w.shape[3]
                acc = F.conv1d(xt, wt[0, :, 0, :].T.reshape(C, 1, -1).contiguous(), dilation=op.opts["dil_h"], groups=C)
            else:                               # w [Cout, K, 1, Cin] -> conv_transpose1d weight [Cin, Cout, K]
                acc = F.conv_transpose1d(xt, wt[:, :, 0, :].permute(2, 0, 1).contiguous(), stride=op.opts["stride_h"])
            acc = acc[0].T.numpy()              # [Hout, Cout]
            if b is not None:
                acc = acc + b.astype(np.float64)[None, :]
            yy = y.reshape(acc.shape)
            if not is_q:
                assert np.allclose(yy, acc, rtol=2e-6, atol=2e-6), (op.name, op.outputs[0])
            else:
                acc_i = np.rint(acc).astype(np.int64)
                assert np.abs(acc - acc_i).max() == 0.0          # integer sums are exact in float64
                ws = T[w_i].scale.astype(np.float64)
                cout = acc.shape[1]
                ws = np.repeat(ws, cout) if ws.size == 1 else ws
                s_in, s_out, z_out = float(T[x_i].scale[0]), float(T[op.outputs[0]].scale[0]), int(T[op.outputs[0]].zero_point[0])
                want = np.empty_like(acc_i)
                for c in range(cout):
                    M, sh = _qm(s_in * float(ws[c]) / s_out)
                    for r in range(acc_i.shape[0]):
                        want[r, c] = min(127, max(-128, _requant_py(acc_i[r, c], M, sh, mode) + z_out))
                assert np.array_equal(yy.astype(np.int64), want), (op.name, op.outputs[0], mode)
            n_checked[op.name] += 1
    return n_checked


def _speech(n):
    w = wave.open(REF + "/testdata/sample1_16kHz.wav")
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    return pcm[20 * 320:(20 + n) * 320].reshape(n, 320)   # skip the leading near-silence


@pytest.mark.parametrize("mode", ["exact", "gemmlowp_double"])
def test_encoder_convs_vs_torch(mode):
    frames = [(h.astype(np.float32) / np.float32(32768.0)).reshape(1, 320) for h in _speech(3)]
    n = _check_graph(REF + "/model_coeffs/soundstream_encoder.tflite", "input_audio", frames, mode)
    assert n == {"CONV_2D": 23 * 3, "DEPTHWISE_CONV_2D": 9 * 3, "TRANSPOSE_CONV": 0}   # SURVEY.md section 2 op histogram


@pytest.mark.parametrize("mode", ["exact", "gemmlowp_double"])
def test_decoder_convs_vs_torch(mode, golden_dir):
    g = np.load(os.path.join(golden_dir, "speech_sample1.npz"))
    frames = [f.reshape(1, 1, 64) for f in g["lossy_exact"][20:23]]
    n = _check_graph(REF + "/model_coeffs/lyragan.tflite", "input_audio", frames, mode)
    assert n == {"CONV_2D": 19 * 3, "DEPTHWISE_CONV_2D": 9 * 3, "TRANSPOSE_CONV": 8 * 3}

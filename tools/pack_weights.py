#!/usr/bin/env python3
"""Pack a Lyra model directory (the three .tflite graphs + lyra_config.binarypb)
into one flat little-endian tensor container, `lyra_v1.lyrapack`.

Why: TFLite (the reference's numeric engine, WORKSPACE:168-174) is not
available, and the GPU box has no /root/reference.  The product
(lyra_amd/csrc) and the C oracle (oracle/lyra_oracle.c) both load this
container; it holds nothing but the reference's coefficients and quantisation
parameters re-keyed by role, in TFLite tensor layout.

    python tools/pack_weights.py /root/reference/lyra/model_coeffs lyra_amd/assets/lyra_v1.lyrapack

Container format (all little-endian):
    char     magic[8] = "LYRAPK01"
    uint32   n_entries
    uint32   reserved
    entry[n] { char name[56]; uint32 dtype (0 f32, 1 i8, 2 i32); uint32 ndim;
               uint32 shape[4]; uint64 offset (from file start); uint64 nbytes }
    payloads, each 64-byte aligned

Naming: ops of each kind are numbered in graph order, e.g. `enc.conv.7.w`,
`enc.conv.7.b`, `enc.conv.7.q` (f32[4] = in_scale, in_zp, out_scale, out_zp;
zeros for float ops), `enc.conv.7.wscale` (per-channel weight scales, int8
only), `enc.conv.7.opt` (i32[4] = stride, dilation, groups, kernel).
Kinds: conv (CONV_2D), dw (DEPTHWISE_CONV_2D), tconv (TRANSPOSE_CONV),
lrelu8 (int8 LEAKY_RELU), add8 (int8 ADD), quant (QUANTIZE), dequant
(DEQUANTIZE), sub (SUB constants = float biases removed from tconv state).
"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import tflite_reader as tr  # noqa: E402

DT = {np.dtype(np.float32): 0, np.dtype(np.int8): 1, np.dtype(np.int32): 2}


def qparams(t):
    if len(t.scale) == 0:
        return 0.0, 0.0
    assert len(t.scale) == 1, t
    return float(t.scale[0]), float(t.zero_point[0])


def walk(prefix, sg, out):
    cnt = {}

    def nxt(kind):
        i = cnt.get(kind, 0)
        cnt[kind] = i + 1
        return f"{prefix}.{kind}.{i}"

    T = sg.tensors
    for op in sg.ops:
        if op.name in ("CONV_2D", "DEPTHWISE_CONV_2D"):
            kind = "conv" if op.name == "CONV_2D" else "dw"
            x, w, b = (T[i] for i in op.inputs)
            y = T[op.outputs[0]]
            base = nxt(kind)
            if kind == "conv":
                cout, k, _, cig = w.shape
                groups = x.shape[3] // cig
                wd = w.data.reshape(cout, k, cig)
            else:
                _, k, _, c = w.shape
                groups = c
                wd = w.data.reshape(k, c)
            out[base + ".w"] = wd
            out[base + ".b"] = b.data
            out[base + ".q"] = np.array([*qparams(x), *qparams(y)], np.float32)
            out[base + ".opt"] = np.array([op.opts["stride_h"], op.opts["dil_h"], groups, k], np.int32)
            if w.type == 9:
                assert np.all(w.zero_point == 0)
                out[base + ".wscale"] = w.scale.astype(np.float32)
                # bias scale must be s_in * s_w (TFLite convention)
                assert np.allclose(b.scale, x.scale[0] * w.scale, rtol=1e-6)
        elif op.name == "TRANSPOSE_CONV":
            w, x, b = T[op.inputs[1]], T[op.inputs[2]], T[op.inputs[3]]
            y = T[op.outputs[0]]
            base = nxt("tconv")
            cout, k, _, cin = w.shape
            out[base + ".w"] = w.data.reshape(cout, k, cin)
            out[base + ".b"] = b.data
            out[base + ".q"] = np.array([*qparams(x), *qparams(y)], np.float32)
            out[base + ".opt"] = np.array([op.opts["stride_h"], 1, 1, k], np.int32)
            if w.type == 9:
                assert np.all(w.zero_point == 0) and len(w.scale) == 1
                out[base + ".wscale"] = w.scale.astype(np.float32)
        elif op.name == "LEAKY_RELU":
            x, y = T[op.inputs[0]], T[op.outputs[0]]
            assert abs(op.opts["alpha"] - 0.30000001192092896) < 1e-12
            if x.type == 9:
                out[nxt("lrelu8") + ".q"] = np.array([*qparams(x), *qparams(y)], np.float32)
        elif op.name == "ADD":
            a, b = T[op.inputs[0]], T[op.inputs[1]]
            y = T[op.outputs[0]]
            if a.type == 9:
                out[nxt("add8") + ".q"] = np.array([*qparams(a), *qparams(b), *qparams(y)], np.float32)
            elif a.data is not None or b.data is not None:
                raise AssertionError("unexpected constant float ADD")
        elif op.name == "QUANTIZE":
            out[nxt("quant") + ".q"] = np.array(qparams(T[op.outputs[0]]), np.float32)
        elif op.name == "DEQUANTIZE":
            out[nxt("dequant") + ".q"] = np.array(qparams(T[op.inputs[0]]), np.float32)
        elif op.name == "SUB":
            c = T[op.inputs[1]]
            assert c.data is not None
            out[nxt("sub") + ".c"] = c.data.astype(np.float32)
        elif op.name == "CONCATENATION":
            # constant operands of time-axis concats must be all-zero (tconv state padding)
            for i in op.inputs:
                if T[i].data is not None:
                    assert not np.any(T[i].data), "non-zero constant in CONCATENATION"
    return cnt


def collect(model_dir):
    out = {}
    enc = tr.load(os.path.join(model_dir, "soundstream_encoder.tflite"))
    gan = tr.load(os.path.join(model_dir, "lyragan.tflite"))
    qz = tr.load(os.path.join(model_dir, "quantizer.tflite"))
    c1 = walk("enc", enc.subgraphs[0], out)
    c2 = walk("dec", gan.subgraphs[0], out)
    # init subgraphs must assign zeros only (CALL_ONCE -> NoOp)
    for m in (enc, gan):
        for t in m.subgraphs[1].tensors:
            if t.data is not None and t.dtype == np.float32:
                assert not np.any(t.data), "non-zero initial state"
    # RVQ codebooks: second operand of each SQUARED_DIFFERENCE in the encode subgraph
    sgi = qz.signatures["encode"]["subgraph"]
    sg = qz.subgraphs[sgi]
    cbs = [sg.tensors[op.inputs[1]].data.reshape(16, 64) for op in sg.ops if op.name == "SQUARED_DIFFERENCE"]
    assert len(cbs) == 46
    cb = np.stack(cbs).astype(np.float32)
    # the gather tables of encode and decode must be the same codebooks
    for name in ("encode", "decode"):
        s2 = qz.subgraphs[qz.signatures[name]["subgraph"]]
        gs = [s2.tensors[op.inputs[0]].data.reshape(16, 64) for op in s2.ops if op.name == "GATHER"]
        n = len(gs)
        assert n in (45, 46)
        # (the decode graph lists its GATHERs in name order, not stage order)
        for g in gs:
            assert any(np.array_equal(g, cb[k]) for k in range(46)), name
    out["rvq.codebooks"] = cb
    with open(os.path.join(model_dir, "lyra_config.binarypb"), "rb") as f:
        pb = f.read()
    assert len(pb) == 2 and pb[0] == 0x08, pb  # field 1 varint: identifier
    out["meta.version"] = np.array([pb[1]], np.int32)
    return out, c1, c2


def write(out, path):
    names = sorted(out)
    hdr = 16 + 96 * len(names)
    off = (hdr + 63) // 64 * 64
    entries, payload = [], []
    for n in names:
        a = np.ascontiguousarray(out[n])
        if a.ndim == 0:
            a = a.reshape(1)
        assert a.dtype in DT and a.ndim <= 4 and len(n) < 56, (n, a.dtype, a.shape)
        shape = list(a.shape) + [1] * (4 - a.ndim)
        nb = a.nbytes
        entries.append(struct.pack("<56sII4IQQ", n.encode(), DT[a.dtype], a.ndim, *shape, off, nb))
        payload.append((off, a.tobytes()))
        off = (off + nb + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(b"LYRAPK01" + struct.pack("<II", len(names), 0))
        for e in entries:
            f.write(e)
        for o, b in payload:
            f.seek(o)
            f.write(b)
        f.truncate(off)
    return off


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/lyra/model_coeffs"
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "lyra_amd", "assets", "lyra_v1.lyrapack")
    out, c1, c2 = collect(src)
    n = write(out, dst)
    print("encoder op counts", c1)
    print("decoder op counts", c2)
    print(f"{len(out)} tensors, {n} bytes -> {dst}")

"""Comparisons between the graph interpreter's arithmetic models and REAL XNNPACK -- TEST INFRASTRUCTURE ONLY.

Shared by tests/test_xnnpack_witness.py (asserts) and tools/xnn_witness_report.py (writes the table under profiles/).
Everything here needs the reference's flatbuffers and torch's libtorch_cpu.so: this container only.
"""
import wave

import numpy as np

from . import tflite_interp as ti
from . import tflite_reader as tr
from . import xnn_witness as xw

REF = "/root/reference/lyra"
ENC = REF + "/model_coeffs/soundstream_encoder.tflite"
DEC = REF + "/model_coeffs/lyragan.tflite"
ARITH_OPS = ("CONV_2D", "DEPTHWISE_CONV_2D", "TRANSPOSE_CONV", "LEAKY_RELU", "ADD", "QUANTIZE", "DEQUANTIZE")
F32 = np.float32


def speech_frames(n, start=20):
    w = wave.open(REF + "/testdata/sample1_16kHz.wav")
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    return [(h.astype(F32) / F32(32768.0)).reshape(1, 320) for h in pcm[start * 320:(start + n) * 320].reshape(n, 320)]


def is_q_op(sg, op):
    T = sg.tensors
    return any(T[i].dtype == np.int8 for i in list(op.inputs) + list(op.outputs) if i >= 0)


def last_layer(sg, op):
    """the one-output-channel fp32 TRANSPOSE_CONV that ends lyragan.tflite (x86 XNNPACK: nr2 kernel, see chain_f32.c)"""
    return op.name == "TRANSPOSE_CONV" and sg.tensors[op.inputs[1]].shape[0] == 1


def per_op_on_traced(path, key, frames, mode="xnnpack"):
    """Run the interpreter in `mode` over `frames` (state carried) and, for every arithmetic op, recompute the op's output
    from the interpreter's own traced inputs with real XNNPACK.  Returns {(op index, name, 'q'|'f'): dict(n, differ,
    refused, maxabs)}."""
    it = ti.Interpreter(path, requant=mode)
    sg = it.model.subgraphs[0]
    T = sg.tensors
    res = {}
    for x in frames:
        it.trace = {}
        it.run_signature("serving_default", {key: x})
        trc = it.trace
        for op in sg.ops:
            if op.name not in ARITH_OPS:
                continue
            k = (op.index, op.name, "q" if is_q_op(sg, op) else "f")
            r = res.setdefault(k, dict(n=0, differ=0, refused=0, maxabs=0.0, last=last_layer(sg, op)))
            y = xw.run_op(sg, op, trc)
            if y is None:
                r["refused"] += 1
                continue
            want = trc[op.outputs[0]]
            y = y.reshape(want.shape)
            r["n"] += want.size
            if r["last"]:   # compare with the x86 nr2-kernel model instead of the canonical chain
                xin = trc[op.inputs[2]]
                want_c4 = ti.chain_deconv_c4(xin.reshape(xin.shape[1], xin.shape[3]), trc[op.inputs[1]],
                                             trc[op.inputs[3]] if op.inputs[3] >= 0 else None, op.opts["stride_h"])
                r["differ"] += int((y.reshape(-1) != want_c4.reshape(-1)).sum())
                r["maxabs"] = max(r["maxabs"], float(np.abs(y - want).max()))     # vs the canonical chain
                r["canon_differ"] = r.get("canon_differ", 0) + int((y != want).sum())
                continue
            r["differ"] += int((y != want).sum())
            if want.dtype == np.int8:
                r["maxabs"] = max(r["maxabs"], float(np.abs(y.astype(int) - want.astype(int)).max()))
            else:
                r["maxabs"] = max(r["maxabs"], float(np.abs(y - want).max()))
    return res


def _interp_pair(path):
    return {m: ti.Interpreter(path, requant=m) for m in ("exact", "gemmlowp_double", "xnnpack")}


def elementwise_exhaustive(path):
    """Every int8 LEAKY_RELU over its 256 codes, every int8 ADD over its 65,536 code pairs, every QUANTIZE over the
    neighbourhood of every rounding boundary + 200,000 random floats, every DEQUANTIZE over 256 codes: real XNNPACK vs
    the interpreter's "xnnpack" model and vs the TFLite-builtin forms.  Returns a list of row dicts."""
    its = _interp_pair(path)
    sg = its["exact"].model.subgraphs[0]
    T = sg.tensors
    rows = []
    for op in sg.ops:
        if op.name == "LEAKY_RELU" and T[op.inputs[0]].dtype == np.int8:
            ti_, to = T[op.inputs[0]], T[op.outputs[0]]
            x = np.arange(-128, 128).astype(np.int8)
            G = xw.Graph()
            a = G.q8((1, 256), ti_.scale[0], ti_.zero_point[0], io=1)
            y = G.q8((1, 256), to.scale[0], to.zero_point[0], io=2)
            G.leaky_relu(a, y, op.opts["alpha"])
            out = next(iter(G.run({a: x}).values())).ravel()
            G.close()
            rows.append(dict(op=op.index, name="LEAKY_RELU", n=256,
                             vs_model=int((out != its["xnnpack"]._lrelu(ti_, to, x, op.opts["alpha"])).sum()),
                             vs_tflite=int((out != its["exact"]._lrelu(ti_, to, x, op.opts["alpha"])).sum())))
        elif op.name == "ADD" and T[op.inputs[0]].dtype == np.int8:
            t1, t2, to = T[op.inputs[0]], T[op.inputs[1]], T[op.outputs[0]]
            aa, bb = np.meshgrid(np.arange(-128, 128), np.arange(-128, 128))
            aa = aa.astype(np.int8).ravel()
            bb = bb.astype(np.int8).ravel()
            G = xw.Graph()
            a = G.q8((65536,), t1.scale[0], t1.zero_point[0], io=1)
            b = G.q8((65536,), t2.scale[0], t2.zero_point[0], io=1)
            y = G.q8((65536,), to.scale[0], to.zero_point[0], io=2)
            G.add(a, b, y)
            out = next(iter(G.run({a: aa, b: bb}).values())).ravel()
            G.close()
            rows.append(dict(op=op.index, name="ADD", n=65536,
                             vs_model=int((out != its["xnnpack"]._add(t1, t2, to, [aa, bb])).sum()),
                             vs_tflite=int((out != its["exact"]._add(t1, t2, to, [aa, bb])).sum())))
        elif op.name == "QUANTIZE":
            to = T[op.outputs[0]]
            s, z = F32(to.scale[0]), int(to.zero_point[0])
            rng = np.random.default_rng(op.index)
            base = ((np.arange(-140, 140) + 0.5) * np.float64(s)).astype(F32)
            pts = [base]
            up, dn = base.copy(), base.copy()
            for _ in range(5):
                up = np.nextafter(up, F32(np.inf))
                dn = np.nextafter(dn, F32(-np.inf))
                pts += [up.copy(), dn.copy()]
            nb = sum(p.size for p in pts)
            x = np.concatenate(pts + [rng.normal(0, 40 * s, 200000).astype(F32)])
            G = xw.Graph()
            a = G.f32((x.size,), io=1)
            y = G.q8((x.size,), s, z, io=2)
            G.convert(a, y)
            out = next(iter(G.run({a: x}).values())).ravel()
            G.close()
            tfl = its["exact"]._quantize(None, to, x)
            rows.append(dict(op=op.index, name="QUANTIZE", n=int(x.size),
                             vs_model=int((out != its["xnnpack"]._quantize(None, to, x)).sum()),
                             vs_tflite=int((out != tfl).sum()), vs_tflite_random=int((out[nb:] != tfl[nb:]).sum())))
        elif op.name == "DEQUANTIZE":
            t0 = T[op.inputs[0]]
            x = np.arange(-128, 128).astype(np.int8)
            G = xw.Graph()
            a = G.q8((256,), t0.scale[0], t0.zero_point[0], io=1)
            y = G.f32((256,), io=2)
            G.convert(a, y)
            out = next(iter(G.run({a: x}).values())).ravel()
            G.close()
            want = (np.float64(t0.scale[0]) * (x.astype(np.int64) - int(t0.zero_point[0]))).astype(F32)
            rows.append(dict(op=op.index, name="DEQUANTIZE", n=256, vs_model=int((out != want).sum()), vs_tflite=int((out != want).sum())))
    return rows


def _acc_conv(op, T, x):
    """int64 accumulators (bias included) of an int8 CONV_2D / DEPTHWISE_CONV_2D with k = 1 taps or depthwise, on [H, C]."""
    xt, wt, bt = T[op.inputs[0]], T[op.inputs[1]], T[op.inputs[2]]
    H, C = x.shape
    xx = x.astype(np.int64) - int(xt.zero_point[0])
    o = op.opts
    if op.name == "DEPTHWISE_CONV_2D":
        K, dil = int(wt.shape[1]), o["dil_h"]
        w = wt.data.astype(np.int64).reshape(K, C)
        Ho = H - (K - 1) * dil
        acc = sum(xx[k * dil:k * dil + Ho] * w[k][None, :] for k in range(K))
    else:
        cout, K, _, cig = (int(v) for v in wt.shape)
        g = C // cig
        cog = cout // g
        st = o["stride_h"]
        Ho = (H - K) // st + 1
        idx = (np.arange(Ho) * st)[:, None] + np.arange(K)[None, :]
        acc = np.zeros((Ho, cout), np.int64)
        for gi in range(g):
            patches = xx[:, gi * cig:(gi + 1) * cig][idx].reshape(Ho, K * cig)
            acc[:, gi * cog:(gi + 1) * cog] = patches @ wt.data[gi * cog:(gi + 1) * cog].astype(np.int64).reshape(cog, K * cig).T
    return acc + bt.data.astype(np.int64)[None, :]


def conv_requant_random(path, rows_per_layer=20000):
    """Every int8 CONV_2D / DEPTHWISE_CONV_2D / TRANSPOSE_CONV on `rows_per_layer` rows of uniformly random codes: real XNNPACK
    vs the three requantisation models applied to the exact integer accumulators."""
    m = tr.load(path)
    sg = m.subgraphs[0]
    T = sg.tensors
    out = []
    for op in sg.ops:
        if op.name not in ("CONV_2D", "DEPTHWISE_CONV_2D", "TRANSPOSE_CONV"):
            continue
        tc = op.name == "TRANSPOSE_CONV"
        xi, wi, bi = (op.inputs[2], op.inputs[1], op.inputs[3]) if tc else op.inputs[:3]
        xt, wt, ot = T[xi], T[wi], T[op.outputs[0]]
        if xt.dtype != np.int8:
            continue
        rng = np.random.default_rng(1000 + op.index)
        C = int(xt.shape[3])
        H = rows_per_layer if not tc else max(64, rows_per_layer // 16)
        x = rng.integers(-128, 128, size=(H, C)).astype(np.int8)
        o = op.opts
        y = xw.conv(x, wt.data, T[bi].data if bi >= 0 else None, kind="dw" if op.name[0] == "D" else "conv",
                    stride=o["stride_h"], dil=o.get("dil_h", 1), transpose=tc, in_q=(xt.scale[0], xt.zero_point[0]),
                    w_scale=wt.scale, out_q=(ot.scale[0], ot.zero_point[0]))
        assert y is not None, op
        if tc:
            cout, K, _, cin = (int(v) for v in wt.shape)
            st = o["stride_h"]
            xx = x.astype(np.int64) - int(xt.zero_point[0])
            acc = np.zeros(((H - 1) * st + K, cout), np.int64)
            w = wt.data.astype(np.int64).reshape(cout, K, cin)
            for k in range(K):
                acc[k:k + (H - 1) * st + 1:st] += xx @ w[:, k, :].T
            if bi >= 0:
                acc += T[bi].data.astype(np.int64)[None, :]
        else:
            acc = _acc_conv(op, T, x)
        zo = int(ot.zero_point[0])
        nch = acc.shape[1]
        wsc = wt.scale.astype(F32)
        wsc = np.repeat(wsc, nch) if wsc.size == 1 else wsc
        ws64 = wsc.astype(np.float64)
        MS = [ti.quantize_multiplier(np.float64(xt.scale[0]) * w_ / np.float64(ot.scale[0])) for w_ in ws64]
        Ms = np.array([a for a, _ in MS], np.int64)
        Ss = np.array([b for _, b in MS], np.int64)
        yy = y.astype(np.int64)
        ex = np.clip(ti.mbqm_exact(acc, Ms, Ss) + zo, -128, 127)
        db = np.clip(ti.mbqm_double(acc, Ms, Ss) + zo, -128, 127)
        xm = ti.xnn_requant(acc, xt.scale[0], wsc, ot.scale[0], zo).astype(np.int64)
        out.append(dict(op=op.index, name=op.name, n=int(yy.size), max_acc=int(np.abs(acc).max()),
                        vs_model=int((yy != xm).sum()), vs_exact=int((yy != ex).sum()), vs_gemmlowp=int((yy != db).sum())))
    return out


def fp32_layers_random(path, rows=64):
    """Every fp32 CONV_2D / DEPTHWISE_CONV_2D / TRANSPOSE_CONV on random inputs: real XNNPACK vs the candidate fmaf chains."""
    m = tr.load(path)
    sg = m.subgraphs[0]
    T = sg.tensors
    out = []
    for op in sg.ops:
        if op.name not in ("CONV_2D", "DEPTHWISE_CONV_2D", "TRANSPOSE_CONV"):
            continue
        tc = op.name == "TRANSPOSE_CONV"
        xi, wi, bi = (op.inputs[2], op.inputs[1], op.inputs[3]) if tc else op.inputs[:3]
        xt, wt = T[xi], T[wi]
        if xt.dtype != np.float32:
            continue
        rng = np.random.default_rng(2000 + op.index)
        o = op.opts
        x = rng.normal(0, 1, size=(rows, int(xt.shape[3]))).astype(F32)
        b = T[bi].data if bi >= 0 else None
        y = xw.conv(x, wt.data, b, kind="dw" if op.name[0] == "D" else "conv", stride=o["stride_h"], dil=o.get("dil_h", 1),
                    transpose=tc)
        assert y is not None, op
        if tc:
            c1 = ti.chain_deconv(x, wt.data, b, o["stride_h"], True, 0)
            c0 = ti.chain_deconv(x, wt.data, b, o["stride_h"], False, 1)     # round-3 canonical order: oldest first, bias last
        elif op.name[0] == "D":
            c1 = ti.chain_dwconv(x, wt.data, b, o["dil_h"], True)
            c0 = ti.chain_dwconv(x, wt.data, b, o["dil_h"], False)
        else:
            c1 = ti.chain_conv(x, wt.data, b, o["stride_h"], o["dil_h"], True)
            c0 = ti.chain_conv(x, wt.data, b, o["stride_h"], o["dil_h"], False)
        row = dict(op=op.index, name=op.name, shape=tuple(int(v) for v in wt.shape), n=int(y.size),
                   vs_bias_first=int((y != c1).sum()), vs_round3_order=int((y != c0).sum()), last=last_layer(sg, op))
        if row["last"]:
            row["vs_c4"] = int((y != ti.chain_deconv_c4(x, wt.data, b, o["stride_h"])).sum())
            row["max_rel_vs_bias_first"] = float(np.abs(y - c1).max() / np.abs(c1).max())
        out.append(row)
    return out

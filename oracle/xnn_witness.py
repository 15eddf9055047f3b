"""ctypes front of oracle/_xnn/libxnn_witness.so -- TEST INFRASTRUCTURE ONLY.

Rebuilds operators (or whole regions) of the reference's flatbuffers as XNNPACK subgraphs, the way TFLite's XNNPACK
delegate does (tflite_model_wrapper.cc:63-85 of the reference turns it on for the encoder and the GAN), and runs them on
the XNNPACK that torch's libtorch_cpu.so carries in this image.  See oracle/xnn_witness.c for what this is and is not.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
XNN_EXTRA_BYTES = 128


def available():
    return os.path.exists(os.path.join(_HERE, "_xnn", "libxnn_witness.so"))


def lib():
    global _LIB
    if _LIB is None:
        import torch  # noqa: F401  (maps libtorch_cpu.so and its dependencies first)
        L = ctypes.CDLL(os.path.join(_HERE, "_xnn", "libxnn_witness.so"))
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.xw_init.restype = ci
        L.xw_new.restype = vp
        L.xw_new.argtypes = [ci]
        L.xw_tensor_f32.restype = ci
        L.xw_tensor_f32.argtypes = [vp, ci, vp, vp, ci, ci]
        L.xw_tensor_q.restype = ci
        L.xw_tensor_q.argtypes = [vp, ci, ci, cf, ci, vp, vp, ci, ci]
        L.xw_tensor_qc.restype = ci
        L.xw_tensor_qc.argtypes = [vp, ci, vp, ci, ci, vp, vp]
        L.xw_op_conv_q8.restype = ci
        L.xw_op_conv_q8.argtypes = [ci] * 10 + [cf, vp, ci, vp, vp, ci, cf, vp, vp, vp]
        L.xw_op_conv_f32.restype = ci
        L.xw_op_conv_f32.argtypes = [ci] * 8 + [vp] * 5
        L.xw_op_deconv_q8.restype = ci
        L.xw_op_deconv_q8.argtypes = [ci] * 6 + [cf, cf, vp, vp, ci, cf, vp, vp, vp]
        L.xw_op_deconv_f32.restype = ci
        L.xw_op_deconv_f32.argtypes = [ci] * 5 + [vp] * 5
        L.xw_leaky_relu.argtypes = [vp, cf, ci, ci]
        L.xw_leaky_relu.restype = None
        L.xw_convert.argtypes = [vp, ci, ci]
        L.xw_convert.restype = None
        L.xw_add.argtypes = [vp, ci, ci, ci]
        L.xw_add.restype = None
        L.xw_status.argtypes = [vp]
        L.xw_status.restype = ci
        L.xw_run.argtypes = [vp, ci, vp, vp]
        L.xw_run.restype = ci
        L.xw_free.argtypes = [vp]
        L.xw_free.restype = None
        assert L.xw_init() == 0, "xnn_initialize failed"
        _LIB = L
    return _LIB


class Graph:
    """One xnn_subgraph + runtime.  Values are addressed by the ids the define calls return."""

    def __init__(self, n_external=8):
        self.L = lib()
        self.g = self.L.xw_new(n_external)
        assert self.g
        self.keep = []          # static data / scale arrays must outlive the runtime
        self.ext = {}           # ext id -> (shape, dtype, io)
        self.next_ext = 0

    g = None

    def close(self):
        if self.g:
            self.L.xw_free(self.g)
            self.g = None

    def __del__(self):
        self.close()

    def _dims(self, shape):
        a = np.asarray(shape, np.int64)
        self.keep.append(a)
        return a.ctypes.data

    def _static(self, data, dtype):
        if data is None:
            return None
        a = np.ascontiguousarray(data, dtype)
        pad = np.zeros(a.nbytes + XNN_EXTRA_BYTES, np.uint8)
        pad[:a.nbytes] = a.view(np.uint8).ravel()
        self.keep.append(pad)
        return pad.ctypes.data

    def _ext(self, shape, dtype, io):
        if not io:
            return -1
        e = self.next_ext
        self.next_ext += 1
        self.ext[e] = (tuple(int(s) for s in shape), np.dtype(dtype), io)
        return e

    # -- values -------------------------------------------------------------------------------------------------------
    def f32(self, shape, data=None, io=0):
        e = self._ext(shape, np.float32, io)
        vid = self.L.xw_tensor_f32(self.g, len(shape), self._dims(shape), self._static(data, np.float32), e, io)
        return (vid, e)

    def q8(self, shape, scale, zero_point, data=None, io=0):
        e = self._ext(shape, np.int8, io)
        vid = self.L.xw_tensor_q(self.g, 8, int(zero_point), float(scale), len(shape), self._dims(shape),
                                 self._static(data, np.int8), e, io)
        return (vid, e)

    def q32(self, shape, scale, data):
        vid = self.L.xw_tensor_q(self.g, 32, 0, float(scale), len(shape), self._dims(shape), self._static(data, np.int32),
                                 -1, 0)
        return (vid, -1)

    def qc(self, bits, shape, scales, channel_dim, data):
        s = np.ascontiguousarray(scales, np.float32)
        self.keep.append(s)
        vid = self.L.xw_tensor_qc(self.g, bits, s.ctypes.data, len(shape), channel_dim, self._dims(shape),
                                  self._static(data, np.int8 if bits == 8 else np.int32))
        return (vid, -1)

    def value_like(self, t, io=0, static=True):
        """Define a value with the flatbuffer tensor's shape / dtype / quantisation (what the delegate does)."""
        shape = tuple(int(s) for s in t.shape)
        data = t.data if static else None
        if t.dtype == np.float32:
            return self.f32(shape, data, io)
        if t.dtype == np.int8:
            if t.scale is not None and len(t.scale) > 1:
                cd = int(t.qdim)
                return self.qc(8, shape, t.scale, cd, data)
            return self.q8(shape, t.scale[0], t.zero_point[0], data, io)
        if t.dtype == np.int32:
            if len(t.scale) > 1:
                return self.qc(32, shape, t.scale, 0, data)
            return self.q32(shape, t.scale[0], data)
        raise NotImplementedError(t.dtype)

    # -- nodes --------------------------------------------------------------------------------------------------------
    def leaky_relu(self, x, y, alpha):
        self.L.xw_leaky_relu(self.g, float(alpha), x[0], y[0])

    def convert(self, x, y):
        self.L.xw_convert(self.g, x[0], y[0])

    def add(self, a, b, y):
        self.L.xw_add(self.g, a[0], b[0], y[0])

    def ok(self):
        return self.L.xw_status(self.g) == 0

    # -- run ----------------------------------------------------------------------------------------------------------
    def run(self, inputs):
        """inputs: {value: array}; returns {ext id: array} for every external output."""
        assert self.ok(), "subgraph definition failed: xnn_status %d" % self.L.xw_status(self.g)
        bufs = {}
        for e, (shape, dt, io) in self.ext.items():
            n = int(np.prod(shape)) * dt.itemsize
            buf = np.zeros(n + XNN_EXTRA_BYTES, np.uint8)
            bufs[e] = buf
        for v, arr in inputs.items():
            e = v[1]
            shape, dt, io = self.ext[e]
            a = np.ascontiguousarray(arr, dt).reshape(shape)
            bufs[e][:a.nbytes] = a.view(np.uint8).ravel()
        ids = np.array(sorted(bufs), np.int32)
        ptrs = (ctypes.c_void_p * len(ids))(*[bufs[int(e)].ctypes.data for e in ids])
        rc = self.L.xw_run(self.g, len(ids), ids.ctypes.data, ctypes.addressof(ptrs))
        assert rc == 0, "xnn runtime failed: %d" % rc
        out = {}
        for e, (shape, dt, io) in self.ext.items():
            if io == 2:
                n = int(np.prod(shape)) * dt.itemsize
                out[e] = bufs[e][:n].view(dt).reshape(shape).copy()
        return out


def _padded(a, dtype):
    a = np.ascontiguousarray(a, dtype)
    buf = np.zeros(a.nbytes + XNN_EXTRA_BYTES, np.uint8)
    buf[:a.nbytes] = a.view(np.uint8).ravel()
    return buf


def conv(x, w, b, *, kind, stride=1, dil=1, transpose=False, in_q=None, w_scale=None, out_q=None, per_tensor=False):
    """One convolution through XNNPACK's operator API (the functions a subgraph node creates).

    x [H, Cin]; w as in the flatbuffer (CONV_2D [Cout, KH, 1, Cin/g]; DEPTHWISE [1, KH, 1, C]; TRANSPOSE_CONV
    [Cout, KH, 1, Cin]); b [Cout] or None.  fp32 when in_q is None, else int8 with in_q / out_q = (scale, zero_point) and
    w_scale = per-channel (or single) filter scales.  kind: "conv" | "dw".  Returns [Hout, Cout], or None if this
    XNNPACK refuses to create the operator."""
    L = lib()
    H, cin = x.shape
    is_q = in_q is not None
    kh = int(w.shape[1])
    if transpose:
        cout = int(w.shape[0])
        hout = (H - 1) * stride + kh
    elif kind == "dw":
        cout = int(w.shape[3])
        hout = H - (kh - 1) * dil
    else:
        cout = int(w.shape[0])
        hout = (H - (kh - 1) * dil - 1) // stride + 1
    dt = np.int8 if is_q else np.float32
    xin = _padded(x, dt)
    out = np.zeros(hout * cout * np.dtype(dt).itemsize + XNN_EXTRA_BYTES, np.uint8)
    wk = _padded(w, dt)
    bk = None if b is None else _padded(b, np.int32 if is_q else np.float32)
    bptr = None if bk is None else bk.ctypes.data
    ho = ctypes.c_int(0)
    if kind == "dw":
        groups, gic, goc = cin, 1, 1
    elif transpose:
        groups, gic, goc = 1, cin, cout
    else:
        gic = int(w.shape[3])
        groups = cin // gic
        goc = cout // groups
    if transpose:
        if is_q:
            ws = np.atleast_1d(np.asarray(w_scale, np.float32))
            assert ws.size == 1, "only the per-tensor qs8 deconvolution exists in 2022-era XNNPACK and in these graphs"
            rc = L.xw_op_deconv_q8(H, kh, stride, cin, cout, int(in_q[1]), float(in_q[0]), float(ws[0]), wk.ctypes.data, bptr,
                                   int(out_q[1]), float(out_q[0]), xin.ctypes.data, out.ctypes.data, ctypes.addressof(ho))
        else:
            rc = L.xw_op_deconv_f32(H, kh, stride, cin, cout, wk.ctypes.data, bptr, xin.ctypes.data, out.ctypes.data,
                                    ctypes.addressof(ho))
    elif is_q:
        ws = np.ascontiguousarray(np.atleast_1d(w_scale), np.float32)
        rc = L.xw_op_conv_q8(1 if kind == "dw" else 0, 1 if per_tensor else 0, H, kh, stride, dil, groups, gic, goc, int(in_q[1]),
                             float(in_q[0]), ws.ctypes.data, ws.size, wk.ctypes.data, bptr, int(out_q[1]), float(out_q[0]),
                             xin.ctypes.data, out.ctypes.data, ctypes.addressof(ho))
    else:
        rc = L.xw_op_conv_f32(1 if kind == "dw" else 0, H, kh, stride, dil, groups, gic, goc, wk.ctypes.data, bptr,
                              xin.ctypes.data, out.ctypes.data, ctypes.addressof(ho))
    if rc != 0:
        return None
    assert ho.value == hout, (ho.value, hout)
    return out[:hout * cout * np.dtype(dt).itemsize].view(dt).reshape(hout, cout).copy()


def run_op(sg, op, traced):
    """Run ONE operator of a flatbuffer subgraph through XNNPACK on traced operand values.

    sg: tflite_reader subgraph; op: one of its operators; traced: {tensor index: value}.  Returns the output array, or
    None if this XNNPACK refuses the operator (the delegate would leave it to the TFLite builtin kernel)."""
    T = sg.tensors
    name = op.name
    o = op.opts or {}
    out_t = T[op.outputs[0]]
    if name in ("CONV_2D", "DEPTHWISE_CONV_2D", "TRANSPOSE_CONV"):
        tc = name == "TRANSPOSE_CONV"
        xi, wi, bi = (op.inputs[2], op.inputs[1], op.inputs[3]) if tc else (op.inputs[0], op.inputs[1], op.inputs[2])
        xt, wt = T[xi], T[wi]
        x = traced[xi]
        x2 = x.reshape(x.shape[1], x.shape[3])
        b = traced[bi] if bi >= 0 else None
        q = xt.dtype == np.int8
        y = conv(x2, traced[wi], b, kind="dw" if name == "DEPTHWISE_CONV_2D" else "conv", stride=o["stride_h"],
                 dil=o.get("dil_h", 1), transpose=tc,
                 in_q=(xt.scale[0], xt.zero_point[0]) if q else None, w_scale=wt.scale if q else None,
                 out_q=(out_t.scale[0], out_t.zero_point[0]) if q else None)
        return None if y is None else y.reshape(tuple(int(v) for v in out_t.shape))
    G = Graph()
    try:
        if name in ("LEAKY_RELU", "QUANTIZE", "DEQUANTIZE"):
            x = G.value_like(T[op.inputs[0]], io=1, static=False)
            y = G.value_like(out_t, io=2, static=False)
            if name == "LEAKY_RELU":
                G.leaky_relu(x, y, o["alpha"])
            else:
                G.convert(x, y)
            feeds = {x: traced[op.inputs[0]]}
        elif name == "ADD":
            a = G.value_like(T[op.inputs[0]], io=1, static=False)
            b = G.value_like(T[op.inputs[1]], io=1, static=False)
            y = G.value_like(out_t, io=2, static=False)
            G.add(a, b, y)
            feeds = {a: traced[op.inputs[0]], b: traced[op.inputs[1]]}
        else:
            raise NotImplementedError(name)
        if not G.ok():
            return None
        out = G.run(feeds)
        return next(iter(out.values()))
    finally:
        G.close()

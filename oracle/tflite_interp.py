"""Generic numpy interpreter for the three Lyra TFLite graphs.

TEST INFRASTRUCTURE ONLY.  This is the closest thing to "the reference run
here": it executes the reference's own flatbuffers
(/root/reference/lyra/model_coeffs/*.tflite) operator by operator with TFLite
builtin-kernel semantics, without any knowledge of the
network architecture.  It exists to (1) generate the golden vectors committed
under tests/golden/ (tools/make_golden.py) and (2) cross-check the
hand-restated C oracle (oracle/lyra_oracle.c), which is architecture-specific
and fast.  It cannot travel to the GPU box (needs /root/reference) and is far
too slow to be a baseline.

TFLite itself (v2.11.0, WORKSPACE:168-174 of the reference) is an un-vendored
third-party dependency and cannot be built offline, so the op semantics here
are a restatement of its published reference kernels:
  * float conv / depthwise / transpose-conv: plain fp32 (numpy matmul order;
    `acc64=True` accumulates in float64 instead -- SURVEY.md 8(c) measured that
    the float summation order flips no int8 code on real speech);
  * int8 conv / depthwise / transpose-conv: int32 accumulation, per-channel
    requantisation in one of two modes:
      "exact"           single rounding of the exact Q31 product
                        (ruy / XNNPACK behaviour; default),
      "gemmlowp_double" SaturatingRoundingDoublingHighMul + RoundingDivideByPOT
                        (TFLite reference-kernel behaviour);
  * int8 LEAKY_RELU and ADD use the gemmlowp double-rounding
    MultiplyByQuantizedMultiplier in those two modes (their builtin kernels
    have no other path).

Third mode, requant="xnnpack" (round 4): the arithmetic of XNNPACK's QS8 / F32
operators, which is what the reference actually runs for these two graphs
(use_xnn=true: soundstream_encoder.cc:39-40, lyra_gan_model.cc:39-40,
tflite_model_wrapper.cc:63-85).  Every formula below is held against REAL
XNNPACK code run here (the one torch's libtorch_cpu.so exports;
oracle/xnn_witness.c, tests/test_xnnpack_witness.py) -- 0 differing outputs:
  * int8 conv / depthwise / transpose-conv: fp32 requantisation
    q = RNE(min(max(float(acc) * scale_c, -128 - z), 127 - z)) + z with
    scale_c = (s_in * s_w[c]) / s_out evaluated in fp32;
  * int8 LEAKY_RELU: Q8 multipliers, (v * m + (z_out << 8) + 0x80) >> 8 with
    m = lrintf(256 * s_in / s_out) for v >= 0 and lrintf(256 * (s_in / s_out *
    alpha)) otherwise;
  * int8 ADD: integer multipliers a * ma + b * mb + bias, arithmetic shift;
  * QUANTIZE: RNE(x * (1 / s)) (reciprocal multiply, ties to even);
  * fp32 conv / depthwise / transpose-conv: acc starts from the BIAS, then one
    fused multiply-add per (tap ascending, input channel ascending) -- exactly
    what XNNPACK's GEMM / IGEMM / DWCONV micro-kernels compute on an FMA target
    (fp32="chain", oracle/chain_f32.c; for a transpose-conv, taps ascending is
    input rows newest first).
fp32="chain" is available in every mode (and is what tools/make_golden.py
uses since round 4): the canonical fp32 order of oracle/lyra_oracle.c.
"""
import ctypes
import math
import os

import numpy as np

from . import tflite_reader as tr

INT32_MIN = -(1 << 31)
INT32_MAX = (1 << 31) - 1


# ----------------------------------------------------------------------------
# fixed-point helpers (vectorised, int64 numpy)
# ----------------------------------------------------------------------------
def quantize_multiplier(m):
    """TFLite QuantizeMultiplier(double) -> (int32 M, int shift)."""
    m = float(m)
    if m == 0.0:
        return 0, 0
    q, shift = math.frexp(m)
    # TFLite uses std::round (half away from zero)
    qf = q * (1 << 31)
    M = int(math.floor(qf + 0.5)) if qf >= 0 else -int(math.floor(-qf + 0.5))
    if M == (1 << 31):
        M //= 2
        shift += 1
    if shift < -31:
        return 0, 0
    return M, shift


def srdhm(a, b):
    """SaturatingRoundingDoublingHighMul on int64 arrays holding int32 values."""
    a = np.asarray(a, np.int64)
    b = np.asarray(b, np.int64)
    ab = a * b
    nudge = np.where(ab >= 0, 1 << 30, 1 - (1 << 30))
    t = ab + nudge
    # C++ '/' truncation toward zero
    res = np.where(t >= 0, t >> 31, -((-t) >> 31))
    sat = (a == INT32_MIN) & (b == INT32_MIN)
    return np.where(sat, INT32_MAX, res)


def rdivpot(x, e):
    x = np.asarray(x, np.int64)
    e = np.asarray(e, np.int64)
    mask = (np.int64(1) << e) - 1
    rem = x & mask
    thr = (mask >> 1) + (x < 0)
    return (x >> e) + (rem > thr)


def mbqm_double(x, M, shift):
    """MultiplyByQuantizedMultiplier, gemmlowp double rounding."""
    x = np.asarray(x, np.int64)
    shift = np.asarray(shift, np.int64)
    left = np.maximum(shift, 0)
    right = np.maximum(-shift, 0)
    return rdivpot(srdhm(x * (np.int64(1) << left), M), right)


def mbqm_exact(x, M, shift):
    """Single rounding (round-half-up) of the exact Q31 product."""
    x = np.asarray(x, np.int64)
    shift = np.asarray(shift, np.int64)
    total = 31 - shift
    return (x * np.asarray(M, np.int64) + (np.int64(1) << (total - 1))) >> total


def round_half_away(x):
    return np.where(x >= 0, np.floor(x + 0.5), -np.floor(-x + 0.5))


# ----------------------------------------------------------------------------
# XNNPACK QS8 arithmetic (mode "xnnpack"); float32 arithmetic spelled out
# ----------------------------------------------------------------------------
F32 = np.float32


def xnn_requant(acc, s_in, w_scales, s_out, z_out):
    """qs8 / qs8_qc8w conv, dwconv, deconv: fp32 requantisation.  acc int64 [..., C]."""
    scale = ((F32(s_in) * np.asarray(w_scales, F32)).astype(F32) / F32(s_out)).astype(F32)
    v = (acc.astype(F32) * scale).astype(F32)
    v = np.minimum(np.maximum(v, F32(-128 - z_out)), F32(127 - z_out))
    return (np.rint(v).astype(np.int64) + z_out).astype(np.int8)      # np.rint: ties to even


def xnn_lrelu(x, s_in, z_in, s_out, z_out, alpha):
    pos = F32(s_in) / F32(s_out)
    neg = F32(pos * F32(alpha))
    mp = int(np.rint(F32(256.0) * pos))
    mn = int(np.rint(F32(256.0) * neg))
    v = x.astype(np.int64) - z_in
    acc = (z_out << 8) + 0x80 + v * np.where(v >= 0, mp, mn)
    return np.clip(acc >> 8, -128, 127).astype(np.int8)


def xnn_add(a, b, s1, z1, s2, z2, so, zo):
    ao = F32(s1) / F32(so)
    bo = F32(s2) / F32(so)
    mx = max(abs(ao), abs(bo))
    shift = 20 - ((int(np.array(mx, F32).view(np.uint32)) >> 23) - 127)

    def mult(x):
        return int(np.rint(np.array(np.array(x, F32).view(np.uint32) + np.uint32(shift << 23)).view(F32)))

    ma, mb = mult(ao), mult(bo)
    bias = (1 << (shift - 1)) - ma * z1 - mb * z2
    acc = bias + a.astype(np.int64) * ma + b.astype(np.int64) * mb
    return (np.clip(acc >> shift, -128 - zo, 127 - zo) + zo).astype(np.int8)


def xnn_quantize(x, s, z):
    inv = F32(1.0) / F32(s)
    v = (x.astype(F32) * inv).astype(F32)
    v = np.minimum(np.maximum(v, F32(-128 - z)), F32(127 - z))
    return (np.rint(v).astype(np.int64) + z).astype(np.int8)


# ----------------------------------------------------------------------------
# fp32 layers as explicit fmaf chains (oracle/chain_f32.c)
# ----------------------------------------------------------------------------
_CHAIN = None


def chain_lib():
    global _CHAIN
    if _CHAIN is None:
        L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libchain_f32.so"))
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.cf_conv.argtypes = [ci] * 8 + [vp] * 4
        L.cf_dwconv.argtypes = [ci] * 5 + [vp] * 4
        L.cf_deconv.argtypes = [ci] * 7 + [vp] * 4
        L.cf_deconv_c4.argtypes = [ci] * 5 + [vp] * 4
        for f in (L.cf_conv, L.cf_dwconv, L.cf_deconv, L.cf_deconv_c4):
            f.restype = None
        _CHAIN = L
    return _CHAIN


def _p(a):
    return None if a is None else a.ctypes.data


def chain_conv(x, w, b, stride, dil, bias_first=True):
    """x [H, Cin], w [Cout, KH, 1, Cin/g], b [Cout] -> [Hout, Cout]"""
    x = np.ascontiguousarray(x, F32); w = np.ascontiguousarray(w, F32); b = np.ascontiguousarray(b, F32)
    H, cin = x.shape
    cout, kh, _, gic = w.shape
    g = cin // gic
    hout = (H - (kh - 1) * dil - 1) // stride + 1
    y = np.zeros((hout, cout), F32)
    chain_lib().cf_conv(int(bias_first), H, kh, stride, dil, g, gic, cout // g, _p(w), _p(b), _p(x), _p(y))
    return y


def chain_dwconv(x, w, b, dil, bias_first=True):
    """x [H, C], w [1, KH, 1, C]"""
    x = np.ascontiguousarray(x, F32); w = np.ascontiguousarray(w, F32); b = np.ascontiguousarray(b, F32)
    H, c = x.shape
    kh = w.shape[1]
    y = np.zeros((H - (kh - 1) * dil, c), F32)
    chain_lib().cf_dwconv(int(bias_first), H, kh, dil, c, _p(w), _p(b), _p(x), _p(y))
    return y


def chain_deconv(x, w, b, stride, bias_first=True, tap_order=0):
    """x [H, Cin], w [Cout, KH, 1, Cin]"""
    x = np.ascontiguousarray(x, F32); w = np.ascontiguousarray(w, F32)
    b = None if b is None else np.ascontiguousarray(b, F32)
    H, cin = x.shape
    cout, kh = w.shape[0], w.shape[1]
    y = np.zeros(((H - 1) * stride + kh, cout), F32)
    chain_lib().cf_deconv(int(bias_first), tap_order, H, kh, stride, cin, cout, _p(w), _p(b), _p(x), _p(y))
    return y


def chain_deconv_c4(x, w, b, stride, fused=False):
    """cout == 1 transposed conv as XNNPACK's x86 nr2 kernel (4x2c4 SSE) sums it; see oracle/chain_f32.c"""
    x = np.ascontiguousarray(x, F32); w = np.ascontiguousarray(w, F32)
    b = None if b is None else np.ascontiguousarray(b, F32)
    H, cin = x.shape
    assert w.shape[0] == 1
    kh = w.shape[1]
    y = np.zeros(((H - 1) * stride + kh, 1), F32)
    chain_lib().cf_deconv_c4(int(fused), H, kh, stride, cin, _p(w), _p(b), _p(x), _p(y))
    return y


# ----------------------------------------------------------------------------
class Interpreter:
    def __init__(self, path, requant="exact", acc64=False, fp32=None):
        assert requant in ("exact", "gemmlowp_double", "xnnpack", "builtin_mixed")
        self.model = tr.load(path)
        self.requant = requant
        self.acc64 = acc64
        # fp32 layer evaluation: "numpy" (a @ b, or float64 accumulation with acc64=True) or "chain" (bias-first fmaf
        # chains, oracle/chain_f32.c -- XNNPACK's order).  Mode "xnnpack" implies "chain".
        self.fp32 = fp32 if fp32 is not None else ("chain" if requant in ("xnnpack", "builtin_mixed") else "numpy")
        assert self.fp32 in ("numpy", "chain")
        self.vars = {}
        self.called_once = set()
        self.trace = None  # optional dict: tensor index -> value (subgraph 0 only)

    # -- variables ---------------------------------------------------------
    def reset_variables(self):
        self.vars = {}
        self.called_once = set()

    # -- running -----------------------------------------------------------
    def run_signature(self, key, inputs):
        sig = self.model.signatures[key]
        sg = self.model.subgraphs[sig["subgraph"]]
        feed = {sig["inputs"][k]: v for k, v in inputs.items()}
        vals = self.run_subgraph(sig["subgraph"], feed)
        return {k: vals[i] for k, i in sig["outputs"].items()}

    def run_subgraph(self, sgi, feed):
        sg = self.model.subgraphs[sgi]
        vals = {}
        for t in sg.tensors:
            if t.data is not None:
                vals[t.index] = t.data
        for k, v in feed.items():
            t = sg.tensors[k]
            vals[k] = np.asarray(v, t.dtype).reshape(t.shape)
        handles = {}
        for op in sg.ops:
            self._exec(sg, op, vals, handles)
        if self.trace is not None and sgi == 0:
            self.trace.update(vals)
        return vals

    # -- op dispatch -------------------------------------------------------
    def _exec(self, sg, op, vals, handles):
        name = op.name
        T = sg.tensors
        ins = [vals[i] if i >= 0 else None for i in op.inputs]
        out_t = [T[i] for i in op.outputs]

        if name == "CALL_ONCE":
            idx = op.opts["init_subgraph_index"]
            if idx not in self.called_once:
                self.called_once.add(idx)
                self.run_subgraph(idx, {})
            return
        if name == "VAR_HANDLE":
            vals[op.outputs[0]] = ("var", op.opts["shared_name"])
            return
        if name == "READ_VARIABLE":
            vals[op.outputs[0]] = self.vars[ins[0][1]]
            return
        if name == "ASSIGN_VARIABLE":
            self.vars[ins[0][1]] = np.array(ins[1], copy=True)
            return

        if name == "RESHAPE":
            r = ins[0].reshape(tuple(int(x) for x in ins[1]))
        elif name == "CONCATENATION":
            r = np.concatenate(ins, axis=op.opts["axis"])
        elif name == "STRIDED_SLICE":
            r = self._strided_slice(ins, op.opts)
        elif name == "SPLIT":
            axis = int(ins[0])
            parts = np.split(ins[1], op.opts["num_splits"], axis=axis)
            for o, p in zip(op.outputs, parts):
                vals[o] = p
            return
        elif name == "CONV_2D":
            r = self._conv(T[op.inputs[0]], T[op.inputs[1]], T[op.inputs[2]], out_t[0], ins, op.opts)
        elif name == "DEPTHWISE_CONV_2D":
            r = self._dwconv(T[op.inputs[0]], T[op.inputs[1]], T[op.inputs[2]], out_t[0], ins, op.opts)
        elif name == "TRANSPOSE_CONV":
            r = self._tconv(sg, op, ins, out_t[0])
        elif name == "LEAKY_RELU":
            r = self._lrelu(T[op.inputs[0]], out_t[0], ins[0], op.opts["alpha"])
        elif name == "ADD":
            r = self._add(T[op.inputs[0]], T[op.inputs[1]], out_t[0], ins)
        elif name == "SUB":
            assert ins[0].dtype != np.int8
            r = ins[0] - ins[1]
        elif name == "MUL":
            assert ins[0].dtype != np.int8
            r = ins[0] * ins[1]
        elif name == "QUANTIZE":
            r = self._quantize(T[op.inputs[0]], out_t[0], ins[0])
        elif name == "DEQUANTIZE":
            t = T[op.inputs[0]]
            s = np.float64(t.scale[0])
            r = (s * (ins[0].astype(np.int64) - int(t.zero_point[0]))).astype(np.float32)
        elif name == "GATHER":
            assert op.opts["batch_dims"] == 0
            r = np.take(ins[0], ins[1], axis=op.opts["axis"])
        elif name == "SQUARED_DIFFERENCE":
            d = ins[0] - ins[1]
            r = d * d
        elif name == "SUM":
            axes = tuple(int(a) for a in np.atleast_1d(ins[1]))
            # sequential fp32 sum over the reduced axis (TFLite reference reducer order)
            assert len(axes) == 1
            ax = axes[0] % ins[0].ndim
            x = np.moveaxis(ins[0], ax, -1)
            acc = np.zeros(x.shape[:-1], np.float32)
            for i in range(x.shape[-1]):
                acc = acc + x[..., i]
            r = acc
            if op.opts["keep_dims"]:
                r = np.expand_dims(r, ax)
        elif name == "ARG_MIN":
            r = np.argmin(ins[0], axis=int(ins[1])).astype(out_t[0].dtype)
        elif name == "ARG_MAX":
            r = np.argmax(ins[0], axis=int(ins[1])).astype(out_t[0].dtype)
        elif name == "ONE_HOT":
            idx, depth, on, off = ins
            depth = int(depth)
            axis = op.opts["axis"]
            assert axis == -1
            r = np.where(np.arange(depth) == np.expand_dims(idx, -1), on, off).astype(out_t[0].dtype)
        elif name == "PACK":
            r = np.stack(ins, axis=op.opts["axis"])
        elif name == "CAST":
            r = ins[0].astype(out_t[0].dtype)
        elif name == "LESS":
            r = ins[0] < ins[1]
        elif name == "NOT_EQUAL":
            r = ins[0] != ins[1]
        elif name == "MAXIMUM":
            r = np.maximum(ins[0], ins[1])
        else:
            raise NotImplementedError(name)

        ot = out_t[0]
        if ot.dtype is not None and r.dtype != ot.dtype:
            r = r.astype(ot.dtype)
        vals[op.outputs[0]] = r

    # -- helpers -----------------------------------------------------------
    @staticmethod
    def _strided_slice(ins, o):
        x, begin, end, strides = ins
        assert o["ellipsis_mask"] == 0 and o["new_axis_mask"] == 0
        sl = []
        for d in range(len(begin)):
            st = int(strides[d])
            b = None if (o["begin_mask"] >> d) & 1 else int(begin[d])
            e = None if (o["end_mask"] >> d) & 1 else int(end[d])
            if (o["shrink_axis_mask"] >> d) & 1:
                bb = int(begin[d])
                if bb < 0:
                    bb += x.shape[d]
                sl.append(bb)
            else:
                sl.append(slice(b, e, st))
        return x[tuple(sl)]

    def _mm(self, a, b):
        """fp32 matmul a[M,K] @ b[K,N]."""
        if self.acc64:
            return (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
        return a @ b

    def _requant(self, acc, in_t, w_t, out_t, cout_axis_len, ungrouped_conv=False):
        """acc: int64 [..., Cout] -> int8.  Mode "builtin_mixed" (round 6) is the per-operator mixture the graphs compute if
        TFLite's XNNPACK delegate takes the fp32 operators but NOT the signed-int8 ones (the reference ORs in only the QU8
        delegate flag, tflite_model_wrapper.cc:65-67): the builtin int8 kernels of TFLite 2.11 as recalled in DESIGN.md 2 --
        an ungrouped CONV_2D goes through the optimized path (ruy: single rounding of the Q31 product = "exact"), a grouped
        CONV_2D, DEPTHWISE_CONV_2D and TRANSPOSE_CONV through the reference-kernel MultiplyByQuantizedMultiplier
        (gemmlowp double rounding); LEAKY_RELU / ADD / QUANTIZE are the builtin forms of "exact" / "gemmlowp_double"."""
        s_in = np.float64(in_t.scale[0])
        s_out = np.float64(out_t.scale[0])
        if self.requant == "xnnpack":
            wsc = w_t.scale.astype(np.float32)
            if len(wsc) == 1:
                wsc = np.repeat(wsc, cout_axis_len)
            return xnn_requant(acc, in_t.scale[0], wsc, out_t.scale[0], int(out_t.zero_point[0]))
        ws = w_t.scale.astype(np.float64)
        if len(ws) == 1:
            ws = np.repeat(ws, cout_axis_len)
        Ms, Ss = zip(*[quantize_multiplier(s_in * w / s_out) for w in ws])
        Ms = np.array(Ms, np.int64)
        Ss = np.array(Ss, np.int64)
        if self.requant == "exact" or (self.requant == "builtin_mixed" and ungrouped_conv):
            y = mbqm_exact(acc, Ms, Ss)
        else:
            y = mbqm_double(acc, Ms, Ss)
        y = y + int(out_t.zero_point[0])
        return np.clip(y, -128, 127).astype(np.int8)

    def _conv(self, in_t, w_t, b_t, out_t, ins, o):
        x, w, b = ins  # x [1,H,1,Cin]; w [Cout,K,1,Cin/g]
        assert o["padding"] == 1 and o["act"] == 0 and o["dil_h"] == 1
        stride = o["stride_h"]
        N, H, W, Cin = x.shape
        Cout, K, _, Cig = w.shape
        g = Cin // Cig
        Cog = Cout // g
        Hout = (H - K) // stride + 1
        is_q = x.dtype == np.int8
        if not is_q and self.fp32 == "chain":
            return chain_conv(x.reshape(H, Cin), w, b, stride, 1).reshape(1, Hout, 1, Cout)
        if is_q:
            xx = x.astype(np.int64) - int(in_t.zero_point[0])
            ww = w.astype(np.int64)
        else:
            xx, ww = x, w
        xx = xx.reshape(H, Cin)
        out = np.zeros((Hout, Cout), np.int64 if is_q else np.float32)
        for gi in range(g):
            xg = xx[:, gi * Cig:(gi + 1) * Cig]
            # patches [Hout, K*Cig] (tap-major, channel-inner)
            idx = (np.arange(Hout) * stride)[:, None] + np.arange(K)[None, :]
            patches = xg[idx].reshape(Hout, K * Cig)
            wg = ww[gi * Cog:(gi + 1) * Cog].reshape(Cog, K * Cig).T
            out[:, gi * Cog:(gi + 1) * Cog] = (patches @ wg) if is_q else self._mm(patches, wg)
        if is_q:
            out = out + b.astype(np.int64)[None, :]
            r = self._requant(out, in_t, w_t, out_t, Cout, ungrouped_conv=(g == 1))
        else:
            r = out + b[None, :]
        return r.reshape(1, Hout, 1, Cout)

    def _dwconv(self, in_t, w_t, b_t, out_t, ins, o):
        x, w, b = ins  # w [1,K,1,C]
        assert o["padding"] == 1 and o["act"] == 0 and o["stride_h"] == 1 and o["depth_mult"] == 1
        dil = o["dil_h"]
        N, H, W, C = x.shape
        K = w.shape[1]
        Hout = H - (K - 1) * dil
        is_q = x.dtype == np.int8
        if not is_q and self.fp32 == "chain":
            return chain_dwconv(x.reshape(H, C), w, b, dil).reshape(1, Hout, 1, C)
        if is_q:
            xx = (x.astype(np.int64) - int(in_t.zero_point[0])).reshape(H, C)
            ww = w.astype(np.int64).reshape(K, C)
            acc = np.zeros((Hout, C), np.int64)
        else:
            xx = x.reshape(H, C)
            ww = w.reshape(K, C)
            acc = np.zeros((Hout, C), np.float32)
        for k in range(K):
            acc = acc + xx[k * dil:k * dil + Hout] * ww[k][None, :]
        if is_q:
            acc = acc + b.astype(np.int64)[None, :]
            r = self._requant(acc, in_t, w_t, out_t, C)
        else:
            r = acc + b[None, :]
        return r.reshape(1, Hout, 1, C)

    def _tconv(self, sg, op, ins, out_t):
        oshape, w, x, b = ins  # w [Cout,K,1,Cin]
        T = sg.tensors
        in_t, w_t = T[op.inputs[2]], T[op.inputs[1]]
        o = op.opts
        assert o["padding"] == 1
        stride = o["stride_h"]
        N, H, W, Cin = x.shape
        Cout, K, _, _ = w.shape
        Hout = (H - 1) * stride + K
        assert tuple(int(v) for v in oshape) == (1, Hout, 1, Cout), (oshape, Hout, Cout)
        is_q = x.dtype == np.int8
        if not is_q and self.fp32 == "chain":
            return chain_deconv(x.reshape(H, Cin), w, b, stride).reshape(1, Hout, 1, Cout)
        if is_q:
            xx = (x.astype(np.int64) - int(in_t.zero_point[0])).reshape(H, Cin)
            ww = w.astype(np.int64)
            out = np.zeros((Hout, Cout), np.int64)
        else:
            xx = x.reshape(H, Cin)
            ww = w
            out = np.zeros((Hout, Cout), np.float32)
        wk = ww.reshape(Cout, K, Cin)
        for t in range(H):
            for k in range(K):
                contrib = (xx[t][None, :] @ wk[:, k, :].T) if is_q else self._mm(xx[t][None, :], wk[:, k, :].T)
                out[t * stride + k] += contrib[0]
        if is_q:
            if b is not None:
                out = out + b.astype(np.int64)[None, :]
            r = self._requant(out, in_t, w_t, out_t, Cout)
        else:
            r = out + (b[None, :] if b is not None else 0)
        return r.reshape(1, Hout, 1, Cout)

    def _lrelu(self, in_t, out_t, x, alpha):
        if x.dtype != np.int8:
            return np.where(x > 0, x, x * np.float32(alpha)).astype(np.float32)
        if self.requant == "xnnpack":
            return xnn_lrelu(x, in_t.scale[0], int(in_t.zero_point[0]), out_t.scale[0], int(out_t.zero_point[0]), alpha)
        s_in = np.float64(in_t.scale[0])
        s_out = np.float64(out_t.scale[0])
        Mi, si = quantize_multiplier(s_in / s_out)
        Ma, sa = quantize_multiplier(s_in * np.float64(np.float32(alpha)) / s_out)
        v = x.astype(np.int64) - int(in_t.zero_point[0])
        y = np.where(v >= 0, mbqm_double(v, Mi, si), mbqm_double(v, Ma, sa)) + int(out_t.zero_point[0])
        return np.clip(y, -128, 127).astype(np.int8)

    def _add(self, t1, t2, out_t, ins):
        a, b = ins
        if a.dtype != np.int8:
            return a + b
        if self.requant == "xnnpack":
            return xnn_add(a, b, t1.scale[0], int(t1.zero_point[0]), t2.scale[0], int(t2.zero_point[0]), out_t.scale[0],
                           int(out_t.zero_point[0]))
        s1, s2, so = np.float64(t1.scale[0]), np.float64(t2.scale[0]), np.float64(out_t.scale[0])
        twice = 2.0 * max(s1, s2)
        M1, h1 = quantize_multiplier(s1 / twice)
        M2, h2 = quantize_multiplier(s2 / twice)
        Mo, ho = quantize_multiplier(twice / ((1 << 20) * so))
        va = (a.astype(np.int64) - int(t1.zero_point[0])) * (1 << 20)
        vb = (b.astype(np.int64) - int(t2.zero_point[0])) * (1 << 20)
        sa = mbqm_double(va, M1, h1)
        sb = mbqm_double(vb, M2, h2)
        y = mbqm_double(sa + sb, Mo, ho) + int(out_t.zero_point[0])
        return np.clip(y, -128, 127).astype(np.int8)

    def _quantize(self, in_t, out_t, x):
        assert x.dtype == np.float32, "only float->int8 QUANTIZE appears in these graphs"
        if self.requant == "xnnpack":
            return xnn_quantize(x, out_t.scale[0], int(out_t.zero_point[0]))
        s = np.float32(out_t.scale[0])
        q = round_half_away((x / s).astype(np.float32)).astype(np.int64) + int(out_t.zero_point[0])
        return np.clip(q, -128, 127).astype(np.int8)

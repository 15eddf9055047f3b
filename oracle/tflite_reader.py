"""Minimal dependency-free TFLite (schema v3) flatbuffer reader.

TEST INFRASTRUCTURE ONLY (oracle/): used by the numpy graph interpreter
(oracle/tflite_interp.py), by the golden-vector generator and by the weight
packer (tools/pack_weights.py).  Nothing on the product path imports it.

Field ids follow SURVEY.md Appendix B (validated by parsing the three model
files under /root/reference/lyra/model_coeffs/).
"""
import struct
import numpy as np

# TensorType -> numpy dtype
TENSOR_TYPES = {0: np.float32, 2: np.int32, 4: np.int64, 6: np.bool_, 9: np.int8, 13: None}

OPCODES = {
    0: "ADD", 2: "CONCATENATION", 3: "CONV_2D", 4: "DEPTHWISE_CONV_2D", 6: "DEQUANTIZE",
    18: "MUL", 22: "RESHAPE", 36: "GATHER", 41: "SUB", 45: "STRIDED_SLICE", 49: "SPLIT",
    53: "CAST", 55: "MAXIMUM", 56: "ARG_MAX", 58: "LESS", 67: "TRANSPOSE_CONV",
    72: "NOT_EQUAL", 74: "SUM", 79: "ARG_MIN", 83: "PACK", 85: "ONE_HOT", 98: "LEAKY_RELU",
    99: "SQUARED_DIFFERENCE", 114: "QUANTIZE", 129: "CALL_ONCE", 142: "VAR_HANDLE",
    143: "READ_VARIABLE", 144: "ASSIGN_VARIABLE",
}


class _FB:
    def __init__(self, buf):
        self.b = buf

    def u8(self, o): return self.b[o]
    def i8(self, o): return struct.unpack_from("<b", self.b, o)[0]
    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def i32(self, o): return struct.unpack_from("<i", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]
    def f32(self, o): return struct.unpack_from("<f", self.b, o)[0]

    def root(self):
        return self.u32(0)

    def field(self, tbl, fid):
        """Absolute offset of field `fid` in table at `tbl`, or None."""
        vt = tbl - self.i32(tbl)
        vt_size = self.u16(vt)
        slot = 4 + 2 * fid
        if slot >= vt_size:
            return None
        off = self.u16(vt + slot)
        return tbl + off if off else None

    def indirect(self, o):
        return o + self.u32(o)

    def scalar(self, tbl, fid, kind, default=0):
        o = self.field(tbl, fid)
        if o is None:
            return default
        return getattr(self, kind)(o)

    def table(self, tbl, fid):
        o = self.field(tbl, fid)
        return None if o is None else self.indirect(o)

    def vec(self, tbl, fid):
        """(start, length) of a vector field or (None, 0)."""
        o = self.field(tbl, fid)
        if o is None:
            return None, 0
        v = self.indirect(o)
        return v + 4, self.u32(v)

    def vec_np(self, tbl, fid, dtype):
        s, n = self.vec(tbl, fid)
        if s is None:
            return np.zeros(0, dtype)
        return np.frombuffer(self.b, dtype=dtype, count=n, offset=s).copy()

    def vec_tables(self, tbl, fid):
        s, n = self.vec(tbl, fid)
        return [self.indirect(s + 4 * i) for i in range(n)]

    def string(self, tbl, fid):
        o = self.field(tbl, fid)
        if o is None:
            return ""
        v = self.indirect(o)
        n = self.u32(v)
        return bytes(self.b[v + 4:v + 4 + n]).decode()


class Tensor:
    __slots__ = ("index", "name", "shape", "type", "dtype", "buffer", "scale", "zero_point",
                 "qdim", "is_variable", "data")

    def __repr__(self):
        q = ""
        if len(self.scale):
            q = f" q(s={self.scale[:2]}{'...' if len(self.scale) > 2 else ''},z={self.zero_point[:2]})"
        return f"T#{self.index} {self.name} {list(self.shape)} t{self.type}{q}"


class Op:
    __slots__ = ("index", "code", "name", "inputs", "outputs", "opts")

    def __repr__(self):
        return f"Op#{self.index} {self.name} in={self.inputs} out={self.outputs} {self.opts}"


class SubGraph:
    pass


class Model:
    pass


def _parse_options(fb, name, t):
    if t is None:
        return {}
    if name == "CONV_2D":
        return dict(padding=fb.scalar(t, 0, "i8"), stride_w=fb.scalar(t, 1, "i32"),
                    stride_h=fb.scalar(t, 2, "i32"), act=fb.scalar(t, 3, "i8"),
                    dil_w=fb.scalar(t, 4, "i32", 1), dil_h=fb.scalar(t, 5, "i32", 1))
    if name == "DEPTHWISE_CONV_2D":
        return dict(padding=fb.scalar(t, 0, "i8"), stride_w=fb.scalar(t, 1, "i32"),
                    stride_h=fb.scalar(t, 2, "i32"), depth_mult=fb.scalar(t, 3, "i32"),
                    act=fb.scalar(t, 4, "i8"), dil_w=fb.scalar(t, 5, "i32", 1),
                    dil_h=fb.scalar(t, 6, "i32", 1))
    if name == "TRANSPOSE_CONV":
        return dict(padding=fb.scalar(t, 0, "i8"), stride_w=fb.scalar(t, 1, "i32"),
                    stride_h=fb.scalar(t, 2, "i32"))
    if name == "STRIDED_SLICE":
        return dict(begin_mask=fb.scalar(t, 0, "i32"), end_mask=fb.scalar(t, 1, "i32"),
                    ellipsis_mask=fb.scalar(t, 2, "i32"), new_axis_mask=fb.scalar(t, 3, "i32"),
                    shrink_axis_mask=fb.scalar(t, 4, "i32"))
    if name == "LEAKY_RELU":
        return dict(alpha=fb.scalar(t, 0, "f32"))
    if name == "CONCATENATION":
        return dict(axis=fb.scalar(t, 0, "i32"), act=fb.scalar(t, 1, "i8"))
    if name == "SPLIT":
        return dict(num_splits=fb.scalar(t, 0, "i32"))
    if name == "GATHER":
        return dict(axis=fb.scalar(t, 0, "i32"), batch_dims=fb.scalar(t, 1, "i32"))
    if name == "CALL_ONCE":
        return dict(init_subgraph_index=fb.scalar(t, 0, "i32"))
    if name == "VAR_HANDLE":
        return dict(container=fb.string(t, 0), shared_name=fb.string(t, 1))
    if name in ("ARG_MIN", "ARG_MAX"):
        return dict(output_type=fb.scalar(t, 0, "i8"))
    if name == "SUM":
        return dict(keep_dims=bool(fb.scalar(t, 0, "u8")))
    if name == "PACK":
        return dict(values_count=fb.scalar(t, 0, "i32"), axis=fb.scalar(t, 1, "i32"))
    if name in ("ADD", "SUB", "MUL"):
        return dict(act=fb.scalar(t, 0, "i8"))
    if name == "ONE_HOT":
        return dict(axis=fb.scalar(t, 0, "i32", -1))
    if name == "CAST":
        return dict(in_type=fb.scalar(t, 0, "i8"), out_type=fb.scalar(t, 1, "i8"))
    return {}


def load(path):
    with open(path, "rb") as f:
        buf = f.read()
    fb = _FB(buf)
    root = fb.root()
    m = Model()
    m.version = fb.scalar(root, 0, "u32")
    m.description = fb.string(root, 3)
    # operator codes
    codes = []
    for t in fb.vec_tables(root, 1):
        dep = fb.scalar(t, 0, "i8")
        new = fb.scalar(t, 3, "i32")
        codes.append(max(dep, new))
    # buffers
    bufs = []
    for t in fb.vec_tables(root, 4):
        s, n = fb.vec(t, 0)
        bufs.append(None if s is None or n == 0 else (s, n))
    m.subgraphs = []
    for sgt in fb.vec_tables(root, 2):
        sg = SubGraph()
        sg.name = fb.string(sgt, 4)
        sg.inputs = list(fb.vec_np(sgt, 1, np.int32))
        sg.outputs = list(fb.vec_np(sgt, 2, np.int32))
        sg.tensors = []
        for i, tt in enumerate(fb.vec_tables(sgt, 0)):
            T = Tensor()
            T.index = i
            T.shape = tuple(int(x) for x in fb.vec_np(tt, 0, np.int32))
            T.type = fb.scalar(tt, 1, "i8")
            T.dtype = TENSOR_TYPES.get(T.type)
            T.buffer = fb.scalar(tt, 2, "u32")
            T.name = fb.string(tt, 3)
            T.is_variable = bool(fb.scalar(tt, 5, "u8"))
            q = fb.table(tt, 4)
            if q is not None:
                T.scale = fb.vec_np(q, 2, np.float32)
                T.zero_point = fb.vec_np(q, 3, np.int64)
                T.qdim = fb.scalar(q, 6, "i32")
            else:
                T.scale = np.zeros(0, np.float32)
                T.zero_point = np.zeros(0, np.int64)
                T.qdim = 0
            T.data = None
            b = bufs[T.buffer] if T.buffer < len(bufs) else None
            if b is not None and T.dtype is not None:
                s, n = b
                arr = np.frombuffer(buf, dtype=T.dtype, count=n // np.dtype(T.dtype).itemsize,
                                    offset=s).copy()
                T.data = arr.reshape(T.shape) if T.shape else arr.reshape(())
            sg.tensors.append(T)
        sg.ops = []
        for i, ot in enumerate(fb.vec_tables(sgt, 3)):
            op = Op()
            op.index = i
            op.code = codes[fb.scalar(ot, 0, "u32")]
            op.name = OPCODES.get(op.code, f"OP{op.code}")
            op.inputs = [int(x) for x in fb.vec_np(ot, 1, np.int32)]
            op.outputs = [int(x) for x in fb.vec_np(ot, 2, np.int32)]
            op.opts = _parse_options(fb, op.name, fb.table(ot, 4))
            sg.ops.append(op)
        m.subgraphs.append(sg)
    # signatures
    m.signatures = {}
    for st in fb.vec_tables(root, 7):
        key = fb.string(st, 2)
        sgi = fb.scalar(st, 4, "u32")
        ins = {fb.string(t, 0): fb.scalar(t, 1, "u32") for t in fb.vec_tables(st, 0)}
        outs = {fb.string(t, 0): fb.scalar(t, 1, "u32") for t in fb.vec_tables(st, 1)}
        m.signatures[key] = dict(subgraph=sgi, inputs=ins, outputs=outs)
    return m


if __name__ == "__main__":
    import sys
    m = load(sys.argv[1])
    print("version", m.version, m.description, "signatures", m.signatures)
    for si, sg in enumerate(m.subgraphs):
        print(f"== subgraph {si} '{sg.name}' inputs={sg.inputs} outputs={sg.outputs} "
              f"{len(sg.tensors)} tensors {len(sg.ops)} ops")
        if len(sys.argv) > 2:
            for op in sg.ops:
                print(" ", op)
                for t in op.inputs:
                    if t >= 0:
                        print("      in ", sg.tensors[t])
                for t in op.outputs:
                    print("      out", sg.tensors[t])

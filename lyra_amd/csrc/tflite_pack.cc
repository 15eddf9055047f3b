// tflite_pack.cc -- reads a google/lyra model directory exactly as the reference ships it
// (lyra/model_coeffs: soundstream_encoder.tflite, quantizer.tflite, lyragan.tflite, lyra_config.binarypb) and turns it
// into the LYRAPK01 tensor container the model builder consumes (model.hip; same bytes tools/pack_weights.py writes).
// This is what makes lyra_hip_create() a drop-in for CreateFeatureExtractor / CreateQuantizer /
// CreateGenerativeModel (lyra/lyra_components.cc:42-55) on an unmodified reference model directory: the reference
// hands the flatbuffers to TfLiteModelWrapper::Create (lyra/tflite_model_wrapper.cc:36-95); here a dependency-free
// flatbuffer walk (TFLite schema v3, the handful of tables these graphs use) pulls out coefficients and
// quantisation parameters, keyed by role, and checks the structural facts the kernels rely on.
//
// Plain C++17, no HIP.  Also built into lyra_amd/pack_tool (csrc/Makefile) for offline conversion / tests.
#include "tflite_pack.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

namespace lyra {
namespace {

// ---- flatbuffer access -------------------------------------------------------------------------------------------
struct FB {
  const uint8_t* b;
  size_t n;
  bool bad = false;
  template <class T>
  T rd(size_t o) {
    T v{};
    if (o + sizeof(T) > n) { bad = true; return v; }
    memcpy(&v, b + o, sizeof(T));
    return v;
  }
  size_t root() { return rd<uint32_t>(0); }
  // absolute offset of field `fid` of the table at `tbl`, or 0
  size_t field(size_t tbl, int fid) {
    const size_t vt = tbl - (size_t)(int64_t)rd<int32_t>(tbl);
    const size_t vt_size = rd<uint16_t>(vt);
    const size_t slot = 4 + 2 * (size_t)fid;
    if (slot >= vt_size) return 0;
    const uint16_t off = rd<uint16_t>(vt + slot);
    return off ? tbl + off : 0;
  }
  size_t indirect(size_t o) { return o + rd<uint32_t>(o); }
  template <class T>
  T scalar(size_t tbl, int fid, T def) {
    const size_t o = field(tbl, fid);
    return o ? rd<T>(o) : def;
  }
  size_t table(size_t tbl, int fid) {
    const size_t o = field(tbl, fid);
    return o ? indirect(o) : 0;
  }
  // (start, length) of a vector field
  std::pair<size_t, size_t> vec(size_t tbl, int fid) {
    const size_t o = field(tbl, fid);
    if (!o) return {0, 0};
    const size_t v = indirect(o);
    return {v + 4, rd<uint32_t>(v)};
  }
  std::vector<size_t> vec_tables(size_t tbl, int fid) {
    auto [s, len] = vec(tbl, fid);
    std::vector<size_t> r;
    for (size_t i = 0; i < len; ++i) r.push_back(indirect(s + 4 * i));
    return r;
  }
  template <class T>
  std::vector<T> vec_of(size_t tbl, int fid) {
    auto [s, len] = vec(tbl, fid);
    std::vector<T> r(len);
    for (size_t i = 0; i < len; ++i) r[i] = rd<T>(s + sizeof(T) * i);
    return r;
  }
  std::string str(size_t tbl, int fid) {
    const size_t o = field(tbl, fid);
    if (!o) return "";
    const size_t v = indirect(o);
    const size_t len = rd<uint32_t>(v);
    if (v + 4 + len > n) { bad = true; return ""; }
    return std::string(reinterpret_cast<const char*>(b + v + 4), len);
  }
};

// BuiltinOperator codes of the ops that carry something we need (tensorflow/lite/schema/schema.fbs, v3)
enum : int { OP_ADD = 0, OP_CONCATENATION = 2, OP_CONV_2D = 3, OP_DEPTHWISE_CONV_2D = 4, OP_DEQUANTIZE = 6,
              OP_GATHER = 36, OP_SUB = 41, OP_TRANSPOSE_CONV = 67, OP_LEAKY_RELU = 98, OP_SQUARED_DIFFERENCE = 99,
              OP_QUANTIZE = 114 };
enum : int { TT_FLOAT32 = 0, TT_INT32 = 2, TT_INT8 = 9 };

struct Tensor {
  std::vector<int32_t> shape;
  int type = 0;
  std::vector<float> scale;
  std::vector<int64_t> zero_point;
  const uint8_t* data = nullptr;   // constant payload inside the flatbuffer, or nullptr
  size_t nbytes = 0;
  size_t count() const { size_t c = 1; for (int d : shape) c *= (size_t)d; return c; }
};
struct Op {
  int code = 0;
  std::vector<int32_t> inputs, outputs;
  size_t opts = 0;   // builtin options table (0 if none)
};
struct SubGraph {
  std::vector<Tensor> tensors;
  std::vector<Op> ops;
};
struct Graph {
  std::vector<uint8_t> bytes;
  std::vector<SubGraph> subgraphs;
  std::map<std::string, int> signatures;   // signature key -> subgraph index
};

bool read_file(const std::string& path, std::vector<uint8_t>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize(sz > 0 ? (size_t)sz : 0);
  size_t got = sz > 0 ? fread(out->data(), 1, (size_t)sz, f) : 0;
  fclose(f);
  return sz >= 0 && got == (size_t)sz;
}

bool load_graph(const std::string& path, Graph* g, std::string* err) {
  if (!read_file(path, &g->bytes) || g->bytes.size() < 8) { *err = "cannot read " + path; return false; }
  FB fb{g->bytes.data(), g->bytes.size()};
  const size_t root = fb.root();
  if (fb.scalar<uint32_t>(root, 0, 0) != 3) { *err = path + ": not a TFLite schema-version-3 model"; return false; }
  std::vector<int> codes;
  for (size_t t : fb.vec_tables(root, 1)) {
    const int dep = fb.scalar<int8_t>(t, 0, 0), neu = fb.scalar<int32_t>(t, 3, 0);
    codes.push_back(std::max(dep, neu));
  }
  std::vector<std::pair<size_t, size_t>> bufs;
  for (size_t t : fb.vec_tables(root, 4)) bufs.push_back(fb.vec(t, 0));
  for (size_t sgt : fb.vec_tables(root, 2)) {
    SubGraph sg;
    for (size_t tt : fb.vec_tables(sgt, 0)) {
      Tensor T;
      T.shape = fb.vec_of<int32_t>(tt, 0);
      T.type = fb.scalar<int8_t>(tt, 1, 0);
      const uint32_t bi = fb.scalar<uint32_t>(tt, 2, 0);
      if (const size_t q = fb.table(tt, 4)) {
        T.scale = fb.vec_of<float>(q, 2);
        T.zero_point = fb.vec_of<int64_t>(q, 3);
      }
      if (bi < bufs.size() && bufs[bi].second > 0 && bufs[bi].first + bufs[bi].second <= g->bytes.size()) {
        T.data = g->bytes.data() + bufs[bi].first;
        T.nbytes = bufs[bi].second;
      }
      sg.tensors.push_back(std::move(T));
    }
    for (size_t ot : fb.vec_tables(sgt, 3)) {
      Op op;
      const uint32_t ci = fb.scalar<uint32_t>(ot, 0, 0);
      op.code = ci < codes.size() ? codes[ci] : -1;
      op.inputs = fb.vec_of<int32_t>(ot, 1);
      op.outputs = fb.vec_of<int32_t>(ot, 2);
      op.opts = fb.table(ot, 4);
      sg.ops.push_back(std::move(op));
    }
    // operand indices must name tensors of this subgraph (the walkers index without further checks); -1 marks an
    // omitted optional operand and is turned into a reference to an empty tensor appended at the end
    const int32_t nt = (int32_t)sg.tensors.size();
    sg.tensors.push_back(Tensor());
    for (Op& op : sg.ops) {
      for (auto* v : {&op.inputs, &op.outputs})
        for (int32_t& i : *v) {
          if (i == -1) i = nt;
          if (i < 0 || i > nt) { *err = path + ": operand index out of range"; return false; }
        }
    }
    g->subgraphs.push_back(std::move(sg));
  }
  for (size_t st : fb.vec_tables(root, 7)) {
    const int sgi = (int)fb.scalar<uint32_t>(st, 4, 0);
    if (sgi < 0 || sgi >= (int)g->subgraphs.size()) { *err = path + ": signature names a missing subgraph"; return false; }
    g->signatures[fb.str(st, 2)] = sgi;
  }
  if (fb.bad) { *err = path + ": truncated or malformed flatbuffer"; return false; }
  return true;
}

// ---- the tensors of the container ---------------------------------------------------------------------------------
struct Blob {
  uint32_t dtype = 0;   // 0 f32, 1 i8, 2 i32
  std::vector<uint32_t> shape;
  std::vector<uint8_t> bytes;
};
using Blobs = std::map<std::string, Blob>;   // std::map iterates in sorted name order, like sorted() in the tool

template <class T>
Blob make_blob(uint32_t dtype, std::vector<uint32_t> shape, const T* p, size_t count) {
  Blob b;
  b.dtype = dtype;
  b.shape = std::move(shape);
  b.bytes.resize(count * sizeof(T));
  if (count) memcpy(b.bytes.data(), p, count * sizeof(T));
  return b;
}
Blob f32s(std::initializer_list<float> v) { return make_blob<float>(0, {(uint32_t)v.size()}, v.begin(), v.size()); }
Blob i32s(std::initializer_list<int32_t> v) { return make_blob<int32_t>(2, {(uint32_t)v.size()}, v.begin(), v.size()); }

struct Walker {
  FB fb;
  std::string prefix;
  Blobs* out;
  std::string* err;
  std::map<std::string, int> cnt;

  std::string next(const char* kind) {
    const int i = cnt[kind]++;
    return prefix + "." + kind + "." + std::to_string(i);
  }
  bool fail(const std::string& m) { if (err->empty()) *err = prefix + ": " + m; return false; }

  // (scale, zero point) of a per-tensor quantised tensor; (0, 0) for float tensors
  bool qparams(const Tensor& t, float* s, float* z) {
    *s = 0.f; *z = 0.f;
    if (t.scale.empty()) return true;
    if (t.scale.size() != 1) return fail("per-tensor quantisation expected");
    *s = t.scale[0];
    *z = t.zero_point.empty() ? 0.f : (float)t.zero_point[0];
    return true;
  }
  // raw constant payload of a tensor as a blob of its own dtype
  bool payload(const Tensor& t, std::vector<uint32_t> shape, Blob* b) {
    if (!t.data) return fail("constant tensor without data");
    size_t cnt_ = 1;
    for (uint32_t d : shape) cnt_ *= d;
    const uint32_t dt = t.type == TT_FLOAT32 ? 0u : (t.type == TT_INT8 ? 1u : 2u);
    const size_t esz = dt == 1 ? 1 : 4;
    if (t.type != TT_FLOAT32 && t.type != TT_INT8 && t.type != TT_INT32) return fail("unsupported tensor type");
    if (cnt_ * esz != t.nbytes) return fail("constant tensor size mismatch");
    b->dtype = dt;
    b->shape = std::move(shape);
    b->bytes.assign(t.data, t.data + t.nbytes);
    return true;
  }

  bool conv(const SubGraph& sg, const Op& op, bool depthwise) {
    if (op.inputs.size() < 3 || op.outputs.empty()) return fail("conv with too few operands");
    const Tensor &x = sg.tensors[op.inputs[0]], &w = sg.tensors[op.inputs[1]], &b = sg.tensors[op.inputs[2]];
    const Tensor& y = sg.tensors[op.outputs[0]];
    if (w.shape.size() != 4 || x.shape.size() != 4) return fail("conv operands must be rank 4");
    const std::string base = next(depthwise ? "dw" : "conv");
    int groups, k;
    Blob wb;
    if (!depthwise) {
      const int cout = w.shape[0], cig = w.shape[3];
      k = w.shape[1];
      groups = x.shape[3] / cig;
      if (!payload(w, {(uint32_t)cout, (uint32_t)k, (uint32_t)cig}, &wb)) return false;
    } else {
      k = w.shape[1];
      groups = w.shape[3];
      if (!payload(w, {(uint32_t)k, (uint32_t)w.shape[3]}, &wb)) return false;
    }
    (*out)[base + ".w"] = std::move(wb);
    Blob bb;
    if (!payload(b, {(uint32_t)b.count()}, &bb)) return false;
    (*out)[base + ".b"] = std::move(bb);
    float sx, zx, sy, zy;
    if (!qparams(x, &sx, &zx) || !qparams(y, &sy, &zy)) return false;
    (*out)[base + ".q"] = f32s({sx, zx, sy, zy});
    // Conv2DOptions: stride_h = field 2, dilation_h = 5; DepthwiseConv2DOptions: stride_h = 2, dilation_h = 6
    const int stride = fb.scalar<int32_t>(op.opts, 2, 0);
    const int dil = fb.scalar<int32_t>(op.opts, depthwise ? 6 : 5, 1);
    (*out)[base + ".opt"] = i32s({stride, dil, groups, k});
    if (w.type == TT_INT8) {
      for (int64_t z : w.zero_point) if (z != 0) return fail("int8 weights must be symmetric");
      (*out)[base + ".wscale"] = make_blob<float>(0, {(uint32_t)w.scale.size()}, w.scale.data(), w.scale.size());
      // bias scale must be s_in * s_w (TFLite convention) -- the kernels fold the bias into the int32 accumulator
      if (b.scale.size() != w.scale.size()) return fail("bias / weight scale count mismatch");
      for (size_t i = 0; i < w.scale.size(); ++i) {
        const double want = (double)x.scale[0] * (double)w.scale[i];
        if (std::fabs((double)b.scale[i] - want) > 1e-6 * std::fabs(want) + 1e-30) return fail("bias scale != s_in * s_w");
      }
    }
    return true;
  }

  bool tconv(const SubGraph& sg, const Op& op) {
    if (op.inputs.size() < 4) return fail("TRANSPOSE_CONV without bias");
    const Tensor &w = sg.tensors[op.inputs[1]], &x = sg.tensors[op.inputs[2]], &b = sg.tensors[op.inputs[3]];
    const Tensor& y = sg.tensors[op.outputs[0]];
    if (w.shape.size() != 4) return fail("tconv weights must be rank 4");
    const std::string base = next("tconv");
    const int cout = w.shape[0], k = w.shape[1], cin = w.shape[3];
    Blob wb, bb;
    if (!payload(w, {(uint32_t)cout, (uint32_t)k, (uint32_t)cin}, &wb) || !payload(b, {(uint32_t)b.count()}, &bb)) return false;
    (*out)[base + ".w"] = std::move(wb);
    (*out)[base + ".b"] = std::move(bb);
    float sx, zx, sy, zy;
    if (!qparams(x, &sx, &zx) || !qparams(y, &sy, &zy)) return false;
    (*out)[base + ".q"] = f32s({sx, zx, sy, zy});
    (*out)[base + ".opt"] = i32s({fb.scalar<int32_t>(op.opts, 2, 0), 1, 1, k});   // TransposeConvOptions.stride_h
    if (w.type == TT_INT8) {
      if (w.scale.size() != 1) return fail("per-tensor int8 tconv weights expected");
      for (int64_t z : w.zero_point) if (z != 0) return fail("int8 weights must be symmetric");
      (*out)[base + ".wscale"] = make_blob<float>(0, {1}, w.scale.data(), 1);
    }
    return true;
  }

  bool walk(const SubGraph& sg) {
    for (const Op& op : sg.ops) {
      const size_t need_in = op.code == OP_TRANSPOSE_CONV ? 4 : (op.code == OP_CONV_2D || op.code == OP_DEPTHWISE_CONV_2D) ? 3
                           : (op.code == OP_ADD || op.code == OP_SUB) ? 2
                           : (op.code == OP_LEAKY_RELU || op.code == OP_QUANTIZE || op.code == OP_DEQUANTIZE) ? 1 : 0;
      if (op.inputs.size() < need_in || (need_in && op.outputs.empty())) return fail("operator with too few operands");
      switch (op.code) {
        case OP_CONV_2D: if (!conv(sg, op, false)) return false; break;
        case OP_DEPTHWISE_CONV_2D: if (!conv(sg, op, true)) return false; break;
        case OP_TRANSPOSE_CONV: if (!tconv(sg, op)) return false; break;
        case OP_LEAKY_RELU: {
          const Tensor &x = sg.tensors[op.inputs[0]], &y = sg.tensors[op.outputs[0]];
          const float alpha = fb.scalar<float>(op.opts, 0, 0.f);
          if (std::fabs((double)alpha - 0.30000001192092896) > 1e-12) return fail("LeakyReLU alpha is not 0.3f");
          if (x.type == TT_INT8) {
            float sx, zx, sy, zy;
            if (!qparams(x, &sx, &zx) || !qparams(y, &sy, &zy)) return false;
            (*out)[next("lrelu8") + ".q"] = f32s({sx, zx, sy, zy});
          }
          break;
        }
        case OP_ADD: {
          const Tensor &a = sg.tensors[op.inputs[0]], &b = sg.tensors[op.inputs[1]], &y = sg.tensors[op.outputs[0]];
          if (a.type == TT_INT8) {
            float s1, z1, s2, z2, so, zo;
            if (!qparams(a, &s1, &z1) || !qparams(b, &s2, &z2) || !qparams(y, &so, &zo)) return false;
            (*out)[next("add8") + ".q"] = f32s({s1, z1, s2, z2, so, zo});
          } else if (a.data || b.data) {
            return fail("unexpected constant float ADD");
          }
          break;
        }
        case OP_QUANTIZE: {
          float s, z;
          if (!qparams(sg.tensors[op.outputs[0]], &s, &z)) return false;
          (*out)[next("quant") + ".q"] = f32s({s, z});
          break;
        }
        case OP_DEQUANTIZE: {
          float s, z;
          if (!qparams(sg.tensors[op.inputs[0]], &s, &z)) return false;
          (*out)[next("dequant") + ".q"] = f32s({s, z});
          break;
        }
        case OP_SUB: {
          const Tensor& c = sg.tensors[op.inputs[1]];
          if (!c.data || c.type != TT_FLOAT32) return fail("SUB without a float constant");
          Blob b;
          std::vector<uint32_t> shape;
          for (int d : c.shape) shape.push_back((uint32_t)d);
          if (shape.empty()) shape.push_back(1);   // a scalar is stored as [1]
          if (!payload(c, shape, &b)) return false;
          (*out)[next("sub") + ".c"] = std::move(b);
          break;
        }
        case OP_CONCATENATION:
          // constant operands of the time-axis concats must be all-zero (transposed-conv state padding)
          for (int32_t i : op.inputs) {
            const Tensor& t = sg.tensors[i];
            if (t.data && std::any_of(t.data, t.data + t.nbytes, [](uint8_t v) { return v != 0; }))
              return fail("non-zero constant in CONCATENATION");
          }
          break;
        default: break;
      }
    }
    return true;
  }
};

bool walk_graph(const char* prefix, Graph& g, Blobs* out, std::string* err) {
  if (g.subgraphs.size() < 2) { *err = std::string(prefix) + ": expected a main and an init subgraph"; return false; }
  Walker w{FB{g.bytes.data(), g.bytes.size()}, prefix, out, err, {}};
  if (!w.walk(g.subgraphs[0])) return false;
  // the init subgraph (CALL_ONCE) must assign zeros only: the kernels start every stream from zero state
  for (const Tensor& t : g.subgraphs[1].tensors)
    if (t.data && t.type == TT_FLOAT32 && std::any_of(t.data, t.data + t.nbytes, [](uint8_t v) { return v != 0; })) {
      *err = std::string(prefix) + ": non-zero initial state";
      return false;
    }
  return true;
}

}  // namespace

bool pack_from_tflite_dir(const std::string& dir, std::vector<uint8_t>* container, std::string* err) {
  Graph enc, gan, qz;
  if (!load_graph(dir + "/soundstream_encoder.tflite", &enc, err) || !load_graph(dir + "/lyragan.tflite", &gan, err) ||
      !load_graph(dir + "/quantizer.tflite", &qz, err))
    return false;
  Blobs out;
  if (!walk_graph("enc", enc, &out, err) || !walk_graph("dec", gan, &out, err)) return false;
  // RVQ codebooks: second operand of each SQUARED_DIFFERENCE of the `encode` signature, in stage order
  auto sig = qz.signatures.find("encode");
  if (sig == qz.signatures.end() || sig->second >= (int)qz.subgraphs.size()) { *err = "quantizer.tflite: no `encode` signature"; return false; }
  const SubGraph& sge = qz.subgraphs[sig->second];
  std::vector<const float*> cbs;
  for (const Op& op : sge.ops)
    if (op.code == OP_SQUARED_DIFFERENCE) {
      const Tensor& c = sge.tensors[op.inputs[1]];
      if (!c.data || c.nbytes != 16 * 64 * 4) { *err = "quantizer.tflite: codebook is not [16][64] float"; return false; }
      cbs.push_back(reinterpret_cast<const float*>(c.data));
    }
  if (cbs.size() != 46) { *err = "quantizer.tflite: expected 46 RVQ stages"; return false; }
  // the gather tables of encode and decode must be these codebooks (the decode graph lists them in name order)
  for (const char* name : {"encode", "decode"}) {
    auto s2 = qz.signatures.find(name);
    if (s2 == qz.signatures.end()) { *err = std::string("quantizer.tflite: no `") + name + "` signature"; return false; }
    const SubGraph& sg = qz.subgraphs[s2->second];
    int n = 0;
    for (const Op& op : sg.ops)
      if (op.code == OP_GATHER) {
        const Tensor& t = sg.tensors[op.inputs[0]];
        bool found = false;
        for (const float* cb : cbs) found = found || (t.data && t.nbytes == 4096 && memcmp(t.data, cb, 4096) == 0);
        if (!found) { *err = std::string("quantizer.tflite: ") + name + " gathers from a table that is not a codebook"; return false; }
        ++n;
      }
    if (n != 45 && n != 46) { *err = "quantizer.tflite: unexpected number of GATHER ops"; return false; }
  }
  {
    Blob cb;
    cb.dtype = 0;
    cb.shape = {46, 16, 64};
    cb.bytes.resize((size_t)46 * 4096);
    for (int k = 0; k < 46; ++k) memcpy(cb.bytes.data() + (size_t)k * 4096, cbs[k], 4096);
    out["rvq.codebooks"] = std::move(cb);
  }
  // lyra_config.binarypb: field 1 (varint) = the identifier (lyra_config.proto; lyra_config.h:145-166 wants 3)
  std::vector<uint8_t> pb;
  if (!read_file(dir + "/lyra_config.binarypb", &pb) || pb.size() != 2 || pb[0] != 0x08) {
    *err = "cannot read " + dir + "/lyra_config.binarypb (expected one varint field)";
    return false;
  }
  out["meta.version"] = i32s({(int32_t)pb[1]});

  // ---- serialise: LYRAPK01, entries sorted by name, payloads 64-byte aligned (tools/pack_weights.py layout) ----
  const size_t hdr = 16 + 96 * out.size();
  size_t off = (hdr + 63) / 64 * 64;
  std::vector<uint8_t>& c = *container;
  c.assign(off, 0);
  memcpy(c.data(), "LYRAPK01", 8);
  const uint32_t n = (uint32_t)out.size();
  memcpy(c.data() + 8, &n, 4);
  size_t ei = 0;
  for (const auto& [name, b] : out) {
    if (name.size() >= 56 || b.shape.size() > 4) { *err = "internal: bad entry " + name; return false; }
    PackEntry e;
    memset(&e, 0, sizeof e);
    memcpy(e.name, name.data(), name.size());
    e.dtype = b.dtype;
    e.ndim = (uint32_t)b.shape.size();
    for (int i = 0; i < 4; ++i) e.shape[i] = i < (int)b.shape.size() ? b.shape[i] : 1;
    e.offset = off;
    e.nbytes = b.bytes.size();
    memcpy(c.data() + 16 + 96 * ei++, &e, 96);
    const size_t end = (off + b.bytes.size() + 63) / 64 * 64;
    c.resize(end, 0);
    memcpy(c.data() + off, b.bytes.data(), b.bytes.size());
    off = end;
  }
  return true;
}

}  // namespace lyra

/* TEST INFRASTRUCTURE ONLY -- a thin graph builder over the XNNPACK that this image's libtorch_cpu.so exports.
 *
 * The reference runs soundstream_encoder.tflite and lyragan.tflite through TFLite's XNNPACK delegate
 * (soundstream_encoder.cc:39-40, lyra_gan_model.cc:39-40: use_xnn=true; tflite_model_wrapper.cc:63-85), i.e. the
 * delegate turns each supported partition of the flatbuffer into an xnn_subgraph (xnn_define_*_tensor_value with
 * the flatbuffer's quantisation parameters, one xnn_define_* node per operator), creates a runtime and invokes it.
 * This file exposes exactly that sequence through a handful of C entry points so that Python
 * (oracle/xnn_witness.py, tests/test_xnnpack_witness.py) can rebuild single operators or whole int8 regions of
 * the two graphs on operands traced by oracle/tflite_interp.py and read back what *real XNNPACK code* computes.
 *
 * It is a WITNESS, not the binary of record: the XNNPACK inside torch 2.10 is newer than the commit TensorFlow 2.11
 * pins (WORKSPACE:168-174 of the reference), and it runs x86 micro-kernels here.  Nothing under lyra_amd/ links,
 * loads or calls this file.
 *
 * Build: oracle/Makefile target _xnn/libxnn_witness.so (header: torch/include/xnnpack.h; link: torch/lib/libtorch_cpu.so).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <xnnpack.h>

#define XW_MAX_EXT 16

typedef struct {
  xnn_subgraph_t sg;
  xnn_runtime_t rt;
  int status;           /* first failing xnn_status, 0 = ok */
} xw_graph;

static int g_init = 0;

int xw_init(void) {
  if (!g_init) {
    enum xnn_status s = xnn_initialize(NULL);
    if (s != xnn_status_success) return (int)s;
    g_init = 1;
  }
  return 0;
}

xw_graph* xw_new(int n_external) {
  if (xw_init() != 0) return NULL;
  xw_graph* g = (xw_graph*)calloc(1, sizeof(xw_graph));
  if (xnn_create_subgraph((uint32_t)n_external, 0, &g->sg) != xnn_status_success) { free(g); return NULL; }
  return g;
}

static void note(xw_graph* g, enum xnn_status s) { if (s != xnn_status_success && g->status == 0) g->status = (int)s; }

static void dims4(const int64_t* d, int n, size_t* out) { for (int i = 0; i < n; i++) out[i] = (size_t)d[i]; }

/* ext: external id (>= 0) for graph inputs / outputs, -1 for internal / static values.
 * io: 1 = external input, 2 = external output, 0 = neither. */
static uint32_t io_flags(int io) {
  return io == 1 ? XNN_VALUE_FLAG_EXTERNAL_INPUT : io == 2 ? XNN_VALUE_FLAG_EXTERNAL_OUTPUT : 0;
}

int xw_tensor_f32(xw_graph* g, int nd, const int64_t* d, const void* data, int ext, int io) {
  size_t dd[XNN_MAX_TENSOR_DIMS]; dims4(d, nd, dd);
  uint32_t id = XNN_INVALID_VALUE_ID;
  note(g, xnn_define_tensor_value(g->sg, xnn_datatype_fp32, (size_t)nd, dd, data,
                                  ext >= 0 ? (uint32_t)ext : XNN_INVALID_VALUE_ID, io_flags(io), &id));
  return (int)id;
}

/* per-tensor quantised int8 (bits = 8) or int32 bias (bits = 32) */
int xw_tensor_q(xw_graph* g, int bits, int zero_point, float scale, int nd, const int64_t* d, const void* data, int ext,
                int io) {
  size_t dd[XNN_MAX_TENSOR_DIMS]; dims4(d, nd, dd);
  uint32_t id = XNN_INVALID_VALUE_ID;
  note(g, xnn_define_quantized_tensor_value(g->sg, bits == 8 ? xnn_datatype_qint8 : xnn_datatype_qint32, zero_point, scale,
                                            (size_t)nd, dd, data, ext >= 0 ? (uint32_t)ext : XNN_INVALID_VALUE_ID,
                                            io_flags(io), &id));
  return (int)id;
}

/* per-channel quantised static int8 filter (bits = 8) or int32 bias (bits = 32) */
int xw_tensor_qc(xw_graph* g, int bits, const float* scales, int nd, int channel_dim, const int64_t* d, const void* data) {
  size_t dd[XNN_MAX_TENSOR_DIMS]; dims4(d, nd, dd);
  uint32_t id = XNN_INVALID_VALUE_ID;
  note(g, xnn_define_channelwise_quantized_tensor_value(g->sg, bits == 8 ? xnn_datatype_qcint8 : xnn_datatype_qcint32, scales,
                                                        (size_t)nd, (size_t)channel_dim, dd, data, XNN_INVALID_VALUE_ID, 0,
                                                        &id));
  return (int)id;
}

/* This libtorch_cpu.so exports the subgraph API only for values, unary and binary nodes (xnn_define_convolution_2d /
 * _deconvolution_2d / _depthwise_convolution_2d are not among its dynamic symbols), so the convolutions go through the
 * operator API -- the very functions the subgraph's nodes call when the runtime is created (create -> reshape ->
 * setup -> run).  All of them: [KH,1] kernel, VALID padding, batch 1, width 1 (the only form in the two graphs).
 * kind: 0 = CONV_2D (filter [G*goc, KH, 1, gic]), 1 = DEPTHWISE_CONV_2D (filter [1, KH, 1, C]; groups = C).
 * kscale: per-output-channel filter scales (n_kscale = G*goc), or one scale (n_kscale = 1: broadcast for the qc8w
 * operator, and the per-tensor qs8 operator is tried as well -- see per_tensor).  Returns xnn_status (0 = ok). */
static int run_and_delete(xnn_operator_t op) {
  enum xnn_status s = xnn_run_operator(op, NULL);
  xnn_delete_operator(op);
  return (int)s;
}

int xw_op_conv_q8(int kind, int per_tensor, int h, int kh, int stride, int dil, int groups, int gic, int goc, int in_zp,
                  float in_scale, const float* kscale, int n_kscale, const int8_t* kernel, const int32_t* bias, int out_zp,
                  float out_scale, const int8_t* input, int8_t* output, int* h_out) {
  if (xw_init() != 0) return -1;
  xnn_operator_t op = NULL;
  const size_t cin = (size_t)groups * gic, cout = (size_t)groups * goc;
  const uint32_t flags = kind == 1 ? XNN_FLAG_DEPTHWISE_CONVOLUTION : 0;
  enum xnn_status s;
  float* ks = NULL;
  if (per_tensor) {
    s = xnn_create_convolution2d_nhwc_qs8(0, 0, 0, 0, (uint32_t)kh, 1, (uint32_t)stride, 1, (uint32_t)dil, 1, (uint32_t)groups,
                                          (size_t)gic, (size_t)goc, cin, cout, (int8_t)in_zp, in_scale, kscale[0], kernel, bias,
                                          (int8_t)out_zp, out_scale, -128, 127, flags, NULL, NULL, &op);
  } else {
    ks = (float*)malloc(sizeof(float) * cout);
    for (size_t i = 0; i < cout; i++) ks[i] = kscale[n_kscale == 1 ? 0 : i];
    s = xnn_create_convolution2d_nhwc_qs8_qc8w(0, 0, 0, 0, (uint32_t)kh, 1, (uint32_t)stride, 1, (uint32_t)dil, 1,
                                               (uint32_t)groups, (size_t)gic, (size_t)goc, cin, cout, (int8_t)in_zp, in_scale, ks,
                                               kernel, bias, (int8_t)out_zp, out_scale, -128, 127, flags, NULL, NULL, &op);
  }
  free(ks);
  if (s != xnn_status_success) return (int)s;
  size_t ws = 0, wa = 0, oh = 0, ow = 0;
  s = per_tensor ? xnn_reshape_convolution2d_nhwc_qs8(op, 1, (size_t)h, 1, &ws, &wa, &oh, &ow, NULL)
                 : xnn_reshape_convolution2d_nhwc_qs8_qc8w(op, 1, (size_t)h, 1, &ws, &wa, &oh, &ow, NULL);
  if (s != xnn_status_success) { xnn_delete_operator(op); return 1000 + (int)s; }
  void* wsp = ws ? aligned_alloc(wa > 64 ? wa : 64, (ws + 63) / 64 * 64 + 64) : NULL;
  s = per_tensor ? xnn_setup_convolution2d_nhwc_qs8(op, wsp, input, output)
                 : xnn_setup_convolution2d_nhwc_qs8_qc8w(op, wsp, input, output);
  if (s != xnn_status_success) { xnn_delete_operator(op); free(wsp); return 2000 + (int)s; }
  *h_out = (int)oh;
  int r = run_and_delete(op);
  free(wsp);
  return r ? 3000 + r : 0;
}

int xw_op_conv_f32(int kind, int h, int kh, int stride, int dil, int groups, int gic, int goc, const float* kernel,
                   const float* bias, const float* input, float* output, int* h_out) {
  if (xw_init() != 0) return -1;
  xnn_operator_t op = NULL;
  const size_t cin = (size_t)groups * gic, cout = (size_t)groups * goc;
  enum xnn_status s = xnn_create_convolution2d_nhwc_f32(0, 0, 0, 0, (uint32_t)kh, 1, (uint32_t)stride, 1, (uint32_t)dil, 1,
                                                        (uint32_t)groups, (size_t)gic, (size_t)goc, cin, cout, kernel, bias,
                                                        -INFINITY, INFINITY, kind == 1 ? XNN_FLAG_DEPTHWISE_CONVOLUTION : 0,
                                                        NULL, NULL, &op);
  if (s != xnn_status_success) return (int)s;
  size_t ws = 0, wa = 0, oh = 0, ow = 0;
  s = xnn_reshape_convolution2d_nhwc_f32(op, 1, (size_t)h, 1, &ws, &wa, &oh, &ow, NULL);
  if (s != xnn_status_success) { xnn_delete_operator(op); return 1000 + (int)s; }
  void* wsp = ws ? aligned_alloc(wa > 64 ? wa : 64, (ws + 63) / 64 * 64 + 64) : NULL;
  s = xnn_setup_convolution2d_nhwc_f32(op, wsp, input, output);
  if (s != xnn_status_success) { xnn_delete_operator(op); free(wsp); return 2000 + (int)s; }
  *h_out = (int)oh;
  int r = run_and_delete(op);
  free(wsp);
  return r ? 3000 + r : 0;
}

/* TRANSPOSE_CONV: filter [cout, KH, 1, cin]; per-tensor filter scale (what both graphs carry) */
int xw_op_deconv_q8(int h, int kh, int stride, int cin, int cout, int in_zp, float in_scale, float kscale,
                    const int8_t* kernel, const int32_t* bias, int out_zp, float out_scale, const int8_t* input,
                    int8_t* output, int* h_out) {
  if (xw_init() != 0) return -1;
  xnn_operator_t op = NULL;
  enum xnn_status s = xnn_create_deconvolution2d_nhwc_qs8(0, 0, 0, 0, (uint32_t)kh, 1, (uint32_t)stride, 1, 1, 1, 1, (size_t)cin,
                                                          (size_t)cout, (size_t)cin, (size_t)cout, (int8_t)in_zp, in_scale, kscale,
                                                          kernel, bias, (int8_t)out_zp, out_scale, -128, 127, 0, NULL, NULL, &op);
  if (s != xnn_status_success) return (int)s;
  size_t oh = 0, ow = 0;
  s = xnn_reshape_deconvolution2d_nhwc_qs8(op, 1, (size_t)h, 1, 0, 0, &oh, &ow, NULL);
  if (s != xnn_status_success) { xnn_delete_operator(op); return 1000 + (int)s; }
  s = xnn_setup_deconvolution2d_nhwc_qs8(op, input, output);
  if (s != xnn_status_success) { xnn_delete_operator(op); return 2000 + (int)s; }
  *h_out = (int)oh;
  int r = run_and_delete(op);
  return r ? 3000 + r : 0;
}

int xw_op_deconv_f32(int h, int kh, int stride, int cin, int cout, const float* kernel, const float* bias,
                     const float* input, float* output, int* h_out) {
  if (xw_init() != 0) return -1;
  xnn_operator_t op = NULL;
  enum xnn_status s = xnn_create_deconvolution2d_nhwc_f32(0, 0, 0, 0, (uint32_t)kh, 1, (uint32_t)stride, 1, 1, 1, 1, (size_t)cin,
                                                          (size_t)cout, (size_t)cin, (size_t)cout, kernel, bias, -INFINITY,
                                                          INFINITY, 0, NULL, NULL, &op);
  if (s != xnn_status_success) return (int)s;
  size_t oh = 0, ow = 0;
  s = xnn_reshape_deconvolution2d_nhwc_f32(op, 1, (size_t)h, 1, 0, 0, &oh, &ow, NULL);
  if (s != xnn_status_success) { xnn_delete_operator(op); return 1000 + (int)s; }
  s = xnn_setup_deconvolution2d_nhwc_f32(op, input, output);
  if (s != xnn_status_success) { xnn_delete_operator(op); return 2000 + (int)s; }
  *h_out = (int)oh;
  int r = run_and_delete(op);
  return r ? 3000 + r : 0;
}

void xw_leaky_relu(xw_graph* g, float alpha, int in, int out) {
  union xnn_unary_params p; memset(&p, 0, sizeof p); p.leaky_relu.negative_slope = alpha;
  note(g, xnn_define_unary(g->sg, xnn_unary_leaky_relu, &p, (uint32_t)in, (uint32_t)out, 0));
}

/* QUANTIZE / DEQUANTIZE */
void xw_convert(xw_graph* g, int in, int out) {
  note(g, xnn_define_unary(g->sg, xnn_unary_convert, NULL, (uint32_t)in, (uint32_t)out, 0));
}

void xw_add(xw_graph* g, int a, int b, int out) {
  struct xnn_binary_params p; p.output_min = -INFINITY; p.output_max = INFINITY;
  note(g, xnn_define_binary(g->sg, xnn_binary_add, &p, (uint32_t)a, (uint32_t)b, (uint32_t)out, 0));
}

int xw_status(xw_graph* g) { return g->status; }

/* create the runtime on first call (single thread: threadpool NULL, as the reference's num_threads = 1), bind the
 * external values, invoke. */
int xw_run(xw_graph* g, int n, const int* ext_ids, void** ptrs) {
  if (g->status) return g->status;
  if (!g->rt) {
    enum xnn_status s = xnn_create_runtime_v2(g->sg, NULL, 0, &g->rt);
    if (s != xnn_status_success) return (int)s;
  }
  struct xnn_external_value ev[XW_MAX_EXT];
  if (n > XW_MAX_EXT) return -1;
  for (int i = 0; i < n; i++) { ev[i].id = (uint32_t)ext_ids[i]; ev[i].data = ptrs[i]; }
  enum xnn_status s = xnn_reshape_runtime(g->rt);
  if (s != xnn_status_success) return 1000 + (int)s;
  s = xnn_setup_runtime_v2(g->rt, (size_t)n, ev);
  if (s != xnn_status_success) return 2000 + (int)s;
  s = xnn_invoke_runtime(g->rt);
  return s == xnn_status_success ? 0 : 3000 + (int)s;
}

void xw_free(xw_graph* g) {
  if (!g) return;
  if (g->rt) xnn_delete_runtime(g->rt);
  if (g->sg) xnn_delete_subgraph(g->sg);
  free(g);
}

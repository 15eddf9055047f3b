/* chain_f32.c -- TEST INFRASTRUCTURE ONLY: the fp32 convolution layers of the two Lyra graphs as explicit
 * single-rounding fmaf chains, in selectable summation orders.
 *
 * oracle/tflite_interp.py uses these for its fp32 CONV_2D / DEPTHWISE_CONV_2D / TRANSPOSE_CONV (numpy has no fused
 * multiply-add and no way to fix a summation order), and tests/test_xnnpack_witness.py holds them against the real
 * XNNPACK of this image (oracle/xnn_witness.c): the order XNNPACK's f32 GEMM / IGEMM / DWCONV / subconvolution
 * micro-kernels produce on an FMA target is
 *     acc = bias;  for tap k ascending (outer), input channel ascending (inner):  acc = fmaf(x, w, acc)
 * for every layer of both graphs (0 differing outputs), except the one-output-channel transposed conv at the end of
 * lyragan.tflite, which the x86 build routes to its "nr2" kernel (cf_deconv_c4 below).  A transposed conv's taps
 * ascending means input rows from the NEWEST to the OLDEST.
 *
 * Independent of oracle/lyra_oracle.c on purpose (the interpreter cross-checks that file).
 * Build: oracle/Makefile (gcc -O2 -mfma -ffp-contract=off: fmaf is the hardware's correctly rounded fused operation).
 */
#include <math.h>
#include <stddef.h>

/* Flattened operand index kk = tap * cin_g + channel, ascending.
 * bias_first = 1: acc starts from the bias (XNNPACK's GEMM / IGEMM / DWCONV micro-kernels load their accumulators from
 * the packed bias); 0: acc starts from 0, bias added last (the order the oracle used through round 3).
 * x [H, cin], w [cout, kh, cin_g], y [hout, cout]. */
void cf_conv(int bias_first, int h, int kh, int stride, int dil, int groups, int gic, int goc, const float* w,
                       const float* b, const float* x, float* y) {
  const int cin = groups * gic, cout = groups * goc;
  const int hout = (h - (kh - 1) * dil - 1) / stride + 1;
  for (int t = 0; t < hout; t++)
    for (int co = 0; co < cout; co++) {
      const int g = co / goc;
      float acc = bias_first ? b[co] : 0.0f;
      for (int k = 0; k < kh; k++)
        for (int c = 0; c < gic; c++)
          acc = fmaf(x[(size_t)(t * stride + k * dil) * cin + g * gic + c], w[((size_t)co * kh + k) * gic + c], acc);
      y[(size_t)t * cout + co] = bias_first ? acc : acc + b[co];
    }
}

/* depthwise: w [kh, c] */
void cf_dwconv(int bias_first, int h, int kh, int dil, int c, const float* w, const float* b, const float* x,
                         float* y) {
  const int hout = h - (kh - 1) * dil;
  for (int t = 0; t < hout; t++)
    for (int ch = 0; ch < c; ch++) {
      float acc = bias_first ? b[ch] : 0.0f;
      for (int k = 0; k < kh; k++) acc = fmaf(x[(size_t)(t + k * dil) * c + ch], w[(size_t)k * c + ch], acc);
      y[(size_t)t * c + ch] = bias_first ? acc : acc + b[ch];
    }
}

/* transposed conv, w [cout, kh, cin]; output row p sums the taps k with (p - k) % stride == 0 and 0 <= (p-k)/stride < h.
 * tap_order 0: k ascending (input rows from the newest to the oldest -- XNNPACK's subconvolution packs its sub-kernels
 * this way), 1: k descending (input rows ascending). */
void cf_deconv(int bias_first, int tap_order, int h, int kh, int stride, int cin, int cout, const float* w,
                         const float* b, const float* x, float* y) {
  const int hout = (h - 1) * stride + kh;
  for (int p = 0; p < hout; p++)
    for (int co = 0; co < cout; co++) {
      float acc = (bias_first && b) ? b[co] : 0.0f;
      for (int kk = 0; kk < kh; kk++) {
        const int k = tap_order ? kh - 1 - kk : kk;
        const int d = p - k;
        if (d < 0 || d % stride) continue;
        const int t = d / stride;
        if (t >= h) continue;
        for (int c = 0; c < cin; c++) acc = fmaf(x[(size_t)t * cin + c], w[((size_t)co * kh + k) * cin + c], acc);
      }
      y[(size_t)p * cout + co] = (bias_first || !b) ? acc : acc + b[co];
    }
}

/* cout == 1 transposed conv as THIS build's x86 micro-kernel sums it (found by probing; consistent with XNNPACK's
 * "nr2" GEMM configuration f32_igemm_minmax_ukernel_4x2c4__sse, which a convolution with fewer output channels than the
 * main kernel's NR is given): kr = 4 -- lane l of a 4-lane accumulator takes the input channels c == l (mod 4), taps
 * ascending; SSE has no FMA, so every term is a rounded product followed by a rounded add; the bias starts lane 0; the
 * lanes are reduced as (l0 + l2) + (l1 + l3).  fused = 1 replaces mul + add by fmaf (to show which it is). */
void cf_deconv_c4(int fused, int h, int kh, int stride, int cin, const float* w, const float* b, const float* x,
                            float* y) {
  const int hout = (h - 1) * stride + kh;
  for (int p = 0; p < hout; p++) {
    float acc[4] = {b ? b[0] : 0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < kh; k++) {
      const int d = p - k;
      if (d < 0 || d % stride) continue;
      const int t = d / stride;
      if (t >= h) continue;
      for (int c = 0; c < cin; c++) {
        const float xv = x[(size_t)t * cin + c], wv = w[(size_t)k * cin + c];
        if (fused) acc[c & 3] = fmaf(xv, wv, acc[c & 3]);
        else { const float pr = xv * wv; acc[c & 3] = acc[c & 3] + pr; }
      }
    }
    y[p] = (acc[0] + acc[2]) + (acc[1] + acc[3]);
  }
}

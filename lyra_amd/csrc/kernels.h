// kernels.h -- kernel parameter blocks and launch-side declarations (internal to liblyra_hip.so).
#pragma once
#include "lyra_dev.h"
#include "state_layout.h"

namespace lyra {

// ---- encoder ---------------------------------------------------------------------------------
struct EncS0P { ConvF first; DwF dw[3]; ConvF pw[3]; ConvF cv[3]; ConvF down; WarmRange warm; };
struct EncS1P { DwF dw[3]; ConvF pw[3]; ConvF cv[3]; ConvF down; WarmRange warm; };
struct EncS2P {
  DwF dw0; ConvF pw0;
  QP q_r0, dq_r0, q_x1, out;
  LreluQ lr[7];
  const int8_t* lr_lut;    // [7][256]   lrelu_q tabulated (model.hip lrelu_luts)
  const int32_t* add_lut;  // [2][2][256] ADD operand rescalings
  ConvQ r0b;
  DwQ dwq[2]; ConvQ pwq[2]; ConvQ cvq[2]; AddQ add[2];
  ConvQ down2, bott;
  int mode;
  WarmRange warm;
};

// code_bytes: size of the kernel's own machine code to pull into L2 at start (lyra_dev.h code_warm; 0 = skip)
// tile0: first tile of this launch (workgroup i works on tile tile0 + i; api.hip tile_div)
__global__ void enc_s0_kernel(const EncS0P* P, const int16_t* pcm, const int32_t* ids, int B, uint8_t* state, float* out0,
                              int code_bytes, int tile0);
__global__ void enc_s1_kernel(const EncS1P* P, const float* in0, const int32_t* ids, int B, uint8_t* state, float* out1,
                              int code_bytes, int tile0);
__global__ void enc_s2_kernel(const EncS2P* P, const float* in1, const int32_t* ids, int B, uint8_t* state, float* feats,
                              float* codes_dbg, int code_bytes, int tile0);
__global__ void enc_s2_dr_kernel(const EncS2P* P, const float* in1, const int32_t* ids, int B, uint8_t* state,
                                 float* feats, float* codes_dbg, int code_bytes, int tile0);   // gemmlowp double rounding
__global__ void enc_s2_xn_kernel(const EncS2P* P, const float* in1, const int32_t* ids, int B, uint8_t* state,
                                 float* feats, float* codes_dbg, int code_bytes, int tile0);   // XNNPACK QS8 arithmetic (default)
__global__ void enc_s2_bm_kernel(const EncS2P* P, const float* in1, const int32_t* ids, int B, uint8_t* state,
                                 float* feats, float* codes_dbg, int code_bytes, int tile0);   // TFLite builtin kernels, per-operator mixture
__global__ void enc_side_kernel(const EncS0P* P0, const EncS1P* P1, const EncS2P* P2, const int16_t* pcm, const int32_t* ids,
                                int B, uint8_t* st0, uint8_t* st1, uint8_t* st2, float* e0, float* e1, float* feats,
                                float* codes_dbg, int code_bytes);
__global__ void enc_side_dr_kernel(const EncS0P* P0, const EncS1P* P1, const EncS2P* P2, const int16_t* pcm,
                                   const int32_t* ids, int B, uint8_t* st0, uint8_t* st1, uint8_t* st2, float* e0, float* e1,
                                   float* feats, float* codes_dbg, int code_bytes);
__global__ void enc_side_xn_kernel(const EncS0P* P0, const EncS1P* P1, const EncS2P* P2, const int16_t* pcm,
                                   const int32_t* ids, int B, uint8_t* st0, uint8_t* st1, uint8_t* st2, float* e0, float* e1,
                                   float* feats, float* codes_dbg, int code_bytes);
size_t enc_side_lds_bytes();
__global__ void enc_s12_xn_kernel(const EncS1P* P1, const EncS2P* P2, const float* e0, const int32_t* ids, int B, uint8_t* st1,
                                  uint8_t* st2, float* e1, float* feats, float* codes_dbg, int code_bytes);
size_t enc_s12_lds_bytes();
size_t enc_s0_lds_bytes(); int enc_s0_streams_per_wg(); int enc_s0_threads();
size_t enc_s1_lds_bytes(); int enc_s1_streams_per_wg(); int enc_s1_threads();
size_t enc_s2_lds_bytes(); int enc_s2_streams_per_wg();

// ---- decoder ---------------------------------------------------------------------------------
// int8 transpose conv k4/s2 as a GEMM [rows][K=128] x [K][N = 4 taps x 64]; N tiles ordered [co tile][tap].
// zfold[tap*64+co] = -zin * sum_c w[co][tap][c] (the GEMM runs on raw codes); bias added once per output row.
struct TconvQ { const i32x4* w; const int32_t* zfold; const int32_t* bias; int32_t M, sh, zout; };
struct DecS0P {
  ConvF head;                 // conv k3 g4 (fp32): 4 groups, K = 3 taps x 16, N = 128
  QP q0;                      // QUANTIZE after the float LeakyReLU
  TconvQ up0[4]; QP up0_dq[4]; const float* up0_sub[4];
  QP q1;                      // QUANTIZE of lrelu(x164)
  DwQ dwq[3]; ConvQ pwq[3]; ConvQ cvq[3];
  LreluQ lr[6]; AddQ add[2];
  const int8_t* lr_lut;       // [6][256]
  const int32_t* add_lut;     // [2][2][256]
  QP dq_r0, q3;               // DEQUANTIZE of resblock-0 conv out; QUANTIZE of (conv + float skip)
  TconvQ up1[2]; QP up1_dq[2]; const float* up1_sub[2];
  int mode;
  WarmRange warm;
};
struct DecS1P { DwF dw[3]; ConvF pw[3]; ConvF cv[3]; ConvF up; const float* up_sub; WarmRange warm; };
struct DecS2P { DwF dw[3]; ConvF pw[3]; ConvF cv[3]; ConvF up; float up_sub; WarmRange warm; };

__global__ void dec_s0_kernel(const DecS0P* P, const float* feats, const int32_t* ids, int B, uint8_t* state, float* out0,
                              const uint8_t* packets, int num_stages, const float* cb, int code_bytes, int tile0);
__global__ void dec_s0_dr_kernel(const DecS0P* P, const float* feats, const int32_t* ids, int B, uint8_t* state,
                                 float* out0, const uint8_t* packets, int num_stages, const float* cb, int code_bytes, int tile0);
__global__ void dec_s0_xn_kernel(const DecS0P* P, const float* feats, const int32_t* ids, int B, uint8_t* state,
                                 float* out0, const uint8_t* packets, int num_stages, const float* cb, int code_bytes, int tile0);
__global__ void dec_s0_bm_kernel(const DecS0P* P, const float* feats, const int32_t* ids, int B, uint8_t* state,
                                 float* out0, const uint8_t* packets, int num_stages, const float* cb, int code_bytes, int tile0);
__global__ void dec_s1_kernel(const DecS1P* P, const float* in0, const int32_t* ids, int B, uint8_t* state, float* out1,
                              int code_bytes, int tile0);
__global__ void dec_s2_kernel(const DecS2P* P, const float* in1, const int32_t* ids, int B, uint8_t* state, int16_t* pcm,
                              int code_bytes, int tile0);
__global__ void dec_side_kernel(const DecS0P* P0, const DecS1P* P1, const DecS2P* P2, const float* feats, const int32_t* ids,
                                int B, uint8_t* st0, uint8_t* st1, uint8_t* st2, float* d0, float* d1, int16_t* pcm,
                                const uint8_t* packets, int num_stages, const float* cb, int code_bytes);
__global__ void dec_side_dr_kernel(const DecS0P* P0, const DecS1P* P1, const DecS2P* P2, const float* feats,
                                   const int32_t* ids, int B, uint8_t* st0, uint8_t* st1, uint8_t* st2, float* d0, float* d1,
                                   int16_t* pcm, const uint8_t* packets, int num_stages, const float* cb, int code_bytes);
__global__ void dec_side_xn_kernel(const DecS0P* P0, const DecS1P* P1, const DecS2P* P2, const float* feats,
                                   const int32_t* ids, int B, uint8_t* st0, uint8_t* st1, uint8_t* st2, float* d0, float* d1,
                                   int16_t* pcm, const uint8_t* packets, int num_stages, const float* cb, int code_bytes);
size_t dec_side_lds_bytes();
__global__ void dec_s01_xn_kernel(const DecS0P* P0, const DecS1P* P1, const float* feats, const int32_t* ids, int B, uint8_t* st0,
                                  uint8_t* st1, float* d0, float* d1, const uint8_t* packets, int num_stages, const float* cb,
                                  int code_bytes);
size_t dec_s01_lds_bytes();
size_t dec_s0_lds_bytes(); int dec_s0_streams_per_wg();
size_t dec_s1_lds_bytes(); int dec_s1_streams_per_wg(); int dec_s1_threads();
size_t dec_s2_lds_bytes(); int dec_s2_streams_per_wg(); int dec_s2_threads();

// ---- RVQ / packets / log-mel / state ---------------------------------------------------------------
// cb: codebooks, natural layout [46][16][64]
// mask_ids (optional): frame i is skipped (empty packet, packet_bytes[i] = 0) where mask_ids[i] < 0
// cbn: [46][16] |c|^2 per codeword, then [46] 2^-14 max |c|^2 per stage (model.hip); stats (optional): [0] frame-stages that took
// the exact chain, [1] wavefront-stages that entered it
__global__ void rvq_encode_kernel(const float* cb, const float* cbn, const float* feats, int B, int num_stages, int32_t* indices,
                                  uint8_t* packets, const int32_t* mask_ids, int32_t* packet_bytes, unsigned* stats);
__global__ void rvq_encode_chain_kernel(const float* cb, const float* feats, int B, int num_stages, int32_t* indices,
                                  uint8_t* packets, const int32_t* mask_ids, int32_t* packet_bytes);
__global__ void rvq_encode_wide_kernel(const float* cb, const float* feats, int B, int num_stages, int32_t* indices,
                                  uint8_t* packets, const int32_t* mask_ids, int32_t* packet_bytes);
__global__ void rvq_decode_kernel(const float* cb, const int32_t* indices, const uint8_t* packets, int num_stages,
                                  int B, float* feats);
struct MelP { const double* hann; const double* tw_re; const double* tw_im; const int* band; const double* w;
              const double* wsum;   // [160] total forward weight of every mel band (comfort-noise inverse mel)
              const double* tw4_re; const double* tw4_im;   // [768] W_1024^j, radix-4 log-mel FFT
              int start, end; };
// NoiseEstimator::Create's constants (noise_estimator.cc:96-124): round(1 s / 20 ms), 0.5^(20 ms / 0.7 s), 0.5^(20 ms / 1 s)
struct NoiseP { int hops_per_update; float max_smoothing, bound_decay; };
// noise_tail: continue with the NoiseEstimator update of the same hop (state must then be a NoiseEstimator region)
__global__ void logmel_kernel(const MelP* P, const int16_t* pcm, const int32_t* ids, int B, uint8_t* state, int stride,
                              int prev_off, float* mel, int noise_tail, NoiseP NP, int32_t* is_noise_out,
                              int32_t* masked_ids);
__global__ void noise_update_kernel(NoiseP P, const int32_t* ids, int B, uint8_t* state, const float* mel,
                                    int32_t* is_noise_out, int32_t* masked_ids);
// Resampler (lyra/resampler.cc): out/in = up/down, coef[phase][tap] oldest tap first (oracle lo_resampler_design)
struct ResampleP { int up, down; float coef[3][40]; };
int resample_streams_per_wg();
size_t resample_lds_bytes(int n_in);
// in_stride / out_stride: samples between consecutive streams' rows (>= n_in / n_out: a chunk of longer rows)
__global__ void resample_kernel(ResampleP P, const int32_t* ids, int B, uint8_t* state, const int16_t* in, int n_in,
                                int in_stride, int16_t* out, int n_out, int out_stride);
// ---- device half of BatchLyraDecoder (host/lyra_batch_codec.cc): hop buffers stay on the device, the host keeps ints --
// One slice of LyraDecoder::DecodeSamplesInternal's loop for one stream (lyra_decoder.cc:228-315): which samples of the
// conditioned generative-model hop and of the comfort-noise hop go where, and how they are cross-faded
// (MaybeOverlapAndInsert, :342-373).  Layout = lyra_hip_twin_slice (include/lyra_hip.h).
struct TwinSlice { int32_t id, gan_off, gen_n, cng_off, cng_n, fade, fade_dir, out_off, noise_row; };
constexpr int TWIN_FADE_LO = -640, TWIN_FADE_N = 1921;   // fade_progress range the weight table covers
__global__ void twin_scatter_kernel(const int16_t* src, const int32_t* ids, int B, int16_t* dst);
__global__ void twin_assemble_kernel(const TwinSlice* slices, int B, const int16_t* gan, const int16_t* cng,
                                     const float* fade_w, int16_t* out, int out_stride, int16_t* noise_dense);
__global__ void cng_kernel(const MelP* P, unsigned long long seed, const int32_t* ids, int B, uint8_t* state,
                           const uint8_t* noise_state, const float* features, int16_t* pcm);
__global__ void noise_read_kernel(const int32_t* ids, int B, const uint8_t* state, int field_off, float* out);
size_t logmel_lds_bytes();
size_t cng_lds_bytes();
struct ResetP { int8_t e_r2_1, e_r2_2, e_d2, e_bott, d_r0_0, d_r0_1, d_r0_2; };
// region base pointers and per-stream slot sizes (state_layout.h), filled on the host, passed by value
struct StateMap { uint8_t* base[st::R_COUNT]; int bytes[st::R_COUNT]; };
__global__ void reset_kernel(const ResetP* P, const int32_t* ids, int n, int all, StateMap sm);

}  // namespace lyra

// state_layout.h -- per-stream codec state in HBM, one REGION per kernel: [region][stream][bytes].
//
// The reference keeps this state in TFLite resource variables (14 tensors in the encoder graph,
// 18 in the decoder graph) and rewrites every tensor on every Invoke.  Here:
//  * every stage kernel owns one region holding exactly the tensors it reads and writes, laid out as
//    [max_streams][REGION_BYTES]: a kernel's page footprint is its own region (B = 4096: 27-61 MB)
//    instead of ~10-20 % of every page of one 265 MB block-per-stream array;
//  * fp32 states hold the same values as the graph's, channel order AT16 (lyra_dev.h);
//  * states inside the int8 regions are stored as int8 codes (lossless: the graph re-quantises
//    them with the producer's own scale every step), initial value = the tensor's zero point;
//  * depthwise-conv histories that are longer than one frame's rows (T < 2*dilation) are RINGS of
//    R = 2*dilation rows indexed by (phase * T + t) mod R, so a step writes T rows instead of
//    shifting R; `phase` = frames processed mod 18 (periods 2, 3 and 9 all divide 18).  Every region
//    that holds a ring has its own phase word (first word of the region), advanced by its own kernel;
//  * histories with T >= R are simply replaced.
// Offsets below are relative to the start of a stream's slot in its region.
#pragma once
#include <stdint.h>

namespace lyra {
namespace st {

constexpr int PHASE_MOD = 18;
constexpr int PHASE = 0;       // uint32, first word of a region slot (regions with rings)
constexpr int HDR = 64;
constexpr int align256(int n) { return (n + 255) / 256 * 256; }

enum Region { R_E0, R_E1, R_E2, R_D0, R_D1, R_D2, R_MEL, R_NOISE_E, R_NOISE_D, R_RS_E, R_RS_D, R_CNG, R_COUNT };

// ---- R_E0: encoder stage 0 (enc_s0_kernel) ---------------------------------------------------------
constexpr int E_FIRST = 0;                         // f32[48] natural order
constexpr int E_R0_0 = E_FIRST + 48 * 4;           // f32[2][64]
constexpr int E_R0_1 = E_R0_0 + 2 * 64 * 4;        // f32[6][64]
constexpr int E_R0_2 = E_R0_1 + 6 * 64 * 4;        // f32[18][64]
constexpr int E_D0 = E_R0_2 + 18 * 64 * 4;         // f32[5][64]
constexpr int E0_BYTES = align256(E_D0 + 5 * 64 * 4);

// ---- R_E1: encoder stage 1 (enc_s1_kernel) ---------------------------------------------------------
constexpr int E_R1_0 = HDR;                        // f32[2][128]
constexpr int E_R1_1 = E_R1_0 + 2 * 128 * 4;       // f32[6][128]   ring
constexpr int E_R1_2 = E_R1_1 + 6 * 128 * 4;       // f32[18][128]  ring
constexpr int E_D1 = E_R1_2 + 18 * 128 * 4;        // f32[2][128]
constexpr int E1_BYTES = align256(E_D1 + 2 * 128 * 4);

// ---- R_E2: encoder stage 2 (enc_s2_kernel) ---------------------------------------------------------
constexpr int E_R2_0 = HDR;                        // f32[2][256]
constexpr int E_R2_1 = E_R2_0 + 2 * 256 * 4;       // i8[6][256]    ring
constexpr int E_R2_2 = E_R2_1 + 6 * 256;           // i8[18][256]   ring
constexpr int E_D2 = E_R2_2 + 18 * 256;            // i8[2][256]
constexpr int E_BOTT = E_D2 + 2 * 256;             // i8[2][512]    ring (R=2, T=1)
constexpr int E2_BYTES = align256(E_BOTT + 2 * 512);

// ---- R_D0: decoder stage 0 (dec_s0_kernel) ---------------------------------------------------------
constexpr int D_HEAD = HDR;                        // f32[2][64]    ring (R=2, T=1)
constexpr int D_UP0 = D_HEAD + 2 * 64 * 4;         // f32[4 groups][2][64]
constexpr int D_R0_0 = D_UP0 + 4 * 2 * 64 * 4;     // i8[2][256]
constexpr int D_R0_1 = D_R0_0 + 2 * 256;           // i8[6][256]    ring
constexpr int D_R0_2 = D_R0_1 + 6 * 256;           // i8[18][256]   ring
constexpr int D_UP1 = D_R0_2 + 18 * 256;           // f32[2 groups][2][64]
constexpr int D0_BYTES = align256(D_UP1 + 2 * 2 * 64 * 4);

// ---- R_D1: decoder stage 1 (dec_s1_kernel) ---------------------------------------------------------
constexpr int D_R1_0 = HDR;                        // f32[2][128]
constexpr int D_R1_1 = D_R1_0 + 2 * 128 * 4;       // f32[6][128]   ring
constexpr int D_R1_2 = D_R1_1 + 6 * 128 * 4;       // f32[18][128]  ring
constexpr int D_UP2 = D_R1_2 + 18 * 128 * 4;       // f32[5][64]
constexpr int D1_BYTES = align256(D_UP2 + 5 * 64 * 4);

// ---- R_D2: decoder stage 2 (dec_s2_kernel) ---------------------------------------------------------
constexpr int D_R2_0 = 0;                          // f32[2][64]
constexpr int D_R2_1 = D_R2_0 + 2 * 64 * 4;        // f32[6][64]
constexpr int D_R2_2 = D_R2_1 + 6 * 64 * 4;        // f32[18][64]
constexpr int D_UP3 = D_R2_2 + 18 * 64 * 4;        // f32[48]
constexpr int D2_BYTES = align256(D_UP3 + 48 * 4);

// ---- R_MEL: log-mel front end (logmel_kernel) -------------------------------------------------------
constexpr int M_PREV = 0;                          // i16[320] previous hop
constexpr int MEL_BYTES = align256(320 * 2);

// ---- R_NOISE_E / R_NOISE_D: NoiseEstimator of the encoder (DTX) and of the decoder (lyra/noise_estimator.h:96-112:
//      one instance each, with its own log-mel extractor and therefore its own previous-hop history) -------------------
constexpr int N_INIT = 0;                          // int32: smoothed_power_ is non-empty
constexpr int N_HOPS = 4;                          // int32: num_hops_received_
constexpr int N_IS_NOISE = 8;                      // int32: is_noise_ (initially true)
constexpr int N_PREV = HDR;                        // i16[320] previous hop of the log-mel window
constexpr int N_SMOOTH = N_PREV + 320 * 2;         // f32[160] smoothed_power_
constexpr int N_SQ = N_SMOOTH + 160 * 4;           // f32[160] squared_smoothed_power_
constexpr int N_TMPMIN = N_SQ + 160 * 4;           // f32[160] tmp_min_smoothed_power_
constexpr int N_EST = N_TMPMIN + 160 * 4;          // f32[160] noise_estimate_
constexpr int N_BOUND = N_EST + 160 * 4;           // f32[160] noise_bound_
constexpr int NOISE_BYTES = align256(N_BOUND + 160 * 4);

// ---- R_RS_E / R_RS_D: the encoder's (external rate -> 16 kHz) and the decoder's (16 kHz -> external rate) Resampler
//      (lyra/resampler.h; one per codec object) ------------------------------------------------------------------------
constexpr int RS_RADIUS = 17;                      // kernel radius in input samples (resampler.cc:33-38)
constexpr int RS_TAPS = 2 * RS_RADIUS + 1;
constexpr int RS_IN_POS = 0;                       // int32: input samples consumed so far (decimation phase)
constexpr int RS_HIST = 16;                        // f32[34]: the last RS_TAPS - 1 input samples
constexpr int RS_BYTES = align256(RS_HIST + (RS_TAPS - 1) * 4);

// ---- R_CNG: ComfortNoiseGenerator (lyra/comfort_noise_generator.h): overlap-add tail of the inverse STFT --------------
constexpr int C_HOP = 0;                           // uint64: hops generated (random-phase counter)
constexpr int C_OLA = 64;                          // f64[1024] overlap-add accumulator, [0, 320) = next hop
constexpr int CNG_BYTES = align256(C_OLA + 1024 * 8);

constexpr int REGION_BYTES[R_COUNT] = {E0_BYTES, E1_BYTES, E2_BYTES, D0_BYTES, D1_BYTES, D2_BYTES, MEL_BYTES,
                                        NOISE_BYTES, NOISE_BYTES, RS_BYTES, RS_BYTES, CNG_BYTES};
constexpr int BYTES = E0_BYTES + E1_BYTES + E2_BYTES + D0_BYTES + D1_BYTES + D2_BYTES + MEL_BYTES + 2 * NOISE_BYTES +
                      2 * RS_BYTES + CNG_BYTES;

static_assert(E_R0_0 % 16 == 0 && E_R1_0 % 16 == 0 && E_R2_1 % 16 == 0 && D_HEAD % 16 == 0 && D_R0_0 % 16 == 0 &&
                  D_UP1 % 16 == 0 && D_R1_0 % 16 == 0 && D_UP3 % 16 == 0,
              "16-byte alignment of vector-accessed state tensors");

}  // namespace st
}  // namespace lyra

// state_layout.h -- per-stream codec state in HBM (one contiguous block per stream id).
//
// The reference keeps this state in TFLite resource variables (14 tensors in the encoder graph,
// 18 in the decoder graph) and rewrites every tensor on every Invoke.  Here:
//  * fp32 states hold the same values, channel order AT16 (lyra_dev.h);
//  * states inside the int8 regions are stored as int8 codes (lossless: the graph re-quantises
//    them with the producer's own scale every step), initial value = the tensor's zero point;
//  * depthwise-conv histories that are longer than one frame's rows (T < 2*dilation) are RINGS of
//    R = 2*dilation rows indexed by (phase * T + t) mod R, so a step writes T rows instead of
//    shifting R; `phase` = frames processed mod 18 (periods 2, 3 and 9 all divide 18);
//  * histories with T >= R are simply replaced.
#pragma once
#include <stdint.h>

namespace lyra {
namespace st {

constexpr int PHASE_MOD = 18;

// ---- header --------------------------------------------------------------------------------
constexpr int ENC_PHASE = 0;   // uint32
constexpr int DEC_PHASE = 4;   // uint32
constexpr int HDR = 64;

// ---- encoder ---------------------------------------------------------------------------------
constexpr int E_FIRST = HDR;                       // f32[48] natural order
constexpr int E_R0_0 = E_FIRST + 48 * 4;           // f32[2][64]
constexpr int E_R0_1 = E_R0_0 + 2 * 64 * 4;        // f32[6][64]
constexpr int E_R0_2 = E_R0_1 + 6 * 64 * 4;        // f32[18][64]
constexpr int E_D0 = E_R0_2 + 18 * 64 * 4;         // f32[5][64]
constexpr int E_R1_0 = E_D0 + 5 * 64 * 4;          // f32[2][128]
constexpr int E_R1_1 = E_R1_0 + 2 * 128 * 4;       // f32[6][128]   ring
constexpr int E_R1_2 = E_R1_1 + 6 * 128 * 4;       // f32[18][128]  ring
constexpr int E_D1 = E_R1_2 + 18 * 128 * 4;        // f32[2][128]
constexpr int E_R2_0 = E_D1 + 2 * 128 * 4;         // f32[2][256]
constexpr int E_R2_1 = E_R2_0 + 2 * 256 * 4;       // i8[6][256]    ring
constexpr int E_R2_2 = E_R2_1 + 6 * 256;           // i8[18][256]   ring
constexpr int E_D2 = E_R2_2 + 18 * 256;            // i8[2][256]
constexpr int E_BOTT = E_D2 + 2 * 256;             // i8[2][512]    ring (R=2, T=1)
constexpr int E_END = E_BOTT + 2 * 512;

// ---- decoder ---------------------------------------------------------------------------------
constexpr int D_HEAD = E_END;                      // f32[2][64]    ring (R=2, T=1)
constexpr int D_UP0 = D_HEAD + 2 * 64 * 4;         // f32[4 groups][2][64]
constexpr int D_R0_0 = D_UP0 + 4 * 2 * 64 * 4;     // i8[2][256]
constexpr int D_R0_1 = D_R0_0 + 2 * 256;           // i8[6][256]    ring
constexpr int D_R0_2 = D_R0_1 + 6 * 256;           // i8[18][256]   ring
constexpr int D_UP1 = D_R0_2 + 18 * 256;           // f32[2 groups][2][64]
constexpr int D_R1_0 = D_UP1 + 2 * 2 * 64 * 4;     // f32[2][128]
constexpr int D_R1_1 = D_R1_0 + 2 * 128 * 4;       // f32[6][128]   ring
constexpr int D_R1_2 = D_R1_1 + 6 * 128 * 4;       // f32[18][128]  ring
constexpr int D_UP2 = D_R1_2 + 18 * 128 * 4;       // f32[5][64]
constexpr int D_R2_0 = D_UP2 + 5 * 64 * 4;         // f32[2][64]
constexpr int D_R2_1 = D_R2_0 + 2 * 64 * 4;        // f32[6][64]
constexpr int D_R2_2 = D_R2_1 + 6 * 64 * 4;        // f32[18][64]
constexpr int D_UP3 = D_R2_2 + 18 * 64 * 4;        // f32[48]
constexpr int D_END = D_UP3 + 48 * 4;

// ---- log-mel ---------------------------------------------------------------------------------
constexpr int M_PREV = D_END;                      // i16[320] previous hop
constexpr int M_END = M_PREV + 320 * 2;

constexpr int BYTES = (M_END + 255) / 256 * 256;   // per-stream block, 256-byte aligned

static_assert(E_FIRST % 16 == 0 && E_R2_1 % 16 == 0 && D_HEAD % 16 == 0 && D_R0_0 % 16 == 0 && D_UP1 % 16 == 0 &&
                  M_PREV % 16 == 0,
              "16-byte alignment of vector-accessed state tensors");

}  // namespace st
}  // namespace lyra

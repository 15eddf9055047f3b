// fake_lyra_hip_codec.cc -- TEST INFRASTRUCTURE: a CPU stand-in for the C-ABI entry points BatchLyraEncoder /
// BatchLyraDecoder (lyra_amd/host/lyra_batch_codec.cc) call, so that their host logic -- resampling bookkeeping, DTX,
// per-stream packet FIFO, hop-straddling DecodeSamples, concealment / comfort-noise / fade state machine, noise-estimator
// updates -- runs in the CPU test suite against oracle/lyra_codec_model.py driven by the SAME fake arithmetic
// (tests/host_stub/fake_kit.py).  The fakes are integer formulas with small per-stream counters, so that a call made
// for the wrong stream, at the wrong time or a wrong number of times changes the output.  Never linked into the product.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/lyra_hip.h"

struct PerStream { long enc_hops = 0, dec_hops = 0, cng_hops = 0, noise_calls[2] = {0, 0}; };
struct lyra_hip_ctx { int max_streams; std::map<int32_t, PerStream> st; };

namespace {
bool is_noise_hop(const int16_t* pcm) {
  for (int i = 0; i < 320; ++i)
    if (std::abs((int)pcm[i]) >= 64) return false;
  return true;
}
void fake_packet(PerStream& s, const int16_t* pcm, int nbytes, uint8_t* pk) {
  for (int j = 0; j < nbytes; ++j) pk[j] = (uint8_t)((int)pcm[(j * 13) % 320] + 31 * j + (int)s.enc_hops);
  s.enc_hops++;
}
}  // namespace

extern "C" {
int lyra_hip_create(const char*, int, int max_streams, int, lyra_hip_ctx** out) {
  *out = new lyra_hip_ctx{max_streams, {}};
  return 0;
}
void lyra_hip_destroy(lyra_hip_ctx* c) { delete c; }
const char* lyra_hip_last_error(const lyra_hip_ctx*) { return "fake"; }

int lyra_hip_resample(lyra_hip_ctx*, int, const int32_t*, int B, const int16_t* in, int n_in, int in_rate, int out_rate,
                      int16_t* out) {
  const int n_out = n_in * out_rate / in_rate;
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < n_out; ++j) out[b * n_out + j] = in[b * n_in + (int)((long)j * in_rate / out_rate)];
  return 0;
}
int lyra_hip_noise_receive(lyra_hip_ctx* c, int side, const int32_t* ids, int B, const int16_t* pcm, int32_t* is_noise) {
  for (int b = 0; b < B; ++b) {
    c->st[ids[b]].noise_calls[side]++;
    is_noise[b] = is_noise_hop(pcm + b * 320) ? 1 : 0;
  }
  return 0;
}
int lyra_hip_encode(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, int num_bits, uint8_t* packets) {
  const int nbytes = (num_bits + 7) / 8;
  for (int b = 0; b < B; ++b) fake_packet(c->st[ids[b]], pcm + b * 320, nbytes, packets + b * nbytes);
  return 0;
}
int lyra_hip_set_encoder_sample_rate(lyra_hip_ctx*, int) { return 0; }
int lyra_hip_encode_dtx(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, int num_bits, uint8_t* packets,
                        int32_t* packet_bytes) {
  const int nbytes = (num_bits + 7) / 8;
  for (int b = 0; b < B; ++b) {
    PerStream& s = c->st[ids[b]];
    s.noise_calls[LYRA_HIP_SIDE_ENCODER]++;
    if (is_noise_hop(pcm + b * 320)) {
      packet_bytes[b] = 0;
      std::memset(packets + b * nbytes, 0, nbytes);
    } else {
      packet_bytes[b] = nbytes;
      fake_packet(s, pcm + b * 320, nbytes, packets + b * nbytes);
    }
  }
  return 0;
}
int lyra_hip_decode(lyra_hip_ctx* c, const int32_t* ids, int B, const uint8_t* packets, int num_bits, int16_t* pcm) {
  const int nbytes = (num_bits + 7) / 8;
  for (int b = 0; b < B; ++b) {
    PerStream& s = c->st[ids[b]];
    for (int i = 0; i < 320; ++i) pcm[b * 320 + i] = (int16_t)((int)packets[b * nbytes + i % nbytes] * 64 + i + 7 * (int)s.dec_hops);
    s.dec_hops++;
  }
  return 0;
}
int lyra_hip_generate(lyra_hip_ctx* c, const int32_t* ids, int B, const float* features, int16_t* pcm) {
  for (int b = 0; b < B; ++b) {
    PerStream& s = c->st[ids[b]];
    for (int i = 0; i < 320; ++i) pcm[b * 320 + i] = (int16_t)(-500 + i + 7 * (int)s.dec_hops + (int)features[b * 64]);
    s.dec_hops++;
  }
  return 0;
}
int lyra_hip_comfort_noise(lyra_hip_ctx* c, const int32_t* ids, int B, const float* features, int16_t* pcm) {
  if (features) return LYRA_HIP_EINVAL;   // the batch decoder always asks for the decoder-side estimate
  for (int b = 0; b < B; ++b) {
    PerStream& s = c->st[ids[b]];
    for (int i = 0; i < 320; ++i)
      pcm[b * 320 + i] = (int16_t)(2000 + (i & 31) + 3 * (int)s.noise_calls[LYRA_HIP_SIDE_DECODER] + 11 * (int)s.cng_hops);
    s.cng_hops++;
  }
  return 0;
}
}

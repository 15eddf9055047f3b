// fake_lyra_hip_codec.cc -- TEST INFRASTRUCTURE: a CPU stand-in for the C-ABI entry points BatchLyraEncoder /
// BatchLyraDecoder (lyra_amd/host/lyra_batch_codec.cc) call, so that their host logic -- resampling bookkeeping, DTX,
// per-stream packet FIFO, hop-straddling DecodeSamples, concealment / comfort-noise / fade state machine, noise-estimator
// updates -- runs in the CPU test suite against oracle/lyra_codec_model.py driven by the SAME fake arithmetic
// (tests/host_stub/fake_kit.py).  The fakes are integer formulas with small per-stream counters, so that a call made
// for the wrong stream, at the wrong time or a wrong number of times changes the output.  Never linked into the product.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <vector>

#include "../../include/lyra_hip.h"

struct PerStream {
  long enc_hops = 0, dec_hops = 0, cng_hops = 0, noise_calls[2] = {0, 0};
  int16_t gan[320] = {}, cng[320] = {};   // "Decoder twin": the conditioned hops that the real library keeps on the device
};
struct lyra_hip_ctx {
  int max_streams;
  std::map<int32_t, PerStream> st;
  int out_n = 0;                          // the request being assembled
  std::vector<int16_t> out;               // [max_streams][out_n]
  std::vector<std::vector<int16_t>> noise_rows;
  // pipelined calls: results of begun calls, oldest first (at most two in flight)
  struct Enc { std::vector<uint8_t> pk; std::vector<int32_t> len; };
  std::deque<Enc> enc_q;
  std::deque<std::vector<int16_t>> fetch_q;
};

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
  *out = new lyra_hip_ctx{};
  (*out)->max_streams = max_streams;
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
int lyra_hip_set_stream_priorities(lyra_hip_ctx*, int, int, int) { return 0; }
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

// ---- "Decoder twin" (include/lyra_hip.h): same formulas as lyra_hip_decode / _generate / _comfort_noise above, the hops
// kept per stream, slices cut and cross-faded as MaybeOverlapAndInsert writes it (lyra_decoder.cc:342-373) ------------------
int lyra_hip_twin_decode(lyra_hip_ctx* c, const int32_t* ids, int B, const uint8_t* packets, int num_bits) {
  const int nbytes = (num_bits + 7) / 8;
  for (int b = 0; b < B; ++b) {
    PerStream& s = c->st[ids[b]];
    for (int i = 0; i < 320; ++i) s.gan[i] = (int16_t)((int)packets[b * nbytes + i % nbytes] * 64 + i + 7 * (int)s.dec_hops);
    s.dec_hops++;
  }
  return 0;
}
int lyra_hip_twin_conceal(lyra_hip_ctx* c, const int32_t* ids, int B) {
  for (int b = 0; b < B; ++b) {
    PerStream& s = c->st[ids[b]];
    for (int i = 0; i < 320; ++i) s.gan[i] = (int16_t)(-500 + i + 7 * (int)s.dec_hops);
    s.dec_hops++;
  }
  return 0;
}
int lyra_hip_twin_comfort_noise(lyra_hip_ctx* c, const int32_t* ids, int B) {
  for (int b = 0; b < B; ++b) {
    PerStream& s = c->st[ids[b]];
    for (int i = 0; i < 320; ++i)
      s.cng[i] = (int16_t)(2000 + (i & 31) + 3 * (int)s.noise_calls[LYRA_HIP_SIDE_DECODER] + 11 * (int)s.cng_hops);
    s.cng_hops++;
  }
  return 0;
}
int lyra_hip_twin_assemble(lyra_hip_ctx* c, const lyra_hip_twin_slice* sl, int B, int out_samples) {
  if (c->out_n == 0) {
    c->out_n = out_samples;
    c->out.assign((size_t)c->max_streams * out_samples, 0);
  } else if (c->out_n != out_samples) {
    return LYRA_HIP_EINVAL;
  }
  c->noise_rows.clear();
  for (int b = 0; b < B; ++b) {
    const lyra_hip_twin_slice& x = sl[b];
    PerStream& s = c->st[x.id];
    int16_t* o = &c->out[(size_t)x.id * out_samples + x.out_off];
    const int n = x.gen_n > x.cng_n ? x.gen_n : x.cng_n;
    int fade = x.fade;
    for (int i = 0; i < n; ++i) {
      if (x.cng_n == 0) o[i] = s.gan[x.gan_off + i];
      else if (x.gen_n == 0) o[i] = s.cng[x.cng_off + i];
      else {
        const float w = (1.f + std::cos(fade * M_PI / 640)) / 2.f;
        o[i] = static_cast<int16_t>(s.gan[x.gan_off + i] * w + s.cng[x.cng_off + i] * (1.f - w));
        fade += x.fade_dir;
      }
    }
    if (x.noise_row >= 0) {
      if ((size_t)x.noise_row >= c->noise_rows.size()) c->noise_rows.resize(x.noise_row + 1);
      c->noise_rows[x.noise_row].assign(s.gan, s.gan + 320);
    }
  }
  return 0;
}
int lyra_hip_twin_noise(lyra_hip_ctx* c, const int32_t* ids, int B) {
  if ((size_t)B != c->noise_rows.size()) return LYRA_HIP_EINVAL;
  for (int b = 0; b < B; ++b) c->st[ids[b]].noise_calls[LYRA_HIP_SIDE_DECODER]++;
  return 0;
}
// pipelined forms: computed at begin(), handed over at end() -- the host twin must pair them in order, two deep at most
int lyra_hip_encode_begin(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, int rate, int num_bits, int dtx) {
  if (c->enc_q.size() >= 2) return LYRA_HIP_EINVAL;
  const int nbytes = (num_bits + 7) / 8, n_ext = rate / 50;
  std::vector<int16_t> p16((size_t)B * 320);
  if (rate == 16000) std::memcpy(p16.data(), pcm, p16.size() * 2);
  else lyra_hip_resample(c, LYRA_HIP_SIDE_ENCODER, ids, B, pcm, n_ext, rate, 16000, p16.data());
  lyra_hip_ctx::Enc e;
  e.pk.resize((size_t)B * nbytes);
  e.len.assign(B, nbytes);
  const int rc = dtx ? lyra_hip_encode_dtx(c, ids, B, p16.data(), num_bits, e.pk.data(), e.len.data())
                     : lyra_hip_encode(c, ids, B, p16.data(), num_bits, e.pk.data());
  if (rc) return rc;
  c->enc_q.push_back(std::move(e));
  return 0;
}
int lyra_hip_encode_end(lyra_hip_ctx* c, uint8_t* packets, int32_t* packet_bytes) {
  if (c->enc_q.empty()) return LYRA_HIP_EINVAL;
  lyra_hip_ctx::Enc& e = c->enc_q.front();
  std::memcpy(packets, e.pk.data(), e.pk.size());
  if (packet_bytes) std::memcpy(packet_bytes, e.len.data(), e.len.size() * 4);
  c->enc_q.pop_front();
  return 0;
}
int lyra_hip_twin_fetch(lyra_hip_ctx* c, int num_streams, int n, int out_rate, int16_t* out);
int lyra_hip_twin_fetch_begin(lyra_hip_ctx* c, int num_streams, int n, int out_rate) {
  if (c->fetch_q.size() >= 2) return LYRA_HIP_EINVAL;
  std::vector<int16_t> o((size_t)num_streams * (size_t)((long)n * out_rate / 16000));
  const int rc = lyra_hip_twin_fetch(c, num_streams, n, out_rate, o.data());
  if (rc) return rc;
  c->fetch_q.push_back(std::move(o));
  return 0;
}
int lyra_hip_twin_fetch_end(lyra_hip_ctx* c, int16_t* out) {
  if (c->fetch_q.empty()) return LYRA_HIP_EINVAL;
  if (!c->fetch_q.front().empty()) std::memcpy(out, c->fetch_q.front().data(), c->fetch_q.front().size() * 2);
  c->fetch_q.pop_front();
  return 0;
}
int lyra_hip_twin_fetch(lyra_hip_ctx* c, int num_streams, int n, int out_rate, int16_t* out) {
  if (n > 0 && c->out_n != n) return LYRA_HIP_EINVAL;
  const int n_out = (int)((long)n * out_rate / 16000);
  for (int s = 0; s < num_streams && n > 0; ++s)
    for (int j = 0; j < n_out; ++j) out[(size_t)s * n_out + j] = c->out[(size_t)s * n + (int)((long)j * 16000 / out_rate)];
  c->out_n = 0;
  return 0;
}
}

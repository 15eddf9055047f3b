// batch_bench.cc -- throughput of the PUBLIC-API twins, BatchLyraEncoder::Encode / BatchLyraDecoder::SetEncodedPackets +
// DecodeSamples (the batched LyraEncoder / LyraDecoder of lyra_encoder.cc:113-156, lyra_decoder.cc:172-315), driven from
// C++ with HOST buffers the way cli_example/encoder_main_lib.cc:62-93 / decoder_main_lib.cc:94-140 drive the reference:
//   batch_bench <model_dir> <num_streams> <sample_rate_hz> <bitrate> <loss_percent> <hops> [dtx]
// Three timed regions over the same synthetic audio (full-scale uniform int16, lyra_benchmark_lib.cc:233-239):
//   encode only; decode only (packets of the first region, `loss_percent` of them withheld per stream and hop);
//   encode -> decode pipelined on two host threads (the two twins own one context each, so the GPU overlaps them).
// ... and the same three regions through the two-deep pipelined halves of the calls (EncodeAsync / WaitEncoded,
// DecodeSamplesAsync / WaitDecoded: hop n + 1 is started before hop n is collected).
// One JSON line.  PCIe is inside these numbers (that is the point: bench.py's headline has the inputs resident in HBM).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "lyra_batch_codec.h"

using namespace chromemedia::codec;
using Clock = std::chrono::steady_clock;

static double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

int main(int argc, char** argv) {
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s model_dir num_streams sample_rate_hz bitrate loss_percent hops [dtx]\n", argv[0]);
    return 2;
  }
  const std::string model_dir = argv[1];
  const int n = std::atoi(argv[2]), rate = std::atoi(argv[3]), bitrate = std::atoi(argv[4]);
  const int loss = std::atoi(argv[5]), hops = std::atoi(argv[6]);
  const bool dtx = argc > 7 && std::atoi(argv[7]) != 0;
  const int hop_ext = rate / 50, ring = 8, warm = 5;
  auto enc = BatchLyraEncoder::Create(rate, 1, bitrate, dtx, model_dir, n);
  auto dec = BatchLyraDecoder::Create(rate, 1, model_dir, n);
  if (!enc || !dec) { std::fprintf(stderr, "creation failed\n"); return 1; }
  const int ps = enc->packet_size();
  std::mt19937 rng(0x4C797261u);
  std::uniform_int_distribution<int> full(-32768, 32767), pct(0, 99);
  std::vector<std::vector<int16_t>> pcm(ring, std::vector<int16_t>(static_cast<size_t>(n) * hop_ext));
  for (auto& h : pcm)
    for (auto& v : h) v = static_cast<int16_t>(full(rng));

  // ---- encode only ------------------------------------------------------------------------------------------------
  std::vector<std::vector<uint8_t>> packets;   // of every timed hop, for the decode-only region
  for (int t = 0; t < warm; ++t)
    if (!enc->Encode(absl::MakeConstSpan(pcm[t % ring]))) return 4;
  auto t0 = Clock::now();
  for (int t = 0; t < hops; ++t) {
    auto p = enc->Encode(absl::MakeConstSpan(pcm[t % ring]));
    if (!p) return 4;
    packets.push_back(std::move(*p));
  }
  const double enc_s = secs(t0, Clock::now());

  // ---- decode only ------------------------------------------------------------------------------------------------
  std::vector<int16_t> out(static_cast<size_t>(n) * hop_ext);
  std::vector<int32_t> live;
  std::vector<uint8_t> live_pk;
  long lost = 0;
  auto feed = [&](const std::vector<uint8_t>& pk) {
    if (loss == 0) return dec->SetEncodedPackets(absl::MakeConstSpan(pk));
    live.clear(); live_pk.clear();
    for (int s = 0; s < n; ++s) {
      if (pct(rng) < loss) { ++lost; continue; }
      live.push_back(s);
      live_pk.insert(live_pk.end(), pk.begin() + static_cast<size_t>(s) * ps, pk.begin() + static_cast<size_t>(s + 1) * ps);
    }
    return dec->SetEncodedPackets(absl::MakeConstSpan(live), absl::MakeConstSpan(live_pk));
  };
  for (int t = 0; t < warm; ++t)
    if (!feed(packets[t % hops]) || !dec->DecodeSamples(hop_ext, absl::Span<int16_t>(out.data(), out.size()))) return 5;
  lost = 0;
  t0 = Clock::now();
  for (int t = 0; t < hops; ++t)
    if (!feed(packets[t]) || !dec->DecodeSamples(hop_ext, absl::Span<int16_t>(out.data(), out.size()))) return 5;
  const double dec_s = secs(t0, Clock::now());
  const long lost_dec = lost;

  // ---- encode -> decode on two host threads -------------------------------------------------------------------------------
  std::deque<std::vector<uint8_t>> q;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<int> rc{0};
  t0 = Clock::now();
  std::thread producer([&] {
    for (int t = 0; t < hops && rc == 0; ++t) {
      auto p = enc->Encode(absl::MakeConstSpan(pcm[t % ring]));
      if (!p) { rc = 4; cv.notify_all(); return; }
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return q.size() < 4 || rc != 0; });
      q.push_back(std::move(*p));
      cv.notify_all();
    }
  });
  for (int t = 0; t < hops && rc == 0; ++t) {
    std::vector<uint8_t> p;
    {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return !q.empty() || rc != 0; });
      if (rc != 0) break;
      p = std::move(q.front());
      q.pop_front();
      cv.notify_all();
    }
    if (!feed(p) || !dec->DecodeSamples(hop_ext, absl::Span<int16_t>(out.data(), out.size()))) { rc = 5; cv.notify_all(); }
  }
  producer.join();
  const double both_s = secs(t0, Clock::now());
  if (rc != 0) return rc;

  // ---- the same three regions, pipelined two deep ------------------------------------------------------------------------
  std::vector<std::vector<uint8_t>> packets2;
  t0 = Clock::now();
  if (!enc->EncodeAsync(absl::MakeConstSpan(pcm[0]))) return 6;
  for (int t = 0; t < hops; ++t) {
    if (t + 1 < hops && !enc->EncodeAsync(absl::MakeConstSpan(pcm[(t + 1) % ring]))) return 6;
    auto p = enc->WaitEncoded();
    if (!p) return 6;
    if (t < 4) packets2.push_back(std::move(*p));
  }
  const double enc_p_s = secs(t0, Clock::now());
  (void)packets2;
  t0 = Clock::now();
  if (!feed(packets[0]) || !dec->DecodeSamplesAsync(hop_ext)) return 7;
  for (int t = 0; t < hops; ++t) {
    if (t + 1 < hops && (!feed(packets[t + 1]) || !dec->DecodeSamplesAsync(hop_ext))) return 7;
    if (!dec->WaitDecoded(absl::Span<int16_t>(out.data(), out.size()))) return 7;
  }
  const double dec_p_s = secs(t0, Clock::now());
  q.clear();
  rc = 0;
  t0 = Clock::now();
  std::thread producer2([&] {
    if (!enc->EncodeAsync(absl::MakeConstSpan(pcm[0]))) { rc = 6; cv.notify_all(); return; }
    for (int t = 0; t < hops && rc == 0; ++t) {
      if (t + 1 < hops && !enc->EncodeAsync(absl::MakeConstSpan(pcm[(t + 1) % ring]))) { rc = 6; cv.notify_all(); return; }
      auto p = enc->WaitEncoded();
      if (!p) { rc = 6; cv.notify_all(); return; }
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return q.size() < 4 || rc != 0; });
      q.push_back(std::move(*p));
      cv.notify_all();
    }
  });
  {
    auto next_packets = [&](std::vector<uint8_t>* p) {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return !q.empty() || rc != 0; });
      if (rc != 0) return false;
      *p = std::move(q.front());
      q.pop_front();
      cv.notify_all();
      return true;
    };
    std::vector<uint8_t> p;
    int begun = 0;
    for (int t = 0; t < hops && rc == 0; ++t) {
      while (begun < hops && begun < t + 2 && rc == 0) {   // request t + 1 is started before request t is collected
        if (!next_packets(&p)) break;
        if (!feed(p) || !dec->DecodeSamplesAsync(hop_ext)) { rc = 7; cv.notify_all(); break; }
        ++begun;
      }
      if (rc != 0) break;
      if (!dec->WaitDecoded(absl::Span<int16_t>(out.data(), out.size()))) { rc = 7; cv.notify_all(); }
    }
  }
  producer2.join();
  const double both_p_s = secs(t0, Clock::now());
  if (rc != 0) return rc;

  const double frames = static_cast<double>(n) * hops;
  std::printf("{\"what\": \"BatchLyraEncoder / BatchLyraDecoder through the C++ API, host buffers (PCIe included)\", "
              "\"streams\": %d, \"sample_rate_hz\": %d, \"bitrate\": %d, \"dtx\": %s, \"loss_percent\": %d, \"hops\": %d, "
              "\"packets_withheld_in_decode_region\": %ld, "
              "\"encode_frames_per_s\": %.1f, \"decode_frames_per_s\": %.1f, \"encode_decode_pipelined_frames_per_s\": %.1f, "
              "\"encode_ms_per_hop\": %.4f, \"decode_ms_per_hop\": %.4f, \"pipelined_ms_per_hop\": %.4f, "
              "\"two_deep\": {\"what\": \"the same regions through EncodeAsync / WaitEncoded and DecodeSamplesAsync / WaitDecoded, "
              "hop n + 1 started before hop n is collected\", \"encode_frames_per_s\": %.1f, \"decode_frames_per_s\": %.1f, "
              "\"encode_decode_two_threads_frames_per_s\": %.1f}}\n",
              n, rate, bitrate, dtx ? "true" : "false", loss, hops, lost_dec, frames / enc_s, frames / dec_s, frames / both_s,
              enc_s / hops * 1e3, dec_s / hops * 1e3, both_s / hops * 1e3, frames / enc_p_s, frames / dec_p_s, frames / both_p_s);
  return 0;
}

// decoder_demo.cc -- drives BatchLyraEncoder / BatchLyraDecoder through a scripted session with packet loss, DTX,
// arbitrary DecodeSamples sizes and a non-16 kHz sample rate (what cli_example/decoder_main_lib.cc:86-140 does per file
// with a packet-loss model, here for many streams at once):
//   decoder_demo <model_dir> <script.txt> <pcm_in.s16> <sample_rate> <bitrate> <dtx 0|1> <num_streams>
//                <packets_out.bin> <lengths_out.i32> <pcm_out.s16>
// pcm_in: [ticks][num_streams][sample_rate / 50] int16.  script.txt, one line per tick:
//   <mask> <n1> <n2> ...    mask = one character per stream ('1' packet delivered, '0' lost); n_i = DecodeSamples sizes
// whose sum is the tick's playout.  Writes every tick's packets [num_streams][packet_size], lengths [num_streams]
// and all decoded samples, stream-major per DecodeSamples call.
// LYRA_DEMO_PIPELINED=1: the same session through the two-deep pipelined halves of the calls -- EncodeAsync(t + 1) is issued
// before WaitEncoded(t), DecodeSamplesAsync(request k + 1) before WaitDecoded(k) -- which must write the same three files.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "lyra_batch_codec.h"

using namespace chromemedia::codec;

int main(int argc, char** argv) {
  if (argc != 11) { std::fprintf(stderr, "usage: see decoder_demo.cc\n"); return 2; }
  const std::string model_dir = argv[1];
  const int rate = std::atoi(argv[4]), bitrate = std::atoi(argv[5]), dtx = std::atoi(argv[6]), n = std::atoi(argv[7]);
  std::ifstream script(argv[2]);
  std::ifstream in(argv[3], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::vector<int16_t> pcm(raw.size() / 2);
  std::memcpy(pcm.data(), raw.data(), pcm.size() * 2);
  auto enc = BatchLyraEncoder::Create(rate, 1, bitrate, dtx != 0, model_dir, n);
  auto dec = BatchLyraDecoder::Create(rate, 1, model_dir, n);
  if (!enc || !dec) { std::fprintf(stderr, "creation failed\n"); return 1; }
  std::ofstream pk_out(argv[8], std::ios::binary), len_out(argv[9], std::ios::binary), pcm_out(argv[10], std::ios::binary);
  const size_t frame = static_cast<size_t>(n) * (rate / 50);
  std::string line;
  size_t off = 0;
  if (const char* e = std::getenv("LYRA_DEMO_PIPELINED"); e && std::atoi(e) != 0) {
    std::vector<std::string> lines;
    while (std::getline(script, line))
      if (!line.empty()) lines.push_back(line);
    if (lines.size() * frame > pcm.size()) return 3;
    std::vector<int> waiting;   // sizes of the decode requests begun and not yet delivered (at most two)
    auto deliver_oldest = [&]() {
      std::vector<int16_t> out(static_cast<size_t>(n) * waiting.front());
      if (!dec->WaitDecoded(absl::Span<int16_t>(out.data(), out.size()))) return false;
      pcm_out.write(reinterpret_cast<const char*>(out.data()), out.size() * 2);
      waiting.erase(waiting.begin());
      return true;
    };
    if (!lines.empty() && !enc->EncodeAsync(absl::MakeConstSpan(pcm.data(), frame))) return 4;
    for (size_t t = 0; t < lines.size(); ++t) {
      if (t + 1 < lines.size() && !enc->EncodeAsync(absl::MakeConstSpan(pcm.data() + (t + 1) * frame, frame))) return 4;
      if (enc->hops_in_flight() != (t + 1 < lines.size() ? 2 : 1)) return 7;
      auto packets = enc->WaitEncoded();
      if (!packets) return 4;
      std::istringstream ls(lines[t]);
      std::string mask;
      ls >> mask;
      if (static_cast<int>(mask.size()) != n) return 3;
      pk_out.write(reinterpret_cast<const char*>(packets->data()), packets->size());
      len_out.write(reinterpret_cast<const char*>(enc->packet_lengths().data()), n * 4);
      std::vector<int32_t> ids;
      std::vector<uint8_t> delivered;
      const int ps = enc->packet_size();
      for (int s = 0; s < n; ++s)
        if (mask[s] == '1' && enc->packet_lengths()[s] > 0) {
          ids.push_back(s);
          delivered.insert(delivered.end(), packets->begin() + s * ps, packets->begin() + (s + 1) * ps);
        }
      if (!ids.empty() && !dec->SetEncodedPackets(absl::MakeConstSpan(ids), absl::MakeConstSpan(delivered))) return 5;
      int k;
      while (ls >> k) {
        if (waiting.size() == 2 && !deliver_oldest()) return 6;
        if (!dec->DecodeSamplesAsync(k)) return 6;
        waiting.push_back(k);
        if (dec->requests_in_flight() != static_cast<int>(waiting.size())) return 7;
      }
    }
    while (!waiting.empty())
      if (!deliver_oldest()) return 6;
    return 0;
  }
  while (std::getline(script, line)) {
    if (line.empty()) continue;
    std::istringstream ls(line);
    std::string mask;
    ls >> mask;
    if (static_cast<int>(mask.size()) != n || off + frame > pcm.size()) return 3;
    auto packets = enc->Encode(absl::MakeConstSpan(pcm.data() + off, frame));
    off += frame;
    if (!packets) return 4;
    pk_out.write(reinterpret_cast<const char*>(packets->data()), packets->size());
    len_out.write(reinterpret_cast<const char*>(enc->packet_lengths().data()), n * 4);
    // deliver: streams whose packet arrived AND is not a DTX empty packet (an empty packet carries no features:
    // the application does not hand it to the decoder, which then conceals / plays comfort noise)
    std::vector<int32_t> ids;
    std::vector<uint8_t> delivered;
    const int ps = enc->packet_size();
    for (int s = 0; s < n; ++s)
      if (mask[s] == '1' && enc->packet_lengths()[s] > 0) {
        ids.push_back(s);
        delivered.insert(delivered.end(), packets->begin() + s * ps, packets->begin() + (s + 1) * ps);
      }
    if (!ids.empty() && !dec->SetEncodedPackets(absl::MakeConstSpan(ids), absl::MakeConstSpan(delivered))) return 5;
    int k;
    while (ls >> k) {
      auto out = dec->DecodeSamples(k);
      if (!out || out->size() != static_cast<size_t>(n) * k) return 6;
      pcm_out.write(reinterpret_cast<const char*>(out->data()), out->size() * 2);
    }
  }
  return 0;
}

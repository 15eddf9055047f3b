// batch_demo.cc -- drives BatchLyraEncoder / BatchLyraDecoder the way an application drives LyraEncoder /
// LyraDecoder (cli_example/encoder_main_lib.cc:62-93, decoder_main_lib.cc:94-140), for many streams at once:
//   batch_demo <model_dir> <pcm_in.s16> <num_streams> <bitrate> <packets_out.bin> <pcm_out.s16>
// pcm_in holds [frames][num_streams][320] int16; writes [frames][num_streams][packet_size] and
// [frames][num_streams][320].
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "lyra_batch_codec.h"

using namespace chromemedia::codec;

int main(int argc, char** argv) {
  if (argc != 7) { std::fprintf(stderr, "usage: %s model_dir pcm_in num_streams bitrate packets_out pcm_out\n", argv[0]); return 2; }
  const std::string model_dir = argv[1];
  const int n = std::atoi(argv[3]), bitrate = std::atoi(argv[4]);
  std::ifstream in(argv[2], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::vector<int16_t> pcm(raw.size() / 2);
  std::memcpy(pcm.data(), raw.data(), pcm.size() * 2);
  // the reference's parameter validation (lyra_encoder.cc:46-66, lyra_decoder.cc:60-90)
  if (BatchLyraEncoder::Create(16000, 2, bitrate, false, model_dir, n)) return 3;
  if (BatchLyraEncoder::Create(44100, 1, bitrate, false, model_dir, n)) return 3;
  if (BatchLyraEncoder::Create(16000, 1, 1234, false, model_dir, n)) return 3;
  if (BatchLyraDecoder::Create(16000, 3, model_dir, n)) return 3;
  auto enc = BatchLyraEncoder::Create(16000, 1, bitrate, false, model_dir, n);
  auto dec = BatchLyraDecoder::Create(16000, 1, model_dir, n);
  if (!enc || !dec) { std::fprintf(stderr, "creation failed\n"); return 1; }
  if (enc->Encode(absl::MakeConstSpan(pcm.data(), 100)).has_value()) return 3;   // wrong sample count
  if (dec->SetEncodedPackets(absl::MakeConstSpan(reinterpret_cast<const uint8_t*>(pcm.data()), 7 * n))) return 3;
  std::ofstream pk_out(argv[5], std::ios::binary), pcm_out(argv[6], std::ios::binary);
  const size_t frame = static_cast<size_t>(n) * 320;
  for (size_t off = 0; off + frame <= pcm.size(); off += frame) {
    auto packets = enc->Encode(absl::MakeConstSpan(pcm.data() + off, frame));
    if (!packets || packets->size() != static_cast<size_t>(n) * enc->packet_size()) return 4;
    pk_out.write(reinterpret_cast<const char*>(packets->data()), packets->size());
    if (!dec->SetEncodedPackets(*packets)) return 4;
    auto a = dec->DecodeSamples(120);                 // partial requests inside a hop
    auto b = dec->DecodeSamples(200);
    if (!a || !b || dec->is_comfort_noise(0)) return 5;
    std::vector<int16_t> hop(frame);
    for (int s = 0; s < n; ++s) {
      std::memcpy(&hop[static_cast<size_t>(s) * 320], &(*a)[static_cast<size_t>(s) * 120], 240);
      std::memcpy(&hop[static_cast<size_t>(s) * 320 + 120], &(*b)[static_cast<size_t>(s) * 200], 400);
    }
    pcm_out.write(reinterpret_cast<const char*>(hop.data()), hop.size() * 2);
  }
  return 0;
}

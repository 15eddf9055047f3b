// lyra_file_codec.h -- whole-file transcodes, many files at once: batched twins of the CLI
// library functions EncodeWav / EncodeFile (cli_example/encoder_main_lib.h:29-38, .cc:42-133) and DecodeFeatures /
// DecodeFile (cli_example/decoder_main_lib.h:51-69).  File i is stream i of one GPU context; files of different
// lengths simply leave the batch when they run out of full 20 ms hops (the C ABI takes any subset of stream ids
// per call).  The `.lyra` format is the reference's: the packets of a stream concatenated, nothing else
// (encoder_main_lib.cc:120-130; a trailing partial hop is dropped, :71-73).
//
// Same scope as lyra_batch_codec.h: 16 kHz mono 16-bit WAV only, no preprocessing / DTX / packet-loss simulation.
#ifndef LYRA_AMD_HOST_LYRA_FILE_CODEC_H_
#define LYRA_AMD_HOST_LYRA_FILE_CODEC_H_
#include <cstdint>
#include <vector>

#include "include/ghc/filesystem.hpp"

namespace chromemedia {
namespace codec {

// EncodeWav for a batch: wav_data[i] -> encoded_features[i] (packets of stream i, oldest first).
bool EncodeWavs(const std::vector<std::vector<int16_t>>& wav_data, int num_channels, int sample_rate_hz, int bitrate,
                bool enable_preprocessing, bool enable_dtx, const ghc::filesystem::path& model_path,
                std::vector<std::vector<uint8_t>>* encoded_features, int device = 0);

// EncodeFile for a batch: wav_paths[i] -> output_paths[i].
bool EncodeFiles(const std::vector<ghc::filesystem::path>& wav_paths,
                 const std::vector<ghc::filesystem::path>& output_paths, int bitrate, bool enable_preprocessing,
                 bool enable_dtx, const ghc::filesystem::path& model_path, int device = 0);

// DecodeFeatures for a batch, no packet loss: packet_streams[i] (multiple of packet_size bytes) -> decoded_audio[i].
bool DecodeFeaturesBatch(const std::vector<std::vector<uint8_t>>& packet_streams, int packet_size,
                         const ghc::filesystem::path& model_path, std::vector<std::vector<int16_t>>* decoded_audio,
                         int device = 0);

// DecodeFile for a batch: encoded_paths[i] (.lyra) -> output_paths[i] (16-bit mono WAV at sample_rate_hz).
bool DecodeFiles(const std::vector<ghc::filesystem::path>& encoded_paths,
                 const std::vector<ghc::filesystem::path>& output_paths, int sample_rate_hz, int bitrate,
                 const ghc::filesystem::path& model_path, int device = 0);

// Minimal RIFF/WAVE PCM16 I/O (the reference uses audio_dsp's wav_util: Read16BitWavFileToVector /
// Write16BitWavFileFromVector).  False on anything but uncompressed 16-bit PCM.
bool ReadWav16(const ghc::filesystem::path& path, std::vector<int16_t>* samples, int* num_channels,
               int* sample_rate_hz);
bool WriteWav16(const ghc::filesystem::path& path, const std::vector<int16_t>& samples, int num_channels,
                int sample_rate_hz);

}  // namespace codec
}  // namespace chromemedia
#endif

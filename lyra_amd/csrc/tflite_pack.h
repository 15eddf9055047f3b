// tflite_pack.h -- build the model builder's tensor container straight from a reference model directory
// (soundstream_encoder.tflite, quantizer.tflite, lyragan.tflite, lyra_config.binarypb); see tflite_pack.cc.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "pack_format.h"

namespace lyra {

// On success `container` holds a LYRAPK01 image byte-identical to what tools/pack_weights.py writes for `dir`.
bool pack_from_tflite_dir(const std::string& dir, std::vector<uint8_t>* container, std::string* err);

}  // namespace lyra

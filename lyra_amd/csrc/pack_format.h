// pack_format.h -- the LYRAPK01 tensor container (tools/pack_weights.py, tflite_pack.cc -> model.hip).
//   char magic[8] = "LYRAPK01"; uint32 n_entries; uint32 reserved; PackEntry[n]; payloads (64-byte aligned)
#pragma once
#include <cstdint>

namespace lyra {

struct PackEntry {
  char name[56];
  uint32_t dtype, ndim, shape[4];   // dtype: 0 f32, 1 i8, 2 i32
  uint64_t offset, nbytes;          // offset from the start of the image
};
static_assert(sizeof(PackEntry) == 96, "container entry layout");

}  // namespace lyra

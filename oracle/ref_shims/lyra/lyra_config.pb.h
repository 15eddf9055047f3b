// stand-in for the protoc output of lyra/lyra_config.proto (message LyraConfig { optional int32 identifier = 1; })
#ifndef REF_SHIM_LYRA_CONFIG_PB_H_
#define REF_SHIM_LYRA_CONFIG_PB_H_
#include <cstdint>
#include <istream>
#include <iterator>
#include <string>
namespace third_party { namespace lyra_codec { namespace lyra {
class LyraConfig {
 public:
  int identifier() const { return identifier_; }
  bool ParseFromIstream(std::istream* in) {   // wire format: repeated (tag varint, value); field 1 is a varint
    std::string b((std::istreambuf_iterator<char>(*in)), std::istreambuf_iterator<char>());
    size_t i = 0;
    auto varint = [&](uint64_t* v) {
      *v = 0;
      for (int shift = 0; i < b.size() && shift < 64; shift += 7) {
        const uint8_t c = (uint8_t)b[i++];
        *v |= (uint64_t)(c & 127) << shift;
        if (!(c & 128)) return true;
      }
      return false;
    };
    while (i < b.size()) {
      uint64_t tag, v;
      if (!varint(&tag)) return false;
      if ((tag & 7) != 0 || !varint(&v)) return false;   // only varint fields exist in this message
      if ((tag >> 3) == 1) identifier_ = (int)v;
    }
    return true;
  }
 private:
  int identifier_ = 0;
};
}}}  // namespace third_party::lyra_codec::lyra
#endif

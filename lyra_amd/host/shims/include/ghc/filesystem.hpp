// ghc::filesystem is a std::filesystem backport; C++17 has the real thing.
#ifndef LYRA_AMD_SHIM_GHC_FILESYSTEM_H_
#define LYRA_AMD_SHIM_GHC_FILESYSTEM_H_
#include <filesystem>
namespace ghc { namespace filesystem = std::filesystem; }
#endif

#ifndef REF_SHIM_ABSL_BIT_GEN_REF_H_
#define REF_SHIM_ABSL_BIT_GEN_REF_H_
#include "absl/random/random.h"
#endif

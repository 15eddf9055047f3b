// stand-in: the attribute macros the reference's headers use (see ../README.md)
#ifndef REF_SHIM_ABSL_ATTRIBUTES_H_
#define REF_SHIM_ABSL_ATTRIBUTES_H_
#define ABSL_CONST_INIT
#define ABSL_MUST_USE_RESULT
#endif

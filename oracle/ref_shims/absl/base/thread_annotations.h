// stand-in: the annotations are documentation for clang's thread-safety analysis; empty here
#ifndef REF_SHIM_ABSL_THREAD_ANNOTATIONS_H_
#define REF_SHIM_ABSL_THREAD_ANNOTATIONS_H_
#define ABSL_GUARDED_BY(x)
#define ABSL_LOCKS_EXCLUDED(...)
#define ABSL_EXCLUSIVE_LOCKS_REQUIRED(...)
#endif

// anyorder_probe.hip -- does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let a kernel start beside its predecessor in
// the SAME stream on gfx950 (AQL packet without the barrier bit)?  Two 40 us spin kernels back to back, with and without
// the flag, and with an event record / a cross-stream wait between them; prints when the second one started.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Rec { unsigned long long t0, t1; };
__global__ void spin(Rec* out, int ticks) {
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  while ((unsigned long long)wall_clock64() - t0 < (unsigned long long)ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = Rec{t0, (unsigned long long)wall_clock64()};
}
static void launch(hipStream_t s, Rec* out, int ticks, int flags) {
  void* args[] = {&out, &ticks};
  CHECK(hipExtLaunchKernel(reinterpret_cast<const void*>(spin), dim3(64), dim3(256), args, 0, s, nullptr, nullptr, flags));
}
int main() {
  CHECK(hipSetDevice(0));
  hipStream_t s, s2;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev, ev2;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
  CHECK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming | hipEventDisableSystemFence));
  Rec* d;
  CHECK(hipMalloc(&d, 4 * sizeof(Rec)));
  Rec h[4];
  auto report = [&](const char* what, int n) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("%-72s", what);
    for (int i = 0; i < n; ++i) printf("  k%d %+7.1f..%+7.1f us", i, (double)((long long)(h[i].t0 - h[0].t0)) / 100.0, (double)((long long)(h[i].t1 - h[0].t0)) / 100.0);
    printf("\n");
  };
  for (int rep = 0; rep < 2; ++rep) {
    launch(s, d, 4000, 0); launch(s, d + 1, 4000, 0);
    report("A, B in order", 2);
    launch(s, d, 4000, 0); launch(s, d + 1, 4000, hipExtAnyOrderLaunch);
    report("A, B any-order", 2);
    launch(s, d, 4000, 0); CHECK(hipEventRecord(ev, s)); launch(s, d + 1, 4000, hipExtAnyOrderLaunch);
    report("A, record, B any-order", 2);
    launch(s, d, 4000, 0); launch(s, d + 1, 4000, hipExtAnyOrderLaunch); launch(s, d + 2, 4000, 0);
    report("A, B any-order, C in order (C must follow both)", 3);
    // cross-stream wait in front of an any-order kernel: does it still hold the kernel back?
    launch(s2, d + 2, 8000, 0); CHECK(hipEventRecord(ev2, s2));
    launch(s, d, 4000, 0); CHECK(hipStreamWaitEvent(s, ev2, 0)); launch(s, d + 1, 4000, hipExtAnyOrderLaunch);
    report("s2: X(80us), record | s: A, wait(X), B any-order   [k2 = X]", 3);
  }
  return 0;
}

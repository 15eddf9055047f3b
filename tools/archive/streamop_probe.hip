// streamop_probe.hip -- what does a dependency between two streams cost: hipEventRecord / hipStreamWaitEvent (barrier packets
// + signals) vs hipStreamWriteValue32 / hipStreamWaitValue32 (stream memory operations on a device word)?  Two 40 us spin
// kernels A (stream s) and B (stream s2, must start after A); prints when B started relative to A's end, and the gap a
// SATISFIED wait costs between two kernels of one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Rec { unsigned long long t0, t1; };
__global__ void spin(Rec* out, int ticks) {
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  while ((unsigned long long)wall_clock64() - t0 < (unsigned long long)ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = Rec{t0, (unsigned long long)wall_clock64()};
}
int main() {
  CHECK(hipSetDevice(0));
  hipStream_t s, s2;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev, ev_old;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
  CHECK(hipEventCreateWithFlags(&ev_old, hipEventDisableTiming | hipEventDisableSystemFence));
  Rec* d; CHECK(hipMalloc(&d, 4 * sizeof(Rec)));
  unsigned* flag; CHECK(hipMalloc(&flag, 64)); CHECK(hipMemset(flag, 0, 64));
  Rec h[4];
  unsigned seq = 0;
  auto report = [&](const char* what) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("%-86s A %+6.1f..%+6.1f  B %+6.1f..%+6.1f us  (B starts %+5.1f us after A ends)\n", what, 0.0, (double)(h[0].t1 - h[0].t0) / 100.0,
           (double)((long long)(h[1].t0 - h[0].t0)) / 100.0, (double)((long long)(h[1].t1 - h[0].t0)) / 100.0, (double)((long long)(h[1].t0 - h[0].t1)) / 100.0);
  };
  CHECK(hipEventRecord(ev_old, s2));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 1, 4000);
    report("one stream: A, B");
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); CHECK(hipEventRecord(ev, s)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 1, 4000);
    report("one stream: A, eventRecord, B");
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); CHECK(hipStreamWaitEvent(s, ev_old, 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 1, 4000);
    report("one stream: A, waitEvent(long satisfied), B");
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); CHECK(hipEventRecord(ev, s)); CHECK(hipStreamWaitEvent(s, ev_old, 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 1, 4000);
    report("one stream: A, eventRecord, waitEvent(long satisfied), B");
    ++seq;
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); CHECK(hipStreamWriteValue32(s, flag, seq, 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 1, 4000);
    report("one stream: A, writeValue32, B");
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); CHECK(hipStreamWaitValue32(s, flag, seq, hipStreamWaitValueGte, 0xffffffffu)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 1, 4000);
    report("one stream: A, waitValue32(long satisfied), B");
    // cross-stream
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); CHECK(hipEventRecord(ev, s)); CHECK(hipStreamWaitEvent(s2, ev, 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s2, d + 1, 4000);
    report("two streams: s: A, eventRecord | s2: waitEvent, B");
    ++seq;
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 4000); CHECK(hipStreamWriteValue32(s, flag, seq, 0)); CHECK(hipStreamWaitValue32(s2, flag, seq, hipStreamWaitValueGte, 0xffffffffu)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s2, d + 1, 4000);
    report("two streams: s: A, writeValue32 | s2: waitValue32, B");
  }
  return 0;
}

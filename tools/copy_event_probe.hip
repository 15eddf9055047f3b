// copy_event_probe.hip -- does a kernel on stream B, held back by hipStreamWaitEvent on an event recorded on stream A right
// behind a host-to-device copy, always see the copied data?  (The first form of lyra_hip_decode_begin relied on that edge and
// decoded stale packets; profiles/EXPERIMENTS.md.)  Both streams are kept busy with back-to-back kernels, as in the library.
//   hipcc --offload-arch=gfx950 -O2 tools/copy_event_probe.hip -o tools/copy_event_probe.bin && tools/copy_event_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
__global__ void spin(float* p, int iters) {   // busy work: keeps the queue full
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
__global__ void consume(const int* __restrict__ x, int* __restrict__ y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] + 1;
}
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 2000, n = 4096;
  const bool disable_fence = argc > 2 && std::atoi(argv[2]);
  const int variant = argc > 3 ? std::atoi(argv[3]) : 0;   // 1: the copy follows a wait for B's previous round (as the library's did)
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  int *hx, *hy, *dx[2], *dy[2]; float* scratch;
  CK(hipHostMalloc((void**)&hx, 2 * n * 4)); CK(hipHostMalloc((void**)&hy, 2 * n * 4));
  for (int s = 0; s < 2; ++s) { CK(hipMalloc(&dx[s], n * 4)); CK(hipMalloc(&dy[s], n * 4)); CK(hipMemset(dx[s], 0, n * 4)); }
  CK(hipMalloc(&scratch, 512 * 256 * 4)); CK(hipMemset(scratch, 0, 512 * 256 * 4));
  hipEvent_t up[2], done[2], bprev;
  CK(hipEventCreateWithFlags(&bprev, hipEventDisableTiming | hipEventDisableSystemFence));
  for (int s = 0; s < 2; ++s) {
    CK(hipEventCreateWithFlags(&up[s], hipEventDisableTiming | (disable_fence ? hipEventDisableSystemFence : 0)));
    CK(hipEventCreateWithFlags(&done[s], hipEventDisableTiming));
  }
  long bad_rounds = 0, bad_values = 0;
  for (int r = 0; r < rounds; ++r) {
    const int s = r & 1;
    if (r >= 2) {   // slot s's previous round: check, then reuse
      CK(hipEventSynchronize(done[s]));
      long bad = 0;
      for (int i = 0; i < n; ++i) bad += hy[s * n + i] != (r - 2) * 7 + i + 1;
      bad_rounds += bad != 0; bad_values += bad;
    }
    for (int i = 0; i < n; ++i) hx[s * n + i] = r * 7 + i;
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, scratch, 2000);            // A is busy in front of the copy
    if (variant == 1 && r > 0) CK(hipStreamWaitEvent(a, bprev, 0));
    CK(hipMemcpyAsync(dx[s], hx + s * n, n * 4, hipMemcpyHostToDevice, a));
    CK(hipEventRecord(up[s], a));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, b, scratch + 256 * 256, 1000); // B is busy too
    CK(hipStreamWaitEvent(b, up[s], 0));
    hipLaunchKernelGGL(consume, dim3(n / 256), dim3(256), 0, b, dx[s], dy[s], n);
    CK(hipMemcpyAsync(hy + s * n, dy[s], n * 4, hipMemcpyDeviceToHost, b));
    CK(hipEventRecord(done[s], b));
    if (variant == 1) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, b, scratch + 256 * 256, 3000); CK(hipEventRecord(bprev, b)); }
  }
  CK(hipDeviceSynchronize());
  std::printf("variant %d  rounds %d  (event %s system fence)  rounds with stale data %ld  stale values %ld\n", variant, rounds,
              disable_fence ? "without" : "with", bad_rounds, bad_values);
  return bad_rounds != 0;
}

// mfma_probe.hip -- checks the two hardware assumptions the kernels rest on (run on the GPU box):
//  1. v_mfma_f32_16x16x4_f32: A[m=lane&15][k=lane>>4], B[k=lane>>4][n=lane&15], C[row=(lane>>4)*4+reg][col=lane&15],
//     and the result is bitwise the k-ascending fmaf chain.
//  2. v_mfma_i32_16x16x64_i8: lane (r, q) supplies 16 consecutive bytes of row r / column r for k = q*16..q*16+15
//     (any consistent k assignment works for integers; this checks ours).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void kf(const float* A, const float* B, float* C, int K) {  // A[16][K], B[K][16], C[16][16]
  int l = threadIdx.x, m = l & 15, q = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * K + k0 + q], B[(k0 + q) * 16 + m], acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) C[(q * 4 + e) * 16 + m] = acc[e];
}
__global__ void ki(const int8_t* A, const int8_t* B, int* C) {  // A[16][64], Bt[16][64] (column-major B), C[16][16]
  int l = threadIdx.x, m = l & 15, q = l >> 4;
  i32x4 a = *reinterpret_cast<const i32x4*>(A + m * 64 + q * 16);
  i32x4 b = *reinterpret_cast<const i32x4*>(B + m * 64 + q * 16);
  i32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) C[(q * 4 + e) * 16 + m] = acc[e];
}
int main() {
  const int K = 64;
  float hA[16 * K], hB[K * 16], hC[256], ref[256];
  srand(1);
  for (auto& v : hA) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2 - 1) * 37.f;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(hA[m * K + k], hB[k * 16 + n], acc);
      ref[m * 16 + n] = acc;
    }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  kf<<<1, 64>>>(dA, dB, dC, K);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += memcmp(&hC[i], &ref[i], 4) != 0;
  printf("f32 mfma 16x16x4 vs k-ascending fmaf chain: %d / 256 bitwise mismatches\n", bad);
  int8_t iA[16 * 64], iB[16 * 64]; int iC[256];
  for (auto& v : iA) v = (int8_t)(rand() % 256 - 128);
  for (auto& v : iB) v = (int8_t)(rand() % 256 - 128);
  int8_t *qA, *qB; int* qC;
  hipMalloc(&qA, sizeof iA); hipMalloc(&qB, sizeof iB); hipMalloc(&qC, sizeof iC);
  hipMemcpy(qA, iA, sizeof iA, hipMemcpyHostToDevice); hipMemcpy(qB, iB, sizeof iB, hipMemcpyHostToDevice);
  ki<<<1, 64>>>(qA, qB, qC);
  hipMemcpy(iC, qC, sizeof iC, hipMemcpyDeviceToHost);
  int ibad = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      int acc = 0;
      for (int k = 0; k < 64; ++k) acc += (int)iA[m * 64 + k] * (int)iB[n * 64 + k];
      ibad += acc != iC[m * 16 + n];
    }
  printf("i8 mfma 16x16x64: %d / 256 mismatches\n", ibad);
  return (bad || ibad) ? 1 : 0;
}

// hazard_probe.hip -- does gfx950 interlock a vector read against an MFMA that is still writing its destination?  No:
// a v_pk_mul_f32 / v_max_f32 right behind a v_mfma_f32_16x16x4_f32 reads elements 2, 3 of the destination stale (elements
// 0, 1 have landed by then).  The compiler pads its own instructions; an asm block is on its own (lyra_dev.h lrelu4).
//   hipcc --offload-arch=gfx950 -O2 tools/hazard_probe.hip -o tools/hazard_probe.bin && tools/hazard_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
// acc = v[20:23] (1,2,3,4 initially), d2 = v[24:27], A = v28, B = v29 (1.0 each: every MFMA adds 4 to each element on
// lanes whose row/col..., with all-ones A/B: C += sum_k a*b = 4), results r0 = v30, r1 = v31, v[32:33] pk, v[34:35] = 2.0
#define PRE "v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 2.0\n\tv_mov_b32 v22, 0x40400000\n\tv_mov_b32 v23, 4.0\n\t" \
            "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 1.0\n\tv_mov_b32 v29, 1.0\n\t" \
            "v_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v34, 2.0\n\tv_mov_b32 v35, 2.0\n\ts_nop 7\n\t"
#define CL "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"
#define M_IN "v_mfma_f32_16x16x4_f32 v[20:23], v28, v29, v[20:23]\n\t"
#define M_OUT "v_mfma_f32_16x16x4_f32 v[24:27], v28, v29, v[20:23]\n\t"
template <int MODE>
__global__ void k(float* out) {
  float r0, r1, r2, r3;
  if (MODE == 0)       asm volatile(PRE M_IN M_IN M_IN "v_mov_b32 v30, v20\n\tv_mov_b32 v31, v23\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v30\n\tv_mov_b32 %1, v31\n\tv_mov_b32 %2, v20\n\tv_mov_b32 %3, v23" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: CL);
  else if (MODE == 1)  asm volatile(PRE M_IN M_IN M_OUT "v_mov_b32 v30, v24\n\tv_mov_b32 v31, v27\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v30\n\tv_mov_b32 %1, v31\n\tv_mov_b32 %2, v24\n\tv_mov_b32 %3, v27" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: CL);
  else if (MODE == 2)  asm volatile(PRE M_IN M_IN M_OUT "v_pk_mul_f32 v[32:33], v[26:27], v[34:35]\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: CL);
  else if (MODE == 3)  asm volatile(PRE M_IN M_IN M_IN "v_pk_mul_f32 v[32:33], v[22:23], v[34:35]\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v22\n\tv_mov_b32 %3, v23" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: CL);
  else if (MODE == 4)  asm volatile(PRE M_IN M_IN M_OUT "v_pk_mul_f32 v[32:33], v[24:25], v[34:35]\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v24\n\tv_mov_b32 %3, v25" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: CL);
  else                 asm volatile(PRE M_IN M_IN M_OUT "v_max_f32 v30, v26, v26\n\tv_max_f32 v31, v27, v27\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v30\n\tv_mov_b32 %1, v31\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: CL);
  if (threadIdx.x == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; }
}
int main() {
  float* d; (void)hipMalloc(&d, 64);
  float h[4];
  const char* what[] = {"in-place chain, v_mov of elements 0 / 3 right behind the last MFMA", "last MFMA into ANOTHER destination, v_mov right behind",
                        "other destination, v_pk_mul_f32 (x2) of elements 2,3 right behind", "in-place, v_pk_mul_f32 (x2) of elements 2,3 right behind",
                        "other destination, v_pk_mul_f32 (x2) of elements 0,1 right behind", "other destination, v_max_f32 of elements 2 / 3 right behind"};
  #define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, d); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
    printf("mode %d (%s): early read %g %g   settled %g %g\n", M, what[M], h[0], h[1], h[2], h[3]);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  return 0;
}

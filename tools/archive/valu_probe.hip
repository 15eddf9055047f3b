// valu_probe.hip -- issue cost of the VALU / LDS instructions the int8 epilogues and the quantizer are made of, on gfx950.
// Each kernel runs REP x 64 copies of one instruction pattern in one wavefront per SIMD (256 threads / CU) and with 2, 4, 8
// wavefronts per SIMD, and reports shader-clock cycles (s_memtime) per instruction per wave and per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_probe.hip -o tools/valu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r[0] = a; r[1] = b; return r; }
__device__ long long g_cycles[64];

#define PROBE(NAME, SETUP, BODY64, SINK)                                                        \
  __global__ __launch_bounds__(256) void NAME(float* out, int rep) {                            \
    __shared__ float lds[4096];                                                                 \
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;                            \
    __syncthreads();                                                                            \
    float a0 = threadIdx.x * 1e-3f, a1 = 1.0001f, a2 = 0.5f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;  \
    float b0 = 1.f, b1 = 2.f, b2 = 3.f, b3 = 4.f;                                                \
    int i0 = threadIdx.x * 77 + 3, i1 = 0x12345, i2 = 77, i3 = 5;                               \
    const float* lp = lds + (threadIdx.x & 255);                                                \
    SETUP;                                                                                      \
    long long t0 = clock64();                                                                   \
    for (int r = 0; r < rep; ++r) { BODY64 }                                                    \
    long long t1 = clock64();                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_cycles[0] = t1 - t0;                             \
    SINK;                                                                                       \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + (float)(i0 + i1 + i2 + i3) == 1.2345f) out[threadIdx.x] = a0; \
    (void)lp;                                                                                   \
  }

// 1 independent v_add_f32 (8 accumulators round robin)
PROBE(k_add_indep, , R16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(b0));
                         asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(b0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(b0));), )
// 2 dependent v_add_f32
PROBE(k_add_dep, , R64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b0));), )
// 3 the quantizer's term: sub_dpp, mul, dependent add
PROBE(k_rvq_term, , R16(asm volatile("v_sub_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mul_f32 %0, %0, %0\n\tv_add_f32 %3, %3, %0" : "=&v"(a1), "+v"(a2), "+v"(b1), "+v"(a0));
                        ) , )
// 4 same without DPP
PROBE(k_rvq_term_nodpp, , R16(asm volatile("v_sub_f32 %0, %1, %2\n\tv_mul_f32 %0, %0, %0\n\tv_add_f32 %3, %3, %0" : "=&v"(a1), "+v"(a2), "+v"(b1), "+v"(a0));
                              ) , )
// 5 independent v_pk_add_f32
PROBE(k_pk_add_indep, f2 p0 = mk2(a0, a1); f2 p1 = mk2(a2, a3); f2 p2 = mk2(a4, a5); f2 p3 = mk2(a6, a7); f2 q = mk2(b0, b1);,
      R16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(q)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(q));
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(q)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(q));),
      a0 = p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];)
// 6 dependent v_pk_add_f32
PROBE(k_pk_add_dep, f2 p0 = mk2(a0, a1); f2 q = mk2(b0, b1);,
      R64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(q));), a0 = p0[0] + p0[1];)
// 7 packed quantizer term: pk_add(neg) , pk_mul, dependent pk_add
PROBE(k_pk_term, f2 p0 = mk2(a0, a1); f2 d = mk2(a2, a3); f2 rr = mk2(a4, a5); f2 q = mk2(b0, b1);,
      R16(asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %0\n\tv_pk_add_f32 %3, %3, %0" : "=&v"(d), "+v"(rr), "+v"(q), "+v"(p0));
          ),
      a0 = p0[0] + p0[1] + d[0];)
// 8 v_mul_lo_u32 / v_mul_hi_i32 (requantisation)
PROBE(k_mul_lo, , R16(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(i0) : "v"(i1), "v"(i2)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(i3) : "v"(i1), "v"(i2));
                      asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(i0) : "v"(i1), "v"(i2)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(i3) : "v"(i1), "v"(i2));), )
PROBE(k_mul_hi, , R16(asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(i0) : "v"(i1), "v"(i2)); asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(i3) : "v"(i1), "v"(i2));
                      asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(i0) : "v"(i1), "v"(i2)); asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(i3) : "v"(i1), "v"(i2));), )
// 9 v_mad_i64_i32
PROBE(k_mad_i64, long long w0 = i0; long long w1 = i1;,
      R16(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w0) : "v"(i1), "v"(i2) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w1) : "v"(i1), "v"(i2) : "vcc");
          asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w0) : "v"(i1), "v"(i2) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w1) : "v"(i1), "v"(i2) : "vcc");),
      i0 = (int)(w0 ^ w1);)
// 10 cheap integer ops: v_med3_i32, v_add_u32, v_ashrrev_i32, v_perm_b32, v_mad_u32_u24 / v_mad_i32_i24
PROBE(k_med3, , R16(asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3)); asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3));
                    asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3)); asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3));), )
PROBE(k_perm, , R16(asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3));), )
PROBE(k_mad24, , R16(asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3)); asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3));
                     asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3)); asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(i0) : "v"(i1), "v"(i2), "v"(i3));), )
PROBE(k_dot4, , R16(asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(i0) : "v"(i1), "v"(i2)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(i3) : "v"(i1), "v"(i2));
                    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(i0) : "v"(i1), "v"(i2)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(i3) : "v"(i1), "v"(i2));), )
PROBE(k_cvt_f32_i32, , R16(asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a0) : "v"(i1)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a1) : "v"(i1));
                           asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a2) : "v"(i1)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a3) : "v"(i1));), )
PROBE(k_fma_f64, double d0 = a0; double d1 = a1; double d2 = a2; double d3 = a3; double e = 1.000001;,
      R16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(e)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d1) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d2) : "v"(e)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d3) : "v"(e));),
      a0 = (float)(d0 + d1 + d2 + d3);)
// 11 LDS: byte gathers (table lookups), dword reads, b128 reads; 4 in flight
PROBE(k_lds_u8, const char* bp = (const char*)lds + (threadIdx.x * 37 & 1023);,
      R16(asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:64\n\tds_read_u8 %2, %4 offset:128\n\tds_read_u8 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                       : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3) : "v"((unsigned)(size_t)bp) : "memory");), )
PROBE(k_lds_b32, , R16(asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:1024\n\tds_read_b32 %2, %4 offset:2048\n\tds_read_b32 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3) : "v"((unsigned)(size_t)lp) : "memory");), )
PROBE(k_lds_b128, i4 v0; i4 v1; i4 v2; i4 v3; const float* lq = lds + (threadIdx.x & 63) * 4;,
      R16(asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"((unsigned)(size_t)lq) : "memory");), i0 = v0[0] + v1[1] + v2[2] + v3[3];)
// b128 broadcast read: all 16 lanes of a row read the same address
PROBE(k_lds_b128_bcast, i4 v0; i4 v1; i4 v2; i4 v3; const float* lq = lds + (threadIdx.x >> 4 & 3) * 68;,
      R16(asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                       : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"((unsigned)(size_t)lq) : "memory");), i0 = v0[0] + v1[1] + v2[2] + v3[3];)
PROBE(k_lds_w8, const char* bp = (const char*)lds + (threadIdx.x * 37 & 1023);,
      R16(asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %1 offset:64\n\tds_write_b8 %0, %1 offset:128\n\tds_write_b8 %0, %1 offset:192\n\ts_waitcnt lgkmcnt(0)"
                       : : "v"((unsigned)(size_t)bp), "v"(i1) : "memory");), )
PROBE(k_lds_w32, , R16(asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:1024\n\tds_write_b32 %0, %1 offset:2048\n\tds_write_b32 %0, %1 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : : "v"((unsigned)(size_t)lp), "v"(i1) : "memory");), )

struct Entry { const char* name; void (*fn)(float*, int); int insts_per_rep; };

int main() {
  Entry es[] = {{"v_add_f32 independent", k_add_indep, 64}, {"v_add_f32 dependent chain", k_add_dep, 64},
                {"rvq term: v_sub_f32_dpp row_newbcast, v_mul, dependent v_add", k_rvq_term, 48},
                {"rvq term without DPP", k_rvq_term_nodpp, 48},
                {"v_pk_add_f32 independent", k_pk_add_indep, 64}, {"v_pk_add_f32 dependent chain", k_pk_add_dep, 64},
                {"packed term: v_pk_add(neg), v_pk_mul, dependent v_pk_add", k_pk_term, 48},
                {"v_mul_lo_u32", k_mul_lo, 64}, {"v_mul_hi_i32", k_mul_hi, 64}, {"v_mad_i64_i32", k_mad_i64, 64},
                {"v_med3_i32", k_med3, 64}, {"v_perm_b32", k_perm, 64}, {"v_mad_i32_i24", k_mad24, 64},
                {"v_dot4_i32_i8", k_dot4, 64}, {"v_cvt_f32_i32", k_cvt_f32_i32, 64}, {"v_fma_f64", k_fma_f64, 64},
                {"ds_read_u8 x4 + wait", k_lds_u8, 64}, {"ds_read_b32 x4 + wait", k_lds_b32, 64},
                {"ds_read_b128 x4 + wait", k_lds_b128, 64}, {"ds_read_b128 x4 row-broadcast + wait", k_lds_b128_bcast, 64},
                {"ds_write_b8 x4 + wait", k_lds_w8, 64}, {"ds_write_b32 x4 + wait", k_lds_w32, 64}};
  float* out;
  hipMalloc(&out, 4096);
  const int rep = 2048;
  hipEvent_t ea, eb;
  hipEventCreate(&ea); hipEventCreate(&eb);
  printf("%-66s %17s %17s %17s %17s\n  (per cell: shader-clock cycles per instruction seen by wave 0 | SIMD cycles per wave-instruction from the kernel's "
         "wall time at 2.4 GHz, all waves)\n", "pattern", "1 wave/SIMD", "2", "4", "8");
  for (const Entry& e : es) {
    printf("%-66s", e.name);
    for (int wps : {1, 2, 4, 8}) {
      hipLaunchKernelGGL(e.fn, dim3(256 * wps), dim3(256), 0, 0, out, 16);   // warm
      hipDeviceSynchronize();
      hipEventRecord(ea, 0);
      hipLaunchKernelGGL(e.fn, dim3(256 * wps), dim3(256), 0, 0, out, rep);
      hipEventRecord(eb, 0);
      hipEventSynchronize(eb);
      float ms = 0;
      hipEventElapsedTime(&ms, ea, eb);
      long long cyc = 0;
      hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cycles), 8);
      const double per = (double)cyc / ((double)rep * e.insts_per_rep);
      const double simd = (double)ms * 1e-3 * 2.4e9 / ((double)rep * e.insts_per_rep * wps);
      printf("   %6.2f | %6.2f", per, simd);
    }
    printf("\n");
  }
  // wall-clock cross-check of the cycle counter: the dependent chain, 1 wave / SIMD
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k_add_dep, dim3(256), dim3(256), 0, 0, out, 65536);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  long long cyc = 0;
  hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cycles), 8);
  printf("clock64 ticks per second (dependent chain, 65536 x 64 adds): %.3f GHz-equivalent over %.3f ms\n", cyc / (ms * 1e6), ms);
  return 0;
}

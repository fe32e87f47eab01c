// Microbenchmark (diagnostic): cycles per instruction of ONE wave64 on gfx950 for the instruction kinds the extended-range
// redo (end2end_amd/csrc/ctc_ext.h) is made of -- v_ldexp_f64, v_frexp_*, v_cvt_f64_f32, v_cmp_neq_f64, f64 add / mul / fma,
// integer max / sub -- as dependent chains and as four independent streams.
//   hipcc -O3 --offload-arch=gfx950 tools/diag/microbench/ubench_f64ops.hip -o build/diag/ubench_f64ops ; on the box: ./build/diag/ubench_f64ops
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define N_OUTER 256
template <int KIND>
__global__ __launch_bounds__(64) void bench(unsigned long long* out, double* sink) {
  double a = 1.0 + threadIdx.x * 1e-9, a2 = 1.1, a3 = 1.2, a4 = 1.3, b = 1.0000001, c = 1e-9;
  int e = threadIdx.x & 3, e2 = 1, e3 = 2, e4 = 3;
  float f = 1.0f + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_OUTER; i++) {
    if (KIND == 0) { REP16(asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(e));) }
    if (KIND == 1) { for (int k = 0; k < 4; k++) asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(e)); }
    if (KIND == 2) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 3) { for (int k = 0; k < 4; k++) asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b)); }
    if (KIND == 4) { REP16(asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a));) }
    if (KIND == 5) { for (int k = 0; k < 4; k++) asm volatile("v_frexp_exp_i32_f64 %0, %4\n v_frexp_exp_i32_f64 %1, %4\n v_frexp_exp_i32_f64 %2, %4\n v_frexp_exp_i32_f64 %3, %4" : "+v"(e), "+v"(e2), "+v"(e3), "+v"(e4) : "v"(a)); }
    if (KIND == 6) { for (int k = 0; k < 4; k++) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(f)); }
    if (KIND == 7) { for (int k = 0; k < 16; k++) asm volatile("v_cmp_neq_f64 vcc, 0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(e) : "v"(a), "v"(e2) : "vcc"); }
    if (KIND == 8) { REP16(asm volatile("v_max_i32 %0, %0, %1" : "+v"(e) : "v"(e2));) }
    if (KIND == 9) { for (int k = 0; k < 4; k++) asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b)); }
    if (KIND == 10) { for (int k = 0; k < 4; k++) asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c)); }
    if (KIND == 11) { for (int k = 0; k < 4; k++) asm volatile("v_max3_i32 %0, %0, %4, %5\n v_sub_u32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_sub_u32 %3, %3, %5" : "+v"(e), "+v"(e2), "+v"(e3), "+v"(e4) : "v"(i), "v"(k)); }
    if (KIND == 12) { for (int k = 0; k < 4; k++) asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4" : "+v"(f), "+v"(e2), "+v"(e3), "+v"(e4) : "v"(e)); }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = a + a2 + a3 + a4 + e + e2 + e3 + e4 + f;
}
template <int KIND> void run(const char* name, int nwaves_per_simd) {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 8 * 4096); hipMalloc(&sink, 8 * 64 * 4096);
  const int grid = 256 * 4 * nwaves_per_simd;               // one 64-thread workgroup per wave slot
  bench<KIND><<<grid, 64>>>(out, sink); hipDeviceSynchronize();
  bench<KIND><<<grid, 64>>>(out, sink); hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD %d: %.2f s_memtime ticks per instruction (100 MHz ticks x 21-24 = core cycles)\n", name, nwaves_per_simd, (double)h[3] / (N_OUTER * 16));
  hipFree(out); hipFree(sink);
}
int main() {
  for (int w = 1; w <= 2; w++) {
    run<0>("v_ldexp_f64 dependent", w); run<1>("v_ldexp_f64 4 streams", w); run<2>("v_add_f64 dependent", w); run<3>("v_add_f64 4 streams", w);
    run<4>("v_frexp_mant_f64 dependent", w); run<5>("v_frexp_exp_i32_f64 4 streams", w); run<6>("v_cvt_f64_f32 4 streams", w);
    run<7>("v_cmp_neq_f64 + v_cndmask dependent pair (/2)", w); run<8>("v_max_i32 dependent", w); run<9>("v_mul_f64 4 streams", w);
    run<10>("v_fma_f64 4 streams", w); run<11>("int max3/sub/max/sub 4 streams", w); run<12>("v_ldexp_f32 4 streams", w);
  }
  return 0;
}

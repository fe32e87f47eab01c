// Micro-benchmark: cycles per wave-instruction for one lone wave on a CU (s_memtime around unrolled loops).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 512

template <typename F>
__device__ unsigned long long timed(F f) {
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0);
  f();
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  return t1 - t0;
}

__global__ void bench(unsigned long long* out, double* sink, int zero) {
  __shared__ double lds[1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = i * 0.5;
  __syncthreads();
  double a[8], m = 1.0 + zero * 1e-9, c = zero * 1e-9;
  float fa[8], fm = 1.0f + zero * 1e-9f, fc = zero * 1e-9f;
  for (int k = 0; k < 8; k++) { a[k] = lane + k; fa[k] = lane + k; }
  unsigned long long r[16];
  // 0: 8 independent f64 fma per iteration
  r[0] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = __builtin_fma(a[k], m, c); } });
  // 1: dependent f64 fma chain (8 per iteration on one value)
  r[1] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[0] = __builtin_fma(a[0], m, c); } });
  // 2: 8 independent f64 mul
  r[2] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = a[k] * m; } });
  // 3: 8 independent f64 add
  r[3] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = a[k] + c; } });
  // 4: 8 independent f32 fma
  r[4] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) fa[k] = __builtin_fmaf(fa[k], fm, fc); } });
  // 5: dependent f32 fma chain
  r[5] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) fa[0] = __builtin_fmaf(fa[0], fm, fc); } });
  // 6: 8 x ldexp f64
  r[6] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = ldexp(a[k], zero); } });
  // 7: 8 x ds_read_b64 (independent addresses), consumed by one add each
  r[7] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] += lds[(lane * 3 + k * 67 + i) & 1023]; } });
  // 8: dpp wave_shr of a double (2 movs) + fma, dependent through the shift
  r[8] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a[0]), 0x138, 0xf, 0xf, false);
      int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a[0]), 0x138, 0xf, 0xf, false);
      a[0] = __builtin_fma(__hiloint2double(hi, lo), m, a[0]); } } });
  // 9: f32 exp (v_exp_f32) x8 independent
  r[9] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) fa[k] = __builtin_amdgcn_exp2f(fa[k] * fc); } });
  // 10: packed f32 fma: 8 x v_pk_fma_f32 (16 flops-lanes)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa[8]; for (int k = 0; k < 8; k++) pa[k] = f2{fa[k], fa[k] + 1};
  f2 pm = {fm, fm}, pc = {fc, fc};
  r[10] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) pa[k] = __builtin_elementwise_fma(pa[k], pm, pc); } });
  // 11: ds_bpermute round trip (shfl_xor) dependent x8
  r[11] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) fa[0] += __shfl_xor(fa[0], 16, 64); } });
  double keep = 0; for (int k = 0; k < 8; k++) keep += a[k];
  // 12: 8 x v_cvt_f64_f32 (independent)
  r[12] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(fa[k])); } } });
  // 13: 8 x v_cvt_f32_f64
  r[13] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[k]) : "v"(a[k])); } } });
  // 14: 8 x v_mov_b32_dpp (independent)
  int ia[8]; for (int k = 0; k < 8; k++) ia[k] = lane + k;
  r[14] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) ia[k] = __builtin_amdgcn_update_dpp(0, ia[k], 0x138, 0xf, 0xf, true); } });
  for (int k = 0; k < 8; k++) fa[k] += ia[k];
  double s = keep; for (int k = 0; k < 8; k++) s += a[k] + fa[k] + pa[k].x + pa[k].y;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) for (int k = 0; k < 15; k++) out[blockIdx.x * 16 + k] = r[k];
}

int main() {
  const char* names[15] = {"f64 fma x8 indep", "f64 fma dependent", "f64 mul x8 indep", "f64 add x8 indep", "f32 fma x8 indep",
                           "f32 fma dependent", "f64 ldexp x8", "ds_read_b64 x8 + add", "dpp64 + fma dependent", "v_exp_f32 x8",
                           "v_pk_fma_f32 x8", "shfl_xor(bpermute)+add dep", "v_cvt_f64_f32 x8", "v_cvt_f32_f64 x8", "v_mov_b32_dpp x8"};
  for (int waves : {1, 4}) {
    unsigned long long* out; double* sink;
    hipMalloc(&out, 16 * 8 * 1024); hipMalloc(&sink, 8 * 64 * 1024 * 8);
    bench<<<1, 64 * waves>>>(out, sink, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16);
    hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost);
    printf("== %d wave(s) in one workgroup (one CU): cycles (s_memtime ticks) per wave-instruction, wave 0\n", waves);
    for (int k = 0; k < 15; k++) printf("  %-28s %7.2f\n", names[k], (double)h[k] / (N_ITER * 8.0));
  }
  return 0;
}

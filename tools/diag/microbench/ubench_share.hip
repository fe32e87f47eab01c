// Micro-benchmark: how much does a second wave on the SAME SIMD slow a high-priority f64 chain wave?
// 512-thread workgroup, waves 0 and 4 land on SIMD 0.  Wave 0: timed loop of independent f64 FMAs + DPP shifts
// (s_setprio 3).  Wave 4: nothing / LDS polling with s_sleep / exp-heavy VALU work / f64 work.  Others exit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
__global__ __launch_bounds__(512) void bench(unsigned long long* out, double* sink, int mode, int partner) {
  __shared__ int flag[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) flag[0] = 0;
  __syncthreads();
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    double a[8]; for (int k = 0; k < 8; k++) a[k] = lane + k;
    const double m = 1.0000001, c = 1e-9;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) a[k] = __builtin_fma(a[k], m, c);
      int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a[7]), 0x138, 0xf, 0xf, true);
      int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a[7]), 0x138, 0xf, 0xf, true);
      a[0] += __hiloint2double(hi, lo) * c;
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int k = 0; k < 8; k++) s += a[k];
    sink[lane] = s;
    if (lane == 0) { out[0] = t1 - t0; }
    __atomic_store_n(&flag[0], 1, __ATOMIC_RELAXED);
  } else if (wave == partner) {
    float f = lane * 0.001f; double d = lane;
    if (mode == 1) { while (__atomic_load_n(&flag[0], __ATOMIC_RELAXED) == 0) __builtin_amdgcn_s_sleep(1); }
    else if (mode == 2) { while (__atomic_load_n(&flag[0], __ATOMIC_RELAXED) == 0) { for (int k = 0; k < 16; k++) f = __builtin_amdgcn_exp2f(f * 0.5f) ; } }
    else if (mode == 3) { while (__atomic_load_n(&flag[0], __ATOMIC_RELAXED) == 0) { for (int k = 0; k < 16; k++) d = __builtin_fma(d, 1.0000001, 1e-9); } }
    else if (mode == 4) { while (__atomic_load_n(&flag[0], __ATOMIC_RELAXED) == 0) { for (int k = 0; k < 16; k++) f = __builtin_fmaf(f, 1.0001f, 1e-9f); } }
    sink[64 + lane] = f + d;
  }
}
int main() {
  unsigned long long* out; double* sink; hipMalloc(&out, 64); hipMalloc(&sink, 4096);
  const char* names[5] = {"alone", "partner polls LDS (s_sleep 1)", "partner: v_exp_f32 loop", "partner: f64 fma loop", "partner: f32 fma loop"};
  for (int partner : {4, 1}) {
    printf("partner wave %d (%s SIMD):\n", partner, partner == 4 ? "same" : "another");
    for (int mode = 0; mode < 5; mode++) {
      bench<<<1, 512>>>(out, sink, mode, partner); hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      printf("  %-34s %7.2f cycles per iteration (8 f64 fma + dpp64 + fma)\n", names[mode], (double)h / N_ITER);
    }
  }
}

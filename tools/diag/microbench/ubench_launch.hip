// Micro-benchmark: how long does the chip take just to dispatch and retire N small workgroups
// (64 or 128 threads, some LDS, many VGPRs) that do (almost) nothing?  Bounds F2's launch-side cost.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NV>
__global__ __launch_bounds__(64) void k_empty(float* out, int spin) {
  extern __shared__ float smem[];
  if (NV > 128) asm volatile("v_mov_b32 v240, 0" ::: "v240");
  else if (NV > 64) asm volatile("v_mov_b32 v100, 0" ::: "v100");
  smem[threadIdx.x] = 1.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
  if (smem[threadIdx.x] == 2.f) out[blockIdx.x] = 1.f;
}
template <int NV>
void run(const char* name, int gx, int gy, int lds, int spin, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_empty<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_empty<NV>, dim3(gx, gy), dim3(64), lds, 0, out, spin);
  hipEventRecord(e0);
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_empty<NV>, dim3(gx, gy), dim3(64), lds, 0, out, spin);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s grid %dx%d lds %5d spin %6d cycles: %7.1f us per launch\n", name, gx, gy, lds, spin, ms * 100.f);
}
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  for (int spin : {0, 10000, 40000}) {
    run<32>("few VGPRs", 63, 256, 1024, spin, out);
    run<32>("few VGPRs, 13 KB LDS", 63, 256, 13312, spin, out);
    run<100>("~100 VGPRs, 13 KB LDS", 63, 256, 13312, spin, out);
    run<248>("~248 VGPRs, 13 KB LDS", 63, 256, 13312, spin, out);
    run<248>("~248 VGPRs, 13 KB LDS, 1/4 the WGs", 16, 256, 13312, spin * 4, out);
  }
}

// Micro-benchmark: cost of __syncthreads() for 256 / 512 / 1024-thread workgroups (one workgroup on a CU),
// alone and with a little LDS traffic between the barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void bench(unsigned long long* out, int* sink, int mode) {
  __shared__ int buf[2048];
  const int tid = threadIdx.x;
  buf[tid] = tid; buf[tid + 1024] = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int acc = 0;
  for (int i = 0; i < 1000; i++) {
    if (mode == 1) { buf[(tid + i) & 1023] = acc; }
    __syncthreads();
    if (mode == 1) { acc += buf[(tid * 7 + i) & 1023]; }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * blockDim.x + tid] = acc;
  if (tid == 0) out[0] = t1 - t0;
}
int main() {
  unsigned long long* out; int* sink; hipMalloc(&out, 64); hipMalloc(&sink, 4096 * 4);
  for (int mode = 0; mode < 2; mode++)
    for (int th : {64, 256, 512, 1024}) {
      bench<<<1, th>>>(out, sink, mode); hipDeviceSynchronize();
      unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      printf("%4d threads, %s: %7.1f cycles per barrier\n", th, mode ? "LDS write + barrier + LDS read" : "bare barrier", (double)h / 1000.0);
    }
}

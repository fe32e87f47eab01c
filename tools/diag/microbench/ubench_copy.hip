// What streaming copy rate does this box reach, and with which launch shape?  (MI355X_MICROARCH.md: ~6.3 TB/s achievable.)
//   hipcc --offload-arch=gfx950 -O3 -o build/diag/ubench_copy tools/diag/microbench/ubench_copy.hip && build/diag/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT, int U, int MODE>   // MODE 0: plain, 1: nt load + nt store, 2: nt store only
__global__ __launch_bounds__(NT) void copy_k(f4* __restrict__ dst, const f4* __restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * NT;
  size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = MODE == 1 ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) { if (MODE >= 1) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
// each workgroup copies one contiguous chunk (what a row-per-wave kernel looks like)
template <int NT, int U, int MODE>
__global__ __launch_bounds__(NT) void copy_rows(f4* __restrict__ dst, const f4* __restrict__ src, size_t n16, int row16) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * (NT / 64) + w;
  const size_t base = row * row16;
  if (base + row16 > n16) return;
  for (int i = lane; i + (U - 1) * 64 < row16; i += U * 64) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = MODE == 1 ? __builtin_nontemporal_load(&src[base + i + u * 64]) : src[base + i + u * 64];
#pragma unroll
    for (int u = 0; u < U; u++) { if (MODE >= 1) __builtin_nontemporal_store(v[u], &dst[base + i + u * 64]); else dst[base + i + u * 64] = v[u]; }
  }
}
// the dense row kernel's shape: the whole row (32 x 16 bytes per lane) loaded, COMP dependent-free VALU operations per
// element plus two wave reductions, the whole row stored
template <int COMP, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void rows_full(f4* __restrict__ dst, const f4* __restrict__ src, size_t n16, int row16) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t row = (size_t)blockIdx.x * 4 + w;
  const size_t base = row * row16;
  if (base + row16 > n16) return;
  const f4* x = src + base; f4* g = dst + base;
  f4 v[32];
#pragma unroll
  for (int u = 0; u < 32; u++) v[u] = __builtin_nontemporal_load(&x[min(64 * u + lane, row16 - 1)]);
  float m = -1e30f;
#pragma unroll
  for (int u = 0; u < 32; u++) m = fmaxf(m, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 32; u++) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float t = v[u][e] - m;
#pragma unroll
      for (int c = 0; c < COMP; c++) t = __builtin_fmaf(t, 1.0001f, 0.5f);
      v[u][e] = t; sum += t;
    }
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int u = 0; u < 32; u++) __builtin_nontemporal_store(v[u] * inv, &g[min(64 * u + lane, row16 - 1)]);
}
// two passes over a row: running maximum and sum first (nothing stored), then the row is read AGAIN -- from L2 / the
// Infinity Cache if it is still there -- scaled and stored.  U chunks in flight; NTL: non-temporal loads in pass 1 / 2.
template <int U, int NT1, int NT2, int WPB>
__global__ __launch_bounds__(64 * WPB) void rows_two_pass(f4* __restrict__ dst, const f4* __restrict__ src, size_t n16, int row16) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t row = (size_t)blockIdx.x * WPB + w;
  const size_t base = row * row16;
  if (base + row16 > n16) return;
  const f4* x = src + base; f4* g = dst + base;
  float m = -1e30f, sum = 0.f;
  for (int i0 = 0; i0 < row16; i0 += 64 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const f4* a = &x[min(i0 + 64 * u + lane, row16 - 1)]; v[u] = NT1 ? __builtin_nontemporal_load(a) : *a; }
    float cm = m;
#pragma unroll
    for (int u = 0; u < U; u++) cm = fmaxf(cm, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    sum *= __expf(m - cm); m = cm;
#pragma unroll
    for (int u = 0; u < U; u++) sum += __expf(v[u].x - m) + __expf(v[u].y - m) + __expf(v[u].z - m) + __expf(v[u].w - m);
  }
  float M = m;
  for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
  sum *= __expf(m - M);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.f / sum;
  for (int i0 = 0; i0 < row16; i0 += 64 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const f4* a = &x[min(i0 + 64 * u + lane, row16 - 1)]; v[u] = NT2 ? __builtin_nontemporal_load(a) : *a; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      f4 o = {__expf(v[u].x - M), __expf(v[u].y - M), __expf(v[u].z - M), __expf(v[u].w - M)};
      __builtin_nontemporal_store(o * inv, &g[min(i0 + 64 * u + lane, row16 - 1)]);
    }
  }
}
// one row per WORKGROUP: each of its WPR waves keeps 32 / WPR chunks of the row in registers; the maximum and the sum cross
// the waves through LDS (two barriers per row); RPB rows per block one after the other
template <int WPR, int COMP>
__global__ __launch_bounds__(64 * WPR) void rows_split(f4* __restrict__ dst, const f4* __restrict__ src, size_t n16, int row16) {
  constexpr int U = 32 / WPR;
  __shared__ float red[2][8];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = (size_t)blockIdx.x * row16;
  if (base + row16 > n16) return;
  const f4* x = src + base; f4* g = dst + base;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(&x[min(64 * (U * w + u) + lane, row16 - 1)]);
  float m = -1e30f;
#pragma unroll
  for (int u = 0; u < U; u++) m = fmaxf(m, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[0][w] = m;
  __syncthreads();
  float M = red[0][0];
#pragma unroll
  for (int k = 1; k < WPR; k++) M = fmaxf(M, red[0][k]);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < U; u++) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float t = v[u][e] - M;
#pragma unroll
      for (int c = 0; c < COMP; c++) t = __builtin_fmaf(t, 1.0001f, 0.5f);
      v[u][e] = t; sum += t;
    }
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) red[1][w] = sum;
  __syncthreads();
  float S = 0.f;
#pragma unroll
  for (int k = 0; k < WPR; k++) S += red[1][k];
  const float inv = 1.f / S;
#pragma unroll
  for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u] * inv, &g[min(64 * (U * w + u) + lane, row16 - 1)]);
}
template <typename F> float timeit(F f, int reps = 10) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < reps; r++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const size_t bytes = (size_t)4 << 30;            // 4 GiB each way
  f4 *src, *dst; hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMemset(src, 1, bytes);
  const size_t n16 = bytes / 16;
#define RUN(NT, U, MODE, GRID) { float ms = timeit([&] { hipLaunchKernelGGL((copy_k<NT, U, MODE>), dim3(GRID), dim3(NT), 0, 0, dst, src, n16); }); \
    printf("grid-stride NT=%4d U=%d mode=%d grid=%6d: %.2f TB/s\n", NT, U, MODE, GRID, 2.0 * bytes / ms / 1e9); }
  RUN(256, 8, 1, 2048) RUN(256, 8, 0, 2048) RUN(256, 8, 2, 2048) RUN(256, 4, 1, 4096) RUN(256, 8, 1, 1024) RUN(256, 8, 1, 8192)
  RUN(512, 8, 1, 1024) RUN(1024, 4, 1, 1024) RUN(256, 16, 1, 2048) RUN(256, 2, 1, 16384) RUN(64, 8, 1, 8192)
  {
    float ms = timeit([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpyAsync D2D: %.2f TB/s\n", 2.0 * bytes / ms / 1e9);
  }
#define ROWS(NT, U, MODE) { const int row16 = 2000; const unsigned grid = (unsigned)((n16 / row16 + NT / 64 - 1) / (NT / 64)); \
    float ms = timeit([&] { hipLaunchKernelGGL((copy_rows<NT, U, MODE>), dim3(grid), dim3(NT), 0, 0, dst, src, n16, row16); }); \
    printf("row-per-wave (32 KB rows) NT=%4d U=%d mode=%d: %.2f TB/s\n", NT, U, MODE, 2.0 * (n16 / row16 * row16 * 16.0) / ms / 1e9); }
  ROWS(256, 8, 1) ROWS(256, 8, 0) ROWS(256, 8, 2) ROWS(256, 4, 1) ROWS(512, 8, 1) ROWS(128, 8, 1) ROWS(64, 8, 1)
#define FULL(COMP, WPE) { const int row16 = 2000; const unsigned grid = (unsigned)((n16 / row16 + 3) / 4); \
    float ms = timeit([&] { hipLaunchKernelGGL((rows_full<COMP, WPE>), dim3(grid), dim3(256), 0, 0, dst, src, n16, row16); }); \
    printf("whole row in registers, %2d VALU per element, %d waves per SIMD asked: %.2f TB/s\n", COMP, WPE, 2.0 * (n16 / row16 * row16 * 16.0) / ms / 1e9); }
  FULL(0, 2) FULL(0, 3) FULL(4, 2) FULL(8, 2) FULL(8, 3) FULL(12, 2) FULL(12, 3) FULL(16, 3)
#define TWOP(U, NT1, NT2, WPB) { const int row16 = 2000; const unsigned grid = (unsigned)((n16 / row16 + WPB - 1) / WPB); \
    float ms = timeit([&] { hipLaunchKernelGGL((rows_two_pass<U, NT1, NT2, WPB>), dim3(grid), dim3(64 * WPB), 0, 0, dst, src, n16, row16); }); \
    printf("two passes over the row, %d chunks in flight, nt loads %d/%d, %d waves per block: %.2f TB/s (algorithmic)\n", U, NT1, NT2, WPB, 2.0 * (n16 / row16 * row16 * 16.0) / ms / 1e9); }
  TWOP(8, 0, 0, 4) TWOP(8, 0, 1, 4) TWOP(8, 1, 1, 4) TWOP(4, 0, 1, 4) TWOP(8, 0, 1, 8) TWOP(8, 0, 1, 2) TWOP(16, 0, 1, 4)
#define SPLIT(WPR, COMP) { const int row16 = 2000; const unsigned grid = (unsigned)(n16 / row16); \
    float ms = timeit([&] { hipLaunchKernelGGL((rows_split<WPR, COMP>), dim3(grid), dim3(64 * WPR), 0, 0, dst, src, n16, row16); }); \
    printf("row split over %d waves of a workgroup, %2d VALU per element: %.2f TB/s\n", WPR, COMP, 2.0 * (n16 / row16 * row16 * 16.0) / ms / 1e9); }
  SPLIT(4, 0) SPLIT(4, 10) SPLIT(2, 10) SPLIT(8, 10) SPLIT(4, 20)
  return 0;
}

// Micro-benchmark: lone-wave cost of LDS read shapes (cycles per wave-instruction incl. one dependent add).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 256
typedef double d2 __attribute__((ext_vector_type(2)));
template <typename F>
__device__ unsigned long long timed(F f) {
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0);
  f();
  __builtin_amdgcn_s_waitcnt(0);
  return __builtin_amdgcn_s_memtime() - t0;
}
__global__ void bench(unsigned long long* out, double* sink, const int* labs) {
  __shared__ __align__(16) double lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.5;
  __syncthreads();
  const int lab = labs[lane];           // 29 distinct values
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long r[8];
  // 0: b64, lane-contiguous (conflict free), 8 independent per iteration, all issued then consumed
  r[0] = timed([&] { for (int i = 0; i < N_ITER; i++) { double v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = lds[lane + 64 * k + (i & 7) * 512];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] += v[k]; } });
  // 1: b64 gather over 29 labels (row of 30 doubles), 8 rows
  r[1] = timed([&] { for (int i = 0; i < N_ITER; i++) { double v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = lds[lab + 30 * k + (i & 7) * 256];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] += v[k]; } });
  // 2: b128 lane-contiguous (2 doubles per lane), 8 per iteration
  r[2] = timed([&] { for (int i = 0; i < N_ITER; i++) { d2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const d2*>(&lds[2 * lane + 128 * k + (i & 3) * 1024]);
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] += v[k].x + v[k].y; } });
  // 3: b64 broadcast (all lanes same address), 8 per iteration
  r[3] = timed([&] { for (int i = 0; i < N_ITER; i++) { double v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = lds[k * 30 + (i & 7) * 256];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] += v[k]; } });
  // 4: b128 broadcast
  r[4] = timed([&] { for (int i = 0; i < N_ITER; i++) { d2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const d2*>(&lds[2 * k + (i & 7) * 256]);
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] += v[k].x + v[k].y; } });
  // 5: b32 gather over 29 labels
  const float* lf = reinterpret_cast<const float*>(lds);
  float fa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  r[5] = timed([&] { for (int i = 0; i < N_ITER; i++) { float v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = lf[lab + 30 * k + (i & 7) * 256];
#pragma unroll
    for (int k = 0; k < 8; k++) fa[k] += v[k]; } });
  // 6: ds_write_b64 lane contiguous x8
  r[6] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) lds[lane + 64 * k + (i & 7) * 512] = a[k]; } });
  // 7: ds_write_b32 scatter by label-ish rank
  float* lw = reinterpret_cast<float*>(lds);
  r[7] = timed([&] { for (int i = 0; i < N_ITER; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) lw[(lane * 37 + k * 64) & 1023] = fa[k]; } });
  double s = 0; for (int k = 0; k < 8; k++) s += a[k] + fa[k];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[lane];
  if (threadIdx.x == 0) for (int k = 0; k < 8; k++) out[blockIdx.x * 8 + k] = r[k];
}
int main() {
  const char* names[8] = {"ds_read_b64 contiguous", "ds_read_b64 gather(29 labels)", "ds_read_b128 contiguous", "ds_read_b64 broadcast",
                          "ds_read_b128 broadcast", "ds_read_b32 gather(29 labels)", "ds_write_b64 contiguous", "ds_write_b32 scatter"};
  std::vector<int> labs(64); for (int i = 0; i < 64; i++) labs[i] = (i * 7 + 3) % 29;
  int* dl; hipMalloc(&dl, 256); hipMemcpy(dl, labs.data(), 256, hipMemcpyHostToDevice);
  for (int waves : {1, 4, 8}) {
    unsigned long long* out; double* sink; hipMalloc(&out, 4096); hipMalloc(&sink, 8 * 1024 * 8);
    bench<<<1, 64 * waves>>>(out, sink, dl); hipDeviceSynchronize();
    std::vector<unsigned long long> h(8); hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost);
    printf("== %d wave(s) on one CU: cycles per LDS wave-instruction (8 issued back to back, then consumed)\n", waves);
    for (int k = 0; k < 8; k++) printf("  %-32s %7.2f\n", names[k], (double)h[k] / (N_ITER * 8.0));
  }
}

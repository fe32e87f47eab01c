// Micro-benchmark: the F1 chain's per-block ring gather (PPL label rows + the blank row, 2 x ds_read_b128 each)
// as a function of the row stride and of how many other waves hammer the LDS at the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 512
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PPL, int ROW>
__device__ unsigned long long gather(const float* ring, const int (&lab)[4], float& acc) {
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0);
  for (int i = 0; i < N_ITER; i++) {
    const float* blk = ring + (i & 7) * 32 * ROW;
    f4 e[PPL][2], b[2];
#pragma unroll
    for (int r = 0; r < PPL; r++) {
      const f4* src = reinterpret_cast<const f4*>(blk + lab[r] * ROW);
      e[r][0] = src[0]; e[r][1] = src[1];
    }
    const f4* sb = reinterpret_cast<const f4*>(blk);
    b[0] = sb[0]; b[1] = sb[1];
#pragma unroll
    for (int r = 0; r < PPL; r++) acc += e[r][0][0] + e[r][1][3] + e[r][0][2];
    acc += b[0][1] + b[1][2];
  }
  __builtin_amdgcn_s_waitcnt(0);
  return __builtin_amdgcn_s_memtime() - t0;
}
__global__ void bench(unsigned long long* out, float* sink, const int* labs) {
  __shared__ __align__(16) float lds[8 * 32 * 20];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 32 * 20; i += blockDim.x) lds[i] = i * 0.5f;
  __syncthreads();
  int lab[4];
  for (int r = 0; r < 4; r++) lab[r] = labs[(lane * 4 + r + 64 * wave) & 255];
  float acc = 0.f;
  unsigned long long r[8];
  r[0] = gather<4, 8>(lds, lab, acc);
  r[1] = gather<4, 12>(lds, lab, acc);
  r[2] = gather<4, 20>(lds, lab, acc);
  r[3] = gather<1, 8>(lds, lab, acc);
  r[4] = gather<2, 8>(lds, lab, acc);
  int same[4] = {3, 5, 7, 9};
  r[5] = gather<4, 8>(lds, same, acc);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (lane == 0) for (int k = 0; k < 6; k++) out[wave * 8 + k] = r[k];
}
int main() {
  const char* names[6] = {"PPL4 row 32B", "PPL4 row 48B", "PPL4 row 80B", "PPL1 row 32B", "PPL2 row 32B", "PPL4 uniform rows"};
  std::vector<int> labs(256); unsigned s = 12345; for (int i = 0; i < 256; i++) { s = s * 1664525u + 1013904223u; labs[i] = 1 + (s >> 8) % 28; }
  int* dl; hipMalloc(&dl, 1024); hipMemcpy(dl, labs.data(), 1024, hipMemcpyHostToDevice);
  for (int waves : {1, 2, 4, 8}) {
    unsigned long long* out; float* sink; hipMalloc(&out, 4096); hipMalloc(&sink, 8 * 1024 * 8);
    bench<<<1, 64 * waves>>>(out, sink, dl); hipDeviceSynchronize();
    std::vector<unsigned long long> h(64); hipMemcpy(h.data(), out, 512, hipMemcpyDeviceToHost);
    printf("== %d wave(s) on one CU: cycles per block gather (wave 0)\n", waves);
    for (int k = 0; k < 6; k++) printf("  %-20s %8.1f\n", names[k], (double)h[k] / N_ITER);
  }
}

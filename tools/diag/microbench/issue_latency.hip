// Microbenchmark (diagnostic, not part of the library): cycles per instruction of ONE wave64 on gfx950 for dependent and
// independent streams of the instruction kinds the CTC chain kernels are made of.  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/diag/microbench/issue_latency.hip -o build/diag/issue_latency   (here; the binary
//   travels with gpurun when it is taken off .gpurunignore), then on the box: ./build/diag/issue_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define N_OUTER 256

template <int KIND>
__global__ __launch_bounds__(64) void bench(unsigned long long* out, double* sink, int waves_note) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
  double a2 = 1.1, a3 = 1.2, a4 = 1.3;
  float f = 1.0f + threadIdx.x * 1e-6f, g = 1.0000001f, h = 1e-6f;
  float f2 = 1.1f, f3 = 1.2f, f4 = 1.3f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p = {f, f2}, q = {g, g}, r = {h, h}, p2 = {f3, f4};
  int iv = threadIdx.x;
  int si = blockIdx.x;
  __shared__ double lds[256];
  lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_OUTER; i++) {
    if (KIND == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }                       // dependent f64 fma
    if (KIND == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));) }                       // dependent f32 fma
    if (KIND == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(q), "v"(r));) }                    // dependent packed f32 fma
    if (KIND == 3) {                                                                                                       // 4 independent f64 fma streams
      for (int k = 0; k < 4; k++) { asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                                 : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c)); }
    }
    if (KIND == 4) {                                                                                                       // 4 independent f32 fma streams
      for (int k = 0; k < 4; k++) { asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                                 : "+v"(f), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(g), "v"(h)); }
    }
    if (KIND == 5) {                                                                                                       // 2 independent packed streams
      for (int k = 0; k < 8; k++) { asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p), "+v"(p2) : "v"(q), "v"(r)); }
    }
    if (KIND == 6) {                                                                                                       // f64: fma -> 2 dpp moves -> fma ... (the chain's cross-lane hop)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x138, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x138, 0xf, 0xf, true);
        a = __hiloint2double(hi, lo);
      }
    }
    if (KIND == 7) {                                                                                                       // f32: fma -> 1 dpp move -> fma
      for (int k = 0; k < 8; k++) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f) : "v"(g), "v"(h));
      }
    }
    if (KIND == 8) { REP16(asm volatile("s_add_u32 %0, %0, 1" : "+s"(si) : : "scc");) }                                            // dependent scalar adds
    if (KIND == 9) { REP16(asm volatile("v_add_u32 %0, %0, %0" : "+v"(iv));) }                                           // dependent 32-bit VALU
    if (KIND == 10) {                                                                                                      // dependent LDS read (address from data)
      for (int k = 0; k < 16; k++) { asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(iv)); iv &= 0xfc; }
    }
    if (KIND == 11) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }                                  // dependent f64 mul
    if (KIND == 12) { REP16(asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a));) }
    if (KIND == 13) {                                                                                                      // f64 fma chain with an independent scalar op in between
      REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_add_u32 %1, %1, 1" : "+v"(a), "+s"(si) : "v"(b), "v"(c) : "scc");)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = a + a2 + a3 + a4 + f + f2 + f3 + f4 + p.x + p.y + p2.x + p2.y + iv + si;
}

template <int KIND>
void run(const char* name, int instr_per_iter, int nwaves_per_cu) {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 8 * 4096); hipMalloc(&sink, 8 * 64 * 4096);
  // nwaves_per_cu workgroups of one wave each land on one CU only if the grid is large; here: grid = 256 CUs x n
  const int grid = 256 * nwaves_per_cu;
  bench<KIND><<<grid, 64>>>(out, sink, 0);
  bench<KIND><<<grid, 64>>>(out, sink, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), out, 8 * grid, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("%-58s %2d wave(s)/CU: %6.2f memtime ticks per instruction\n", name, nwaves_per_cu, s / grid / (double)(N_OUTER * instr_per_iter));
  hipFree(out); hipFree(sink);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("start\n");
  for (int n : {1, 4, 8, 16}) {
    run<0>("dependent v_fma_f64", 16, n);
    run<11>("dependent v_mul_f64", 16, n);
    run<12>("dependent v_ldexp_f64", 16, n);
    run<1>("dependent v_fma_f32", 16, n);
    run<2>("dependent v_pk_fma_f32", 16, n);
    run<9>("dependent v_add_u32", 16, n);
    run<3>("4 independent v_fma_f64 streams", 16, n);
    run<4>("4 independent v_fma_f32 streams", 16, n);
    run<5>("2 independent v_pk_fma_f32 streams", 16, n);
    run<6>("f64: fma, s_nop 1, 2 dpp moves (per group of 4)", 8 * 4, n);
    run<7>("f32: fma, s_nop 1, 1 dpp move (per group of 3)", 8 * 3, n);
    run<8>("dependent s_add_u32", 16, n);
    run<13>("v_fma_f64 chain + independent s_add between (per pair)", 32, n);
    run<10>("dependent ds_read_b32 + wait (per pair)", 32, n);
    printf("\n");
  }
  return 0;
}

// Greedy CTC decode: per-frame argmax, then blank/repeat collapse with an
// in-workgroup stream compaction.  Restates src/decoders/ctc_decoder.cpp:443-490.
// One workgroup per utterance; HBM-bound (reads the (T,V) slab once, writes the
// (T) int64 row once).  Small alphabets (V <= 64) stage a 256-frame tile in LDS
// with coalesced loads and take one frame per lane; wide alphabets take one
// frame per wavefront with a lane-strided scan and a wave arg-max.
#include "common.h"

namespace e2e {
namespace {

constexpr int kThreads = 256;
constexpr int kSmallV = 64;

struct GreedyParams {
  const void* x; int64_t sB, sT, sV; const int64_t* x_len;
  int B, T, V, blank; int64_t* out; int64_t* out_len;
};

// torch CPU argmax semantics: first maximum wins, NaN counts as the maximum
template <typename F>
__device__ __forceinline__ bool better(F cand, int cand_i, F best, int best_i) {
  const bool cn = cand != cand, bn = best != best;
  if (bn) return cn && cand_i < best_i;
  if (cn) return true;
  return cand > best || (cand == best && cand_i < best_i);
}

template <typename IO>
__global__ __launch_bounds__(kThreads) void ctc_greedy_kernel(GreedyParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int sym[kThreads];
  __shared__ int wave_tot[kThreads / 64];
  __shared__ int carry_prev, carry_n;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V, Tmax = p.T, blank = p.blank;
  const IO* x = reinterpret_cast<const IO*>(p.x) + (int64_t)b * p.sB;
  int64_t* out = p.out + (int64_t)b * Tmax;
  int64_t Tq = p.x_len[b];
  const int T = Tq < 0 ? 0 : (Tq > Tmax ? Tmax : (int)Tq);
  if (tid == 0) { carry_prev = blank; carry_n = 0; }
  __syncthreads();
  const bool rows_contig = (p.sV == 1 && p.sT == V);
  const int ldv = V | 1;                                                // LDS row stride (odd)
  const unsigned vmagic = (1u << 20) / (unsigned)V + 1u;                // i / V for i < 256 * 64
  IO* tile = reinterpret_cast<IO*>(smem);

  for (int t0 = 0; t0 < T; t0 += kThreads) {
    const int nt = min(kThreads, T - t0);
    int my = blank;
    if (V <= kSmallV) {
      // (LDS rows have an odd stride: with an even V the 64 lanes' row scans would hit the same few banks --
      // V = 32 ran 2.3x slower than V = 29)
      if (rows_contig && ldv == V) {
        const IO* src = x + (int64_t)t0 * V;
        for (int i = tid; i < nt * V; i += kThreads) tile[i] = src[i];
      } else if (rows_contig) {
        const IO* src = x + (int64_t)t0 * V;
        for (int i = tid; i < nt * V; i += kThreads) {
          const int r = (int)(((unsigned)i * vmagic) >> 20);
          tile[i + r] = src[i];                                         // r*ldv + (i - r*V) with ldv = V + 1
        }
      } else {
        for (int i = tid; i < nt * V; i += kThreads) {
          const int r = (int)(((unsigned)i * vmagic) >> 20), v = i - r * V;
          tile[r * ldv + v] = x[(int64_t)(t0 + r) * p.sT + (int64_t)v * p.sV];
        }
      }
      __syncthreads();
      if (tid < nt) {
        const IO* row = tile + tid * ldv;
        IO bv = row[0]; int bi = 0;
        for (int v = 1; v < V; v++) { const IO c = row[v]; if (better(c, v, bv, bi)) { bv = c; bi = v; } }
        my = bi;
      }
    } else {
      for (int r = wid; r < nt; r += kThreads / 64) {
        const IO* row = x + (int64_t)(t0 + r) * p.sT;
        IO bv = row[0]; int bi = 0;   // every lane starts from element 0: a valid candidate
        for (int v = lane; v < V; v += 64) { const IO c = row[(int64_t)v * p.sV]; if (better(c, v, bv, bi)) { bv = c; bi = v; } }
        for (int o = 32; o > 0; o >>= 1) {
          const IO ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
          if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) sym[r] = bi;
      }
      __syncthreads();
      if (tid < nt) my = sym[tid];
      __syncthreads();
    }
    // collapse: emit iff sym != blank && sym != previous frame's sym (ctc_decoder.cpp:475-481)
    sym[tid] = my;
    __syncthreads();
    const int prev = (tid == 0) ? carry_prev : sym[tid - 1];
    const int flag = (tid < nt && my != blank && my != prev) ? 1 : 0;
    // exclusive scan of flags over the workgroup
    int incl = flag;
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int base = carry_n;
    for (int w = 0; w < wid; w++) base += wave_tot[w];
    if (flag) out[base + incl - 1] = my;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < kThreads / 64; w++) tot += wave_tot[w];
      carry_n += tot;
      carry_prev = sym[nt - 1];
    }
    __syncthreads();
  }
  const int n = carry_n;
  for (int i = n + tid; i < Tmax; i += kThreads) out[i] = 0;   // zeros_like padding (Q5)
  if (tid == 0) p.out_len[b] = n;
}

}  // namespace

int launch_greedy(const void* x, int dtype, int64_t sB, int64_t sT, int64_t sV, const int64_t* x_len,
                  int B, int T, int V, int blank, int64_t* out, int64_t* out_len, hipStream_t stream) {
  GreedyParams p{x, sB, sT, sV, x_len, B, T, V, blank, out, out_len};
  if (B == 0) return E2E_OK;
  const size_t esz = dtype == E2E_F32 ? 4 : 8;
  const size_t lds = V <= kSmallV ? (size_t)kThreads * (V | 1) * esz : 16;
  if (dtype == E2E_F32)
    hipLaunchKernelGGL(ctc_greedy_kernel<float>, dim3(B), dim3(kThreads), lds, stream, p);
  else
    hipLaunchKernelGGL(ctc_greedy_kernel<double>, dim3(B), dim3(kThreads), lds, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_greedy_kernel launch");
  return E2E_OK;
}

}  // namespace e2e

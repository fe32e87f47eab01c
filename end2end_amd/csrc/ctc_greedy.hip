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

// (16-bit values travel between lanes as floats: the conversion is exact both ways)
template <typename IO> __device__ __forceinline__ IO shfl_xor_io(IO v, int o) {
  if constexpr (sizeof(IO) == 2) return (IO)__shfl_xor((float)v, o, 64); else return __shfl_xor(v, o, 64);
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
          const IO ov = shfl_xor_io(bv, o); const int oi = __shfl_xor(bi, o, 64);
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

// ---- streaming form for contiguous rows of a small alphabet ------------------------------------------------------
// The kernel above synchronises the workgroup seven times per 256 frames and waits for every tile's loads before it
// touches them.  Here each WAVE streams 64-frame chunks on its own: the chunk after the one being worked on is already
// requested (16-byte loads into registers), the chunk itself goes through a wave-private LDS tile (lane = frame, odd
// row stride), and the only workgroup barriers are three per super-tile of 4096 frames, around the collapse.  The
// symbols of a super-tile and its compacted output live in LDS as bytes (V <= 64); the output is written coalesced.
// LDS per workgroup stays below 40 KB so that four workgroups share a CU: B = 1024 utterances then run in one round.
constexpr int kChunk = 64;           // frames per wave and chunk
#ifndef E2E_GREEDY_SUPER
#define E2E_GREEDY_SUPER 2048
#endif
#ifndef E2E_GREEDY_DEPTH
#define E2E_GREEDY_DEPTH 1
#endif
constexpr int kSuper = E2E_GREEDY_SUPER;   // frames per super-tile (round 6: 2048 -- BASELINE configs[2]'s 1500 frames are ONE super-tile, so a
                                     // wave's stream of chunk loads is not cut in the middle by a collapse phase and a fresh round trip)
constexpr int kDepth = E2E_GREEDY_DEPTH;   // chunks a wave has asked for beyond the one it works on.  (Round 6, one process, B=1024, T=1500, V=29: one
                                     // chunk ahead 34.6 us, two 35.2, three 39.8; with the old super-tile of 1024 frames 35.6 -- the kernel moves its
                                     // 190 MB at 5.4-5.5 TB/s, the box's streaming rate, and a deeper pipeline has nothing to hide.)
constexpr int kStreamWaves = 4;
constexpr int kMaxPf = 16;           // 16-byte loads per lane and chunk (V <= 64 floats, or V <= 32 doubles)

template <typename IO, int NPF>      // NPF >= ceil(64 * V * sizeof(IO) / 1024): 16-byte pieces per lane and chunk
__global__ __launch_bounds__(64 * kStreamWaves) void ctc_greedy_stream_kernel(GreedyParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  typedef int i4 __attribute__((ext_vector_type(4)));
  constexpr int EPV = 16 / (int)sizeof(IO);                       // elements per 16-byte load
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int V = p.V, Tmax = p.T, blank = p.blank;
  const int ldv = V | 1;
  const IO* x = reinterpret_cast<const IO*>(p.x) + (int64_t)b * p.sB;
  int64_t* out = p.out + (int64_t)b * Tmax;
  int64_t Tq = p.x_len[b];
  const int T = Tq < 0 ? 0 : (Tq > Tmax ? Tmax : (int)Tq);
  IO* tile = reinterpret_cast<IO*>(smem) + (size_t)wid * kChunk * ldv;                 // wave-private
  unsigned char* sym = smem + (size_t)kStreamWaves * kChunk * ldv * sizeof(IO);
  unsigned char* comp = sym + kSuper;
  __shared__ int wave_tot[kStreamWaves];
  __shared__ int carry_prev, carry_n;
  if (tid == 0) { carry_prev = blank; carry_n = 0; }
  const unsigned vmagic = (1u << 20) / (unsigned)V + 1u;          // i / V for i < 64 * 64

  for (int s0 = 0; s0 < T; s0 += kSuper) {
    const int sn = min(kSuper, T - s0);
    const int nchunks = (sn + kChunk - 1) / kChunk;
    // ---- phase 1: per-frame arg-max, one wave per chunk, the next chunk's loads in flight ----
    // (every load is a whole 16-byte piece at a clamped index: pieces past the chunk, or past the utterance's slab of
    // Tmax frames, re-read the slab's last piece -- frames past the utterance's end never reach sym[])
    // (kDepth register sets rotate through a loop unrolled kDepth times: a set is never copied, a copy would wait for its loads)
    i4 pf[kDepth][NPF];
    const int slab_pieces = (int)(((int64_t)Tmax * V) / EPV);
    const i4* const xp = reinterpret_cast<const i4*>(x);
    auto request = [&](int c, i4 (&dst)[NPF]) {
      const int p0 = (s0 + c * kChunk) * V / EPV;                 // (64 * V is a multiple of EPV)
#pragma unroll
#ifndef E2E_GREEDY_NT_LOADS         // (round 6, one process, configs[2]: plain loads 32.5 us, non-temporal ones 33.2)
#define E2E_GREEDY_NT_LOADS 0
#endif
      for (int u = 0; u < NPF; u++) dst[u] = E2E_GREEDY_NT_LOADS ? __builtin_nontemporal_load(&xp[min(p0 + lane + 64 * u, slab_pieces - 1)])
                                                                 : xp[min(p0 + lane + 64 * u, slab_pieces - 1)];
    };
#pragma unroll
    for (int d = 0; d < kDepth; d++) if (wid + d * kStreamWaves < nchunks) request(wid + d * kStreamWaves, pf[d]);
    auto work = [&](int c, i4 (&cur)[NPF]) {
      // park the chunk in the tile (row stride ldv: element i of the chunk lands at i + i / V when V is even)
#pragma unroll
      for (int u = 0; u < NPF; u++) {
        const int e = (lane + 64 * u) * EPV;
        if (e < kChunk * V) {
          if (ldv == V) *reinterpret_cast<i4*>(&tile[e]) = cur[u];
          else {
            const IO* q = reinterpret_cast<const IO*>(&cur[u]);
#pragma unroll
            for (int k = 0; k < EPV; k++) tile[e + k + (int)(((unsigned)(e + k) * vmagic) >> 20)] = q[k];
          }
        }
      }
      if (c + kDepth * kStreamWaves < nchunks) request(c + kDepth * kStreamWaves, cur);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (wave-private tile: no barrier)
      const int t = c * kChunk + lane;                            // frame within the super-tile
      const IO* row = tile + lane * ldv;
      // first maximum; NaN counts as the maximum (torch.argmax): the plain scan ignores NaNs, a row that has one is redone
      IO bv = row[0]; int bi = 0; bool has_nan = bv != bv;
      int v = 1;
      for (; v + 3 < V; v += 4) {                                  // four LDS reads in flight per trip
        const IO c0 = row[v], c1 = row[v + 1], c2 = row[v + 2], c3 = row[v + 3];
        has_nan |= (c0 != c0) | (c1 != c1) | (c2 != c2) | (c3 != c3);
        if (c0 > bv) { bv = c0; bi = v; }
        if (c1 > bv) { bv = c1; bi = v + 1; }
        if (c2 > bv) { bv = c2; bi = v + 2; }
        if (c3 > bv) { bv = c3; bi = v + 3; }
      }
      for (; v < V; v++) { const IO cv = row[v]; has_nan |= cv != cv; if (cv > bv) { bv = cv; bi = v; } }
      if (has_nan) {
        bv = row[0]; bi = 0;
        for (int v2 = 1; v2 < V; v2++) { const IO cv = row[v2]; if (better(cv, v2, bv, bi)) { bv = cv; bi = v2; } }
      }
      if (t < sn) sym[t] = (unsigned char)bi;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the tile is rewritten by the next chunk
    };
    for (int c = wid; c < nchunks; c += kDepth * kStreamWaves) {
#pragma unroll
      for (int d = 0; d < kDepth; d++) if (c + d * kStreamWaves < nchunks) work(c + d * kStreamWaves, pf[d]);
    }
    __syncthreads();
    // ---- phase 2: collapse (emit iff sym != blank && sym != previous frame's sym, ctc_decoder.cpp:475-481) ----
    constexpr int kPer = kSuper / (64 * kStreamWaves);            // consecutive frames per thread
    const int f0 = tid * kPer;
    int prev = f0 == 0 ? carry_prev : (f0 - 1 < sn ? (int)sym[f0 - 1] : blank);
    int mine[kPer], cnt = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int sy = f0 + k < sn ? (int)sym[f0 + k] : -1;
      mine[k] = (sy >= 0 && sy != blank && sy != prev) ? sy : -1;
      cnt += mine[k] >= 0;
      prev = sy;
    }
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int pos = incl - cnt, total = 0;
    for (int w = 0; w < kStreamWaves; w++) { if (w < wid) pos += wave_tot[w]; total += wave_tot[w]; }
#pragma unroll
    for (int k = 0; k < kPer; k++) if (mine[k] >= 0) comp[pos++] = (unsigned char)mine[k];
    const int base = carry_n, last = (int)sym[sn - 1];
    __syncthreads();
    for (int i = tid; i < total; i += 64 * kStreamWaves) out[base + i] = (int64_t)comp[i];
    if (tid == 0) { carry_n = base + total; carry_prev = last; }
    __syncthreads();
  }
  __syncthreads();
  const int n = carry_n;
  for (int i = n + tid; i < Tmax; i += 64 * kStreamWaves) out[i] = 0;   // zeros_like padding (Q5)
  if (tid == 0) p.out_len[b] = n;
}

}  // namespace

int launch_greedy(const void* x, int dtype, int64_t sB, int64_t sT, int64_t sV, const int64_t* x_len,
                  int B, int T, int V, int blank, int64_t* out, int64_t* out_len, hipStream_t stream) {
  GreedyParams p{x, sB, sT, sV, x_len, B, T, V, blank, out, out_len};
  if (B == 0) return E2E_OK;
  const size_t esz = dtype == E2E_F32 ? 4 : dtype_is_16bit(dtype) ? 2 : 8;
  {
    // contiguous 16-byte aligned rows of a small alphabet: the streaming kernel
    const size_t lds_stream = (size_t)kStreamWaves * kChunk * (V | 1) * esz + 2 * kSuper;
    // (every utterance's slab of T*V elements starts on a 16-byte boundary and is a whole number of 16-byte pieces: the
    // kernel's clamped piece loads then never leave the slab)
    const bool aligned = reinterpret_cast<uintptr_t>(x) % 16 == 0 && (sB * (int64_t)esz) % 16 == 0 &&
                         ((int64_t)T * V * (int64_t)esz) % 16 == 0;
    if (sV == 1 && sT == V && aligned && V >= 1 && (size_t)V * esz <= 16 * kMaxPf && lds_stream <= 60 * 1024) {
      const int npf = (int)((64 * (size_t)V * esz + 1023) / 1024);
      if (dtype == E2E_F32) {
        if (npf <= 4) hipLaunchKernelGGL((ctc_greedy_stream_kernel<float, 4>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else if (npf <= 8) hipLaunchKernelGGL((ctc_greedy_stream_kernel<float, 8>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else hipLaunchKernelGGL((ctc_greedy_stream_kernel<float, 16>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
      } else if (dtype == E2E_F16) {          // (16-bit logits: compared as they are -- the ordering of the source dtype, as torch.argmax sees it)
        if (npf <= 4) hipLaunchKernelGGL((ctc_greedy_stream_kernel<f16_t, 4>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else if (npf <= 8) hipLaunchKernelGGL((ctc_greedy_stream_kernel<f16_t, 8>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else hipLaunchKernelGGL((ctc_greedy_stream_kernel<f16_t, 16>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
      } else if (dtype == E2E_BF16) {
        if (npf <= 4) hipLaunchKernelGGL((ctc_greedy_stream_kernel<bf16_t, 4>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else if (npf <= 8) hipLaunchKernelGGL((ctc_greedy_stream_kernel<bf16_t, 8>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else hipLaunchKernelGGL((ctc_greedy_stream_kernel<bf16_t, 16>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
      } else {
        if (npf <= 4) hipLaunchKernelGGL((ctc_greedy_stream_kernel<double, 4>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else if (npf <= 8) hipLaunchKernelGGL((ctc_greedy_stream_kernel<double, 8>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
        else hipLaunchKernelGGL((ctc_greedy_stream_kernel<double, 16>), dim3(B), dim3(64 * kStreamWaves), lds_stream, stream, p);
      }
      E2E_HIP_CHECK(hipGetLastError(), "ctc_greedy_stream_kernel launch");
      return E2E_OK;
    }
  }
  const size_t lds = V <= kSmallV ? (size_t)kThreads * (V | 1) * esz : 16;
  if (dtype == E2E_F32)
    hipLaunchKernelGGL(ctc_greedy_kernel<float>, dim3(B), dim3(kThreads), lds, stream, p);
  else if (dtype == E2E_F16)
    hipLaunchKernelGGL(ctc_greedy_kernel<f16_t>, dim3(B), dim3(kThreads), lds, stream, p);
  else if (dtype == E2E_BF16)
    hipLaunchKernelGGL(ctc_greedy_kernel<bf16_t>, dim3(B), dim3(kThreads), lds, stream, p);
  else
    hipLaunchKernelGGL(ctc_greedy_kernel<double>, dim3(B), dim3(kThreads), lds, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_greedy_kernel launch");
  return E2E_OK;
}

}  // namespace e2e

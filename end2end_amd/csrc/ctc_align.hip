// Viterbi forced alignment on the CTC lattice (and its blank-free ASG variant): the max-plus sibling of the loss.
//
// Replaces pytorch_end2end/utils/alignment.py:50-106 (_get_alignment_ctc_1d), :10-47 (_get_alignment_asg_1d) and the
// batch driver :109-138 (get_alignment_3d), which run as numba-jitted Python loops on the host, one OS thread per
// utterance.  Semantics restated exactly, including what the Python leaves implicit: alpha is float64 whatever the
// input; candidates are compared with a strict ">" in the order stay, i-1, i-2 (ties keep the earlier candidate); the
// skip needs "i - 2 > 0" (:86); back-pointers of cells outside the band are 0 (np.zeros_like, :74); too few frames for
// the labelling is not rejected (the back-trace then walks -inf cells).
//
// One workgroup per utterance.  The forward sweep keeps two alpha rows in LDS (f64) and writes one back-pointer CODE
// per cell and step to the workspace (0 stay, 1 from i-1, 2 from i-2, 3 "cell 0": outside the band), T*L bytes per
// utterance, coalesced.  The back-trace is serial in t, so it runs out of LDS: the workgroup stages the codes of a
// chunk of steps (coalesced), one thread walks the chunk, the labels are written out coalesced.
#include "common.h"

namespace e2e {
namespace {

constexpr int kThreads = 256;
constexpr int kChunkBytes = 32 * 1024;     // back-pointer codes staged per back-trace chunk

struct AlignParams {
  const void* lp; int64_t sB, sT, sV;
  const int64_t* targets; int64_t tgt_stride;
  const int64_t* x_len; const int64_t* t_len;
  int B, T, V, Smax, Lmax, Lpad, blank, is_ctc;        // Lpad: Lmax rounded up to 16 -- the stride of a row of codes
  int64_t* out; int64_t pad;
  unsigned char* bp;       // [B][T][Lpad]
};

__device__ __forceinline__ double ninf() { return -__builtin_huge_val(); }

template <typename IO>
__global__ __launch_bounds__(kThreads) void ctc_align_kernel(AlignParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Tmax = p.T, V = p.V, Lmax = p.Lmax, Lpad = p.Lpad;
  double* row0 = reinterpret_cast<double*>(smem);            // [Lmax]
  double* row1 = row0 + Lmax;                                // [Lmax]
  int* ext = reinterpret_cast<int*>(row1 + Lmax);            // [Lmax] label of every cell
  int64_t* bestbuf = reinterpret_cast<int64_t*>(ext + ((Lmax + 3) & ~3));     // [chunk rows] (16-byte aligned, like what follows)
  unsigned char* codes = reinterpret_cast<unsigned char*>(bestbuf + kChunkBytes / 64);   // [chunk rows][L]
  __shared__ int s_cell;

  int64_t* out = p.out + (int64_t)b * Tmax;
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  const bool bad_len = Tq < 0 || Tq > Tmax || Sq < 0 || Sq > p.Smax;
  const int T = bad_len ? 0 : (int)Tq, S = bad_len ? 0 : (int)Sq;
  for (int t = T + tid; t < Tmax; t += kThreads) out[t] = p.pad;             // frames past the utterance (:132)
  if (T < 1) return;
  const int64_t* tg = p.targets + (int64_t)b * p.tgt_stride;
  const int L = p.is_ctc ? 2 * S + 1 : S;

  // labels of the cells; a label outside the alphabet cannot be looked up: the row is left as padding
  int bad = 0;
  for (int i = tid; i < L; i += kThreads) {
    int64_t lab = p.blank;
    if (!p.is_ctc) lab = tg[i];
    else if (i & 1) lab = tg[i >> 1];
    bad |= (lab < 0) | (lab >= V);
    ext[i] = (int)lab;
  }
  if (__syncthreads_or(bad)) {
    for (int t = tid; t < T; t += kThreads) out[t] = p.pad;
    return;
  }
  // :65-70 / :22-24: nothing to align
  if (L <= (p.is_ctc ? 1 : 0) || T == 1) {
    for (int t = tid; t < T; t += kThreads) out[t] = 0;
    __syncthreads();
    if (tid == 0 && T == 1 && S >= 1) out[0] = tg[0];
    return;
  }

  const IO* lp = reinterpret_cast<const IO*>(p.lp) + (int64_t)b * p.sB;
  unsigned char* bp = p.bp + (size_t)b * (size_t)Tmax * (size_t)Lpad;
  auto LP = [&](int t, int v) -> double { return (double)lp[(int64_t)t * p.sT + (int64_t)v * p.sV]; };

  // ---- forward sweep ----
  for (int i = tid; i < L; i += kThreads) {
    double a = ninf();
    if (i == 0) a = LP(0, ext[0]);
    if (i == 1 && p.is_ctc) a = LP(0, ext[1]);
    row0[i] = a;
  }
  __syncthreads();
  if (L <= 2 * kThreads) {
    // The usual widths (two cells per thread).  A step is one trip to LDS and back -- barrier, the three neighbours of both
    // cells read together, compare / select / add, the row and the codes written, barrier -- and nothing else: the
    // emissions are asked for two sets of steps ahead (register sets, below; fetches and steps unconditional, rows
    // clamped, the stores of the steps past the end switched off), the barrier waits for LDS
    // only (the codes are read after the sweep, behind a fence), no branch around a cell.  Same comparisons in the same
    // order as the loop below, which keeps the wider lattices.  B=64, T=1000, S<=200: 1.35 -> 0.39 ms with the back-trace's changes.
    bool live[2], allow1[2], allow2[2]; int jc[2], jm1[2], jm2[2]; int64_t off[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int i = tid + c * kThreads;
      live[c] = i < L;
      jc[c] = min(i, L - 1);
      const int lab = ext[jc[c]];
      jm1[c] = max(jc[c] - 1, 0); jm2[c] = max(jc[c] - 2, 0);
      allow1[c] = live[c] && i > 0;
      allow2[c] = live[c] && p.is_ctc && lab != p.blank && i - 2 > 0 && ext[jm2[c]] != lab;
      off[c] = (int64_t)lab * p.sV;
    }
    // The emissions of kSet steps sit in one of THREE register sets; the loop is unrolled over all three: [load Z] steps
    // of X, [load X] steps of Y, [load Y] steps of Z.  A set is refilled only after its steps are done (old and new values
    // are never alive together: no register copies at the loop's end, a copy would wait for the load it copies), and at
    // the loop's head -- where the compiler's wait-count bookkeeping drains everything, loads and stores pending together
    // count as out of order for it -- the youngest load is a whole set of steps old: the drain finds it landed.
    constexpr int kSet = 8;
    struct Slot { IO x[2]; };
    auto fetch_set = [&](Slot (&S)[kSet], int k0) {
#pragma unroll
      for (int u = 0; u < kSet; u++) {
        const IO* row = lp + (int64_t)min(k0 + u, T - 1) * p.sT;
        S[u].x[0] = row[off[0]]; S[u].x[1] = row[off[1]];
      }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto step = [&](int k, const Slot& q) {
      const bool valid = k < T;
      const double x0 = (double)q.x[0], x1 = (double)q.x[1];
      const double* prev = (k & 1) ? row0 : row1;
      double* cur = (k & 1) ? row1 : row0;
      const int start = p.is_ctc ? max(0, L - 2 * (T - k)) : max(0, L - (T - k));
      const int end = p.is_ctc ? min(2 * k + 2, L) : min(k + 1, L);
      unsigned char* bprow = bp + (size_t)k * Lpad;
      const double a00 = prev[jc[0]], a01 = prev[jm1[0]], a02 = prev[jm2[0]];
      const double a10 = prev[jc[1]], a11 = prev[jm1[1]], a12 = prev[jm2[1]];
      auto cell = [&](int c, double s0, double s1, double s2, double x) {
        const int i = tid + c * kThreads;
        double a = s0; int code = 0;
        const bool c1 = allow1[c] && s1 > a;
        a = c1 ? s1 : a; code = c1 ? 1 : code;
        const bool c2 = allow2[c] && s2 > a;
        a = c2 ? s2 : a; code = c2 ? 2 : code;
        a += x;
        const bool in_band = i >= start && i < end;
        a = in_band ? a : ninf(); code = in_band ? code : 3;   // outside the band: path_alpha stays 0
        if (live[c] && valid) { cur[i] = a; bprow[i] = (unsigned char)code; }
      };
      cell(0, a00, a01, a02, x0);
      cell(1, a10, a11, a12, x1);
      lds_barrier();
    };
    // (the refill of the set that has just been used up is issued behind the FIRST step of the next set: whatever that
    //  step's wait is, it then only covers loads that are a set of steps old)
    auto steps = [&](int k0, const Slot (&S)[kSet], Slot (&R)[kSet], int r0) {
      step(k0, S[0]);
      fetch_set(R, r0);
#pragma unroll
      for (int u = 1; u < kSet; u++) step(k0 + u, S[u]);
      asm volatile("" ::: "memory");
    };
    Slot X[kSet], Y[kSet], Z[kSet];
    fetch_set(X, 1); fetch_set(Y, 1 + kSet); fetch_set(Z, 1 + 2 * kSet);
    for (int k = 1; k < T; k += 3 * kSet) {
      steps(k, X, Z, k + 2 * kSet);                 // (Z: refilled with what it holds in the first trip; X's successor otherwise)
      steps(k + kSet, Y, X, k + 3 * kSet);
      steps(k + 2 * kSet, Z, Y, k + 4 * kSet);
    }
  } else
  for (int k = 1; k < T; k++) {
    const double* prev = (k & 1) ? row0 : row1;
    double* cur = (k & 1) ? row1 : row0;
    const int start = p.is_ctc ? max(0, L - 2 * (T - k)) : max(0, L - (T - k));
    const int end = p.is_ctc ? min(2 * k + 2, L) : min(k + 1, L);
    unsigned char* bprow = bp + (size_t)k * Lpad;
    for (int i = tid; i < L; i += kThreads) {
      double a = ninf();
      unsigned char code = 3;                               // outside the band: path_alpha stays 0
      if (i >= start && i < end) {
        const int lab = ext[i];
        a = prev[i];
        code = 0;
        if (i > 0) {
          if (prev[i - 1] > a) { a = prev[i - 1]; code = 1; }
          if (p.is_ctc && lab != p.blank && i - 2 > 0 && ext[i - 2] != lab && prev[i - 2] > a) { a = prev[i - 2]; code = 2; }
        }
        a += LP(k, lab);
      }
      cur[i] = a;
      bprow[i] = code;
    }
    __syncthreads();
  }
  const double* last = ((T - 1) & 1) ? row1 : row0;
  if (tid == 0) {
    int i = L - 1;
    if (p.is_ctc && last[i - 1] > last[i]) i = i - 1;       // :98-100
    s_cell = i;
  }
  // the codes were written by other threads of this workgroup: make them visible before they are staged
  __threadfence_block();
  __syncthreads();

  // ---- back-trace, a chunk of steps at a time out of LDS ----
  const int rows = max(1, min(kChunkBytes / Lpad, kChunkBytes / 64));
  for (int k1 = T; k1 > 0; k1 -= rows) {
    const int k0 = max(k1 - rows, 0), n = k1 - k0;
    {
      // rows of codes are Lpad bytes apart in the workspace and in LDS alike: one flat copy in 16-byte pieces
      const uint4* src = reinterpret_cast<const uint4*>(bp + (size_t)k0 * Lpad);
      uint4* dst = reinterpret_cast<uint4*>(codes);
      for (int q = tid; q < n * (Lpad >> 4); q += kThreads) dst[q] = src[q];
    }
    __syncthreads();
    if (k0 == 0) for (int i = tid; i < L; i += kThreads) codes[i] = 3;       // (step 0: every cell points at 0)
    __syncthreads();
    if (tid == 0) {
      // the walk: one dependent LDS read per step (the cell's code); the cells are noted and turned into labels below
      int i = s_cell;
      int* cells = reinterpret_cast<int*>(bestbuf);
      int r = n - 1;
      const unsigned char* rowp = codes + r * Lpad;
      // (four steps per trip: the only dependent LDS access of a step is the read of the cell's code; the cells are
      //  written out in fours, behind the reads)
      for (; r >= 3; r -= 4, rowp -= 4 * Lpad) {
        const int i0 = i;
        const int c0 = rowp[i0];
        const int i1 = c0 == 3 ? 0 : i0 - c0;
        const int c1 = rowp[i1 - Lpad];
        const int i2 = c1 == 3 ? 0 : i1 - c1;
        const int c2 = rowp[i2 - 2 * Lpad];
        const int i3 = c2 == 3 ? 0 : i2 - c2;
        const int c3 = rowp[i3 - 3 * Lpad];
        i = c3 == 3 ? 0 : i3 - c3;
        cells[2 * r] = i0; cells[2 * (r - 1)] = i1; cells[2 * (r - 2)] = i2; cells[2 * (r - 3)] = i3;
      }
      for (; r >= 0; r--) {
        cells[2 * r] = i;
        const int c = codes[r * Lpad + i];
        i = c == 3 ? 0 : i - c;
      }
      s_cell = i;
    }
    __syncthreads();
    for (int r = tid; r < n; r += kThreads) out[k0 + r] = ext[reinterpret_cast<const int*>(bestbuf)[2 * r]];
    __syncthreads();
  }
}

size_t align_lds_bytes(int Lmax) {
  return sizeof(double) * 2 * (size_t)Lmax + sizeof(int) * (size_t)((Lmax + 3) & ~3) + sizeof(int64_t) * (kChunkBytes / 64) +
         kChunkBytes + 64;
}

}  // namespace
}  // namespace e2e

using namespace e2e;

extern "C" size_t e2e_ctc_align_workspace_bytes(int B, int T, int V, int Smax, int is_ctc) {
  (void)V;
  if (B < 0 || T < 1 || Smax < 0) return 0;
  const size_t Lmax = is_ctc ? 2 * (size_t)Smax + 1 : (size_t)(Smax > 0 ? Smax : 1);
  return align_up((size_t)B * (size_t)T * ((Lmax + 15) & ~(size_t)15), 256) + 256;
}

extern "C" int e2e_ctc_align(const void* lp, int dtype, int64_t sB, int64_t sT, int64_t sV,
                             const int64_t* targets, int64_t tgt_stride,
                             const int64_t* x_len, const int64_t* t_len,
                             int B, int T, int V, int Smax, int blank, int is_ctc,
                             int64_t* out, int64_t pad_value,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64 && !dtype_is_16bit(dtype)) { set_error("dtype must be E2E_F32, E2E_F64, E2E_F16 or E2E_BF16"); return E2E_ERR_ARG; }
  if (B < 0 || T < 1 || V < 1 || Smax < 0) { set_error("bad sizes B=%d T=%d V=%d Smax=%d", B, T, V, Smax); return E2E_ERR_ARG; }
  if (blank < 0 || blank >= V) { set_error("blank=%d outside [0,%d)", blank, V); return E2E_ERR_ARG; }
  if (B > 0 && (!lp || !x_len || !t_len || !out || (Smax > 0 && !targets))) { set_error("null pointer argument"); return E2E_ERR_ARG; }
  const int Lmax = is_ctc ? 2 * Smax + 1 : (Smax > 0 ? Smax : 1);
  const size_t lds = align_lds_bytes(Lmax);
  if (lds > 160 * 1024) { set_error("forced alignment: Smax=%d needs %zu B of LDS (> 160 KiB)", Smax, lds); return E2E_ERR_UNSUPPORTED; }
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const uintptr_t aligned = (base + 255) & ~(uintptr_t)255;
  const int Lpad = (Lmax + 15) & ~15;
  const size_t need = align_up((size_t)B * (size_t)T * (size_t)Lpad, 256);
  if (!workspace || workspace_bytes < need + (aligned - base)) { set_error("workspace too small: need %zu", need + 256); return E2E_ERR_WORKSPACE; }
  if (B == 0) return E2E_OK;
  AlignParams p;
  p.lp = lp; p.sB = sB; p.sT = sT; p.sV = sV; p.targets = targets; p.tgt_stride = tgt_stride;
  p.x_len = x_len; p.t_len = t_len; p.B = B; p.T = T; p.V = V; p.Smax = Smax; p.Lmax = Lmax; p.Lpad = Lpad; p.blank = blank;
  p.is_ctc = is_ctc ? 1 : 0; p.out = out; p.pad = pad_value; p.bp = reinterpret_cast<unsigned char*>(aligned);
  hipStream_t s = (hipStream_t)stream;
  // (16-bit log-probabilities are read as they are: the sweep's rows are doubles either way)
#define E2E_ALIGN_GO(IO) { E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_align_kernel<IO>), (int)lds), "hipFuncSetAttribute"); \
                           hipLaunchKernelGGL(ctc_align_kernel<IO>, dim3(B), dim3(kThreads), lds, s, p); }
  if (dtype == E2E_F32) E2E_ALIGN_GO(float) else if (dtype == E2E_F64) E2E_ALIGN_GO(double) else if (dtype == E2E_F16) E2E_ALIGN_GO(f16_t) else E2E_ALIGN_GO(bf16_t)
#undef E2E_ALIGN_GO
  E2E_HIP_CHECK(hipGetLastError(), "ctc_align_kernel launch");
  return E2E_OK;
}

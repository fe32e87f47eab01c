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
  int B, T, V, Smax, Lmax, blank, is_ctc;
  int64_t* out; int64_t pad;
  unsigned char* bp;       // [B][T][Lmax]
};

__device__ __forceinline__ double ninf() { return -__builtin_huge_val(); }

template <typename IO>
__global__ __launch_bounds__(kThreads) void ctc_align_kernel(AlignParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Tmax = p.T, V = p.V, Lmax = p.Lmax;
  double* row0 = reinterpret_cast<double*>(smem);            // [Lmax]
  double* row1 = row0 + Lmax;                                // [Lmax]
  int* ext = reinterpret_cast<int*>(row1 + Lmax);            // [Lmax] label of every cell
  int64_t* bestbuf = reinterpret_cast<int64_t*>(ext + ((Lmax + 1) & ~1));     // [chunk rows]
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
  unsigned char* bp = p.bp + (size_t)b * (size_t)Tmax * (size_t)Lmax;
  auto LP = [&](int t, int v) -> double { return (double)lp[(int64_t)t * p.sT + (int64_t)v * p.sV]; };

  // ---- forward sweep ----
  for (int i = tid; i < L; i += kThreads) {
    double a = ninf();
    if (i == 0) a = LP(0, ext[0]);
    if (i == 1 && p.is_ctc) a = LP(0, ext[1]);
    row0[i] = a;
  }
  __syncthreads();
  for (int k = 1; k < T; k++) {
    const double* prev = (k & 1) ? row0 : row1;
    double* cur = (k & 1) ? row1 : row0;
    const int start = p.is_ctc ? max(0, L - 2 * (T - k)) : max(0, L - (T - k));
    const int end = p.is_ctc ? min(2 * k + 2, L) : min(k + 1, L);
    unsigned char* bprow = bp + (size_t)k * Lmax;
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
  const int rows = max(1, min(kChunkBytes / L, kChunkBytes / 64));
  for (int k1 = T; k1 > 0; k1 -= rows) {
    const int k0 = max(k1 - rows, 0), n = k1 - k0;
    for (int q = tid; q < n * L; q += kThreads) {
      const int r = q / L, i = q - r * L;
      codes[q] = (k0 + r) > 0 ? bp[(size_t)(k0 + r) * Lmax + i] : 3;          // (step 0: every cell points at 0)
    }
    __syncthreads();
    if (tid == 0) {
      int i = s_cell;
      for (int r = n - 1; r >= 0; r--) {
        bestbuf[r] = ext[i];
        const int c = codes[r * L + i];
        i = c == 3 ? 0 : i - c;
      }
      s_cell = i;
    }
    __syncthreads();
    for (int r = tid; r < n; r += kThreads) out[k0 + r] = bestbuf[r];
    __syncthreads();
  }
}

size_t align_lds_bytes(int Lmax) {
  return sizeof(double) * 2 * (size_t)Lmax + sizeof(int) * (size_t)((Lmax + 1) & ~1) + sizeof(int64_t) * (kChunkBytes / 64) +
         kChunkBytes + 64;
}

}  // namespace
}  // namespace e2e

using namespace e2e;

extern "C" size_t e2e_ctc_align_workspace_bytes(int B, int T, int V, int Smax, int is_ctc) {
  (void)V;
  if (B < 0 || T < 1 || Smax < 0) return 0;
  const size_t Lmax = is_ctc ? 2 * (size_t)Smax + 1 : (size_t)(Smax > 0 ? Smax : 1);
  return align_up((size_t)B * (size_t)T * Lmax, 256) + 256;
}

extern "C" int e2e_ctc_align(const void* lp, int dtype, int64_t sB, int64_t sT, int64_t sV,
                             const int64_t* targets, int64_t tgt_stride,
                             const int64_t* x_len, const int64_t* t_len,
                             int B, int T, int V, int Smax, int blank, int is_ctc,
                             int64_t* out, int64_t pad_value,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64) { set_error("dtype must be E2E_F32 or E2E_F64"); return E2E_ERR_ARG; }
  if (B < 0 || T < 1 || V < 1 || Smax < 0) { set_error("bad sizes B=%d T=%d V=%d Smax=%d", B, T, V, Smax); return E2E_ERR_ARG; }
  if (blank < 0 || blank >= V) { set_error("blank=%d outside [0,%d)", blank, V); return E2E_ERR_ARG; }
  if (B > 0 && (!lp || !x_len || !t_len || !out || (Smax > 0 && !targets))) { set_error("null pointer argument"); return E2E_ERR_ARG; }
  const int Lmax = is_ctc ? 2 * Smax + 1 : (Smax > 0 ? Smax : 1);
  const size_t lds = align_lds_bytes(Lmax);
  if (lds > 160 * 1024) { set_error("forced alignment: Smax=%d needs %zu B of LDS (> 160 KiB)", Smax, lds); return E2E_ERR_UNSUPPORTED; }
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const uintptr_t aligned = (base + 255) & ~(uintptr_t)255;
  const size_t need = align_up((size_t)B * (size_t)T * (size_t)Lmax, 256);
  if (!workspace || workspace_bytes < need + (aligned - base)) { set_error("workspace too small: need %zu", need + 256); return E2E_ERR_WORKSPACE; }
  if (B == 0) return E2E_OK;
  AlignParams p;
  p.lp = lp; p.sB = sB; p.sT = sT; p.sV = sV; p.targets = targets; p.tgt_stride = tgt_stride;
  p.x_len = x_len; p.t_len = t_len; p.B = B; p.T = T; p.V = V; p.Smax = Smax; p.Lmax = Lmax; p.blank = blank;
  p.is_ctc = is_ctc ? 1 : 0; p.out = out; p.pad = pad_value; p.bp = reinterpret_cast<unsigned char*>(aligned);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == E2E_F32) {
    E2E_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_align_kernel<float>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    hipLaunchKernelGGL(ctc_align_kernel<float>, dim3(B), dim3(kThreads), lds, s, p);
  } else {
    E2E_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_align_kernel<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    hipLaunchKernelGGL(ctc_align_kernel<double>, dim3(B), dim3(kThreads), lds, s, p);
  }
  E2E_HIP_CHECK(hipGetLastError(), "ctc_align_kernel launch");
  return E2E_OK;
}

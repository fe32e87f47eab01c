// extern "C" entry points of libe2e_ctc.so (see include/e2e_ctc.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#include "common.h"

namespace e2e {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return E2E_ERR_HIP;
}

hipError_t allow_dynamic_lds(const void* fn, int bytes) {
  // the largest size asked for so far per (kernel, device); the attribute is raised only when a launch needs more
  struct Seen { const void* fn; int dev; int bytes; };
  static std::mutex mu;
  static std::vector<Seen> seen;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& s : seen)
    if (s.fn == fn && s.dev == dev) {
      if (s.bytes >= bytes) return hipSuccess;
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) s.bytes = bytes;
      return e;
    }
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) seen.push_back(Seen{fn, dev, bytes});
  return e;
}

int launch_greedy(const void* x, int dtype, int64_t sB, int64_t sT, int64_t sV, const int64_t* x_len,
                  int B, int T, int V, int blank, int64_t* out, int64_t* out_len, hipStream_t stream);
int launch_scale(void* grads, int dtype, const void* scale, int B, int64_t row_elems, hipStream_t stream);

namespace {
template <typename IO>
__global__ void scale_rows_kernel(IO* g, const IO* s, int64_t row_elems) {
  const IO f = s[blockIdx.y];
  if (f == (IO)1) return;                  // nothing to do for this row: no pass over it
  IO* row = g + (size_t)blockIdx.y * (size_t)row_elems;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_elems; i += (int64_t)gridDim.x * blockDim.x)
    row[i] = (IO)(row[i] * f);             // (16-bit: the product in the source dtype's precision, as torch multiplies it)
}

// sum / mean of the B losses in a fixed order (deterministic), f64 accumulation
template <typename IO>
__global__ __launch_bounds__(256) void reduce_losses_kernel(const IO* losses, int B, int mean, IO* out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) s += (double)losses[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (IO)(mean ? part[0] / (double)B : part[0]);
}
}  // namespace

int launch_reduce_losses(const LossArgs& a) {
  if (!a.reduced || a.reduction == E2E_REDUCE_NONE || a.B == 0) return E2E_OK;
  const int mean = a.reduction == E2E_REDUCE_MEAN;
  if (a.dtype == E2E_F32 || dtype_is_16bit(a.dtype))          // (16-bit I/O keeps losses and their reduction in f32)
    hipLaunchKernelGGL(reduce_losses_kernel<float>, dim3(1), dim3(256), 0, a.stream, (const float*)a.losses, a.B, mean, (float*)a.reduced);
  else
    hipLaunchKernelGGL(reduce_losses_kernel<double>, dim3(1), dim3(256), 0, a.stream, (const double*)a.losses, a.B, mean, (double*)a.reduced);
  E2E_HIP_CHECK(hipGetLastError(), "reduce_losses_kernel launch");
  return E2E_OK;
}

int launch_scale(void* grads, int dtype, const void* scale, int B, int64_t row_elems, hipStream_t stream) {
  if (B == 0 || row_elems == 0) return E2E_OK;
  // enough blocks to fill the chip whatever the split into rows is (B = 1: one scalar for the whole tensor)
  int64_t want = (row_elems + 1023) / 1024;
  // (B = 1 -- the scalar grad_output of a reduced loss, which is 1 whenever the caller wrote loss.backward() -- : the launch usually ends
  //  in every workgroup's first test, and 1024 of them end sooner than 4096; a real pass over 30 MB loses nothing with 256 K threads)
  const int64_t cap = B >= 64 ? 64 : B == 1 ? 1024 : 4096 / B;
  if (want > cap) want = cap;
  const int gx = (int)(want < 1 ? 1 : want);
  if (dtype == E2E_F32)
    hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(gx, B), dim3(256), 0, stream,
                       (float*)grads, (const float*)scale, row_elems);
  else if (dtype == E2E_F16)
    hipLaunchKernelGGL(scale_rows_kernel<f16_t>, dim3(gx, B), dim3(256), 0, stream,
                       (f16_t*)grads, (const f16_t*)scale, row_elems);
  else if (dtype == E2E_BF16)
    hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, dim3(gx, B), dim3(256), 0, stream,
                       (bf16_t*)grads, (const bf16_t*)scale, row_elems);
  else
    hipLaunchKernelGGL(scale_rows_kernel<double>, dim3(gx, B), dim3(256), 0, stream,
                       (double*)grads, (const double*)scale, row_elems);
  E2E_HIP_CHECK(hipGetLastError(), "scale_rows_kernel launch");
  return E2E_OK;
}

// Streaming copy used by bench.py to measure the box's achievable HBM rate (roofline.peak_measured).  Every wave copies
// contiguous 32 KB pieces, eight 16-byte non-temporal loads per lane in flight, then their stores: 6.5 - 7.0 TB/s on the
// pool's boxes (tools/diag/microbench/ubench_copy.hip).  The grid-stride form this kernel had until round 4 -- a wave's eight
// loads a whole grid apart -- reaches 4.8 - 5.1 TB/s on the same boxes, and hipMemcpyAsync 4.7 - 4.9: `peak_measured` was
// not a ceiling.
typedef float copy_f4 __attribute__((ext_vector_type(4)));
constexpr int kCopyPiece = 2048;                       // 16-byte elements per piece (32 KB)
__global__ __launch_bounds__(256) void stream_copy_kernel(copy_f4* __restrict__ dst, const copy_f4* __restrict__ src, size_t n16) {
  const int lane = threadIdx.x & 63;
  const size_t nwaves = (size_t)gridDim.x * 4, npieces = n16 / kCopyPiece;
  for (size_t pc = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); pc < npieces; pc += nwaves) {
    const copy_f4* s = src + pc * kCopyPiece; copy_f4* d = dst + pc * kCopyPiece;
    for (int i = lane; i < kCopyPiece; i += 8 * 64) {
      copy_f4 v[8];
#pragma unroll
#ifndef E2E_COPY_NT_LOADS
#define E2E_COPY_NT_LOADS 1
#endif
      for (int u = 0; u < 8; u++) v[u] = E2E_COPY_NT_LOADS ? __builtin_nontemporal_load(&s[i + 64 * u]) : s[i + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; u++) __builtin_nontemporal_store(v[u], &d[i + 64 * u]);
    }
  }
  // (what does not fill a piece)
  for (size_t i = npieces * kCopyPiece + (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// Holds `gridDim.x` workgroups on the chip for `ticks` of the 100 MHz wall clock (tests: the flagged launch's bounded waits
// with another stream's kernels resident; the dynamic LDS is only there to be taken).
__global__ void occupy_kernel(long long ticks, int* sink) {
  extern __shared__ int occ_lds[];
  const unsigned long long t0 = wall_clock64();
  int spins = 0;
  while ((long long)(wall_clock64() - t0) < ticks && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(16);
  if (sink && spins < 0) sink[0] = occ_lds[threadIdx.x];
}

}  // namespace e2e

using namespace e2e;

extern "C" int e2e_debug_occupy(int workgroups, int threads, int lds_bytes, long long nanoseconds, void* stream) {
  if (workgroups < 1 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || nanoseconds < 0 || nanoseconds > 100000000) {
    set_error("e2e_debug_occupy: bad argument"); return E2E_ERR_ARG;
  }
  E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&occupy_kernel), lds_bytes), "hipFuncSetAttribute");
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(threads), lds_bytes, (hipStream_t)stream, nanoseconds / 10, (int*)nullptr);
  E2E_HIP_CHECK(hipGetLastError(), "occupy_kernel launch");
  return E2E_OK;
}

// diagnostics, not part of include/e2e_ctc.h: dst[0..bytes) = src[0..bytes), bytes a multiple of 16
extern "C" int e2e_debug_stream_copy(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src || bytes % 16) { set_error("e2e_debug_stream_copy: bad argument"); return E2E_ERR_ARG; }
  hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream,
                     (copy_f4*)dst, (const copy_f4*)src, bytes / 16);
  E2E_HIP_CHECK(hipGetLastError(), "stream_copy_kernel launch");
  return E2E_OK;
}

extern "C" {

int e2e_ctc_abi_version(void) { return E2E_CTC_ABI_VERSION; }
const char* e2e_last_error(void) { return g_err; }

static bool use_wide(int dtype, int T, int V, int Smax) {
  return !fast_supported(T, V, Smax, dtype) && wide_supported(T, V, Smax, dtype);
}

static int resolve_algo(int algo, int dtype, int T, int V, int Smax) {
  if (algo == E2E_ALGO_EXACT) return E2E_ALGO_EXACT;
  if (algo == E2E_ALGO_FAST) return E2E_ALGO_FAST;
  return ((dtype == E2E_F32 || dtype_is_16bit(dtype)) && (fast_supported(T, V, Smax, dtype) || wide_supported(T, V, Smax, dtype))) ? E2E_ALGO_AUTO : E2E_ALGO_EXACT;
}

size_t e2e_ctc_loss_workspace_bytes(int B, int T, int V, int Smax, int dtype, int algo) {
  if (B < 0 || T < 1 || V < 1 || Smax < 0) return 0;
  const int r = resolve_algo(algo, dtype, T, V, Smax);
  size_t n = 0;
  if (r != E2E_ALGO_EXACT && use_wide(dtype, T, V, Smax)) return wide_workspace_bytes(B, T, V, Smax, r == E2E_ALGO_AUTO) + 256;
  if (r == E2E_ALGO_EXACT) n += exact_workspace_bytes(B, T, V, Smax);
  if (r == E2E_ALGO_AUTO) n += exact_fallback_workspace_bytes(B, T, V, Smax);
  if (r == E2E_ALGO_FAST || r == E2E_ALGO_AUTO) n += fast_workspace_bytes(B, T, V, Smax);
  return n + 256;
}

// Does a loss call with these logits run as it is (1), or does the caller have to up-cast 16-bit logits to f32 first (0)?
// Mirrors the dispatch of e2e_ctc_loss_fwd_bwd_opt below and launch_wide's single-read condition (ctc_loss_wide.hip).
int e2e_ctc_loss_takes_dtype(int dtype, int algo, int T, int V, int Smax, int64_t sB, int64_t sT, int64_t sV,
                             const void* x, const void* grads) {
  if (dtype == E2E_F32 || dtype == E2E_F64) return 1;
  if (!dtype_is_16bit(dtype) || T < 1 || V < 1 || Smax < 0) return 0;
  if (resolve_algo(algo, dtype, T, V, Smax) == E2E_ALGO_EXACT) return 0;
  // (the lattice kernels read any stride, any alignment; so does the wide path since round 6 -- rows it cannot hold in registers, more
  //  than 8192 or unaligned columns, take its two-pass form element by element instead of an up-cast copy on the host)
  (void)sB; (void)sT; (void)sV; (void)x; (void)grads;
  return 1;
}

int e2e_ctc_loss_fwd_bwd(const void* x, int dtype, int input_is_logprobs,
                         int64_t sB, int64_t sT, int64_t sV,
                         const int64_t* targets, int64_t tgt_stride,
                         const int64_t* x_len, const int64_t* t_len,
                         int B, int T, int V, int Smax, int blank,
                         void* losses, void* grads,
                         void* workspace, size_t workspace_bytes,
                         int algo, void* stream) {
  return e2e_ctc_loss_fwd_bwd_opt(x, dtype, input_is_logprobs, sB, sT, sV, targets, tgt_stride, x_len, t_len, B, T, V,
                                  Smax, blank, losses, grads, workspace, workspace_bytes, algo, stream, nullptr);
}

int e2e_ctc_loss_fwd_bwd_opt(const void* x, int dtype, int input_is_logprobs,
                             int64_t sB, int64_t sT, int64_t sV,
                             const int64_t* targets, int64_t tgt_stride,
                             const int64_t* x_len, const int64_t* t_len,
                             int B, int T, int V, int Smax, int blank,
                             void* losses, void* grads,
                             void* workspace, size_t workspace_bytes,
                             int algo, void* stream, const e2e_ctc_loss_opts* opts) {
  if (opts && (opts->reduction < E2E_REDUCE_NONE || opts->reduction > E2E_REDUCE_MEAN ||
               (opts->reduction != E2E_REDUCE_NONE && !opts->reduced))) {
    set_error("bad e2e_ctc_loss_opts: reduction %d, reduced %p", opts->reduction, opts->reduced); return E2E_ERR_ARG;
  }
  if (opts && opts->chains != E2E_CHAINS_F64 && opts->chains != E2E_CHAINS_F32) {
    set_error("bad e2e_ctc_loss_opts: chains %d", opts->chains); return E2E_ERR_ARG;
  }
  if (dtype != E2E_F32 && dtype != E2E_F64 && !dtype_is_16bit(dtype)) { set_error("dtype must be E2E_F32, E2E_F64, E2E_F16 or E2E_BF16"); return E2E_ERR_ARG; }
  if (B < 0 || T < 1 || V < 1 || Smax < 0) { set_error("bad sizes B=%d T=%d V=%d Smax=%d", B, T, V, Smax); return E2E_ERR_ARG; }
  if (blank < 0 || blank >= V) { set_error("blank=%d outside [0,%d)", blank, V); return E2E_ERR_ARG; }
  if (B > 0 && (!x || !x_len || !t_len || !losses || !grads || (Smax > 0 && !targets))) {
    set_error("null pointer argument"); return E2E_ERR_ARG;
  }
  if (algo != E2E_ALGO_AUTO && algo != E2E_ALGO_EXACT && algo != E2E_ALGO_FAST) { set_error("bad algo %d", algo); return E2E_ERR_ARG; }
  // 256-B align the workspace base
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const uintptr_t aligned = (base + 255) & ~(uintptr_t)255;
  if (workspace && workspace_bytes >= (aligned - base)) { workspace_bytes -= (aligned - base); workspace = reinterpret_cast<void*>(aligned); }
  LossArgs a{x, dtype, input_is_logprobs ? 1 : 0, sB, sT, sV, targets, tgt_stride, x_len, t_len,
             B, T, V, Smax, blank, losses, grads, workspace, workspace_bytes, (hipStream_t)stream};
  if (opts) { a.grad_scale = opts->grad_scale; a.reduced = opts->reduced; a.reduction = opts->reduction; a.chains = opts->chains; }
  // AUTO with f32 I/O: wherever the exact kernel stands in for (or finishes) an f32 path it may use its scaled f64 form;
  // E2E_ALGO_EXACT and f64 always run the reference's log-domain arithmetic.  (E2E_EXACT_LOGDOMAIN=1: everywhere.)
  static const bool logdomain_only = [] { const char* e = getenv("E2E_EXACT_LOGDOMAIN"); return e && e[0] == '1'; }();
  a.scaled_exact = (algo == E2E_ALGO_AUTO && (dtype == E2E_F32 || dtype_is_16bit(dtype)) && !logdomain_only) ? 1 : 0;
  const int r = resolve_algo(algo, dtype, T, V, Smax);
  if (dtype_is_16bit(dtype) && r == E2E_ALGO_EXACT) {
    set_error("16-bit logits are taken by the fast and wide paths only (algo AUTO / FAST, shapes they support): up-cast to f32");
    return E2E_ERR_UNSUPPORTED;
  }
  if (r == E2E_ALGO_EXACT) { const int rc = launch_exact(a); return rc != E2E_OK ? rc : launch_reduce_losses(a); }
  if (use_wide(dtype, T, V, Smax)) {
    if (r == E2E_ALGO_FAST && !wide_takes_fast_lattice(T, V, Smax, dtype)) {
      set_error("fast CTC path does not support this shape/dtype (Smax + 1 = %d > 448 compact columns, or T >= 2^22: the compact lattice is the exact kernel's)", Smax + 1);
      return E2E_ERR_UNSUPPORTED;
    }
    const int rc = launch_wide(a, r == E2E_ALGO_AUTO);
    return rc != E2E_OK ? rc : launch_reduce_losses(a);
  }
  if (r == E2E_ALGO_FAST) {
    if (!fast_supported(T, V, Smax, dtype)) { set_error("fast CTC path does not support this shape/dtype"); return E2E_ERR_UNSUPPORTED; }
    return launch_fast(a, false);
  }
  return launch_fast(a, true);
}

int e2e_ctc_scale_grads(void* grads, int dtype, const void* scale, int B, int64_t row_elems, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64 && !dtype_is_16bit(dtype)) { set_error("dtype must be E2E_F32, E2E_F64, E2E_F16 or E2E_BF16"); return E2E_ERR_ARG; }
  if (B < 0 || row_elems < 0 || (B > 0 && row_elems > 0 && (!grads || !scale))) { set_error("bad argument"); return E2E_ERR_ARG; }
  return launch_scale(grads, dtype, scale, B, row_elems, (hipStream_t)stream);
}

int e2e_ctc_greedy(const void* x, int dtype, int64_t sB, int64_t sT, int64_t sV,
                   const int64_t* x_len, int B, int T, int V, int blank,
                   int64_t* out, int64_t* out_len, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64 && !dtype_is_16bit(dtype)) { set_error("dtype must be E2E_F32, E2E_F64, E2E_F16 or E2E_BF16"); return E2E_ERR_ARG; }
  if (B < 0 || T < 1 || V < 1) { set_error("bad sizes B=%d T=%d V=%d", B, T, V); return E2E_ERR_ARG; }
  if (B > 0 && (!x || !x_len || !out || !out_len)) { set_error("null pointer argument"); return E2E_ERR_ARG; }
  return launch_greedy(x, dtype, sB, sT, sV, x_len, B, T, V, blank, out, out_len, (hipStream_t)stream);
}

}  // extern "C"

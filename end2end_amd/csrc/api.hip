// extern "C" entry points of libe2e_ctc.so (see include/e2e_ctc.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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

int launch_greedy(const void* x, int dtype, int64_t sB, int64_t sT, int64_t sV, const int64_t* x_len,
                  int B, int T, int V, int blank, int64_t* out, int64_t* out_len, hipStream_t stream);
int launch_scale(void* grads, int dtype, const void* scale, int B, int64_t row_elems, hipStream_t stream);

namespace {
template <typename IO>
__global__ void scale_rows_kernel(IO* g, const IO* s, int64_t row_elems) {
  const IO f = s[blockIdx.y];
  IO* row = g + (size_t)blockIdx.y * (size_t)row_elems;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_elems; i += (int64_t)gridDim.x * blockDim.x)
    row[i] *= f;
}
}  // namespace

int launch_scale(void* grads, int dtype, const void* scale, int B, int64_t row_elems, hipStream_t stream) {
  if (B == 0 || row_elems == 0) return E2E_OK;
  int gx = (int)((row_elems + 255) / 256); if (gx > 64) gx = 64;
  if (dtype == E2E_F32)
    hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(gx, B), dim3(256), 0, stream,
                       (float*)grads, (const float*)scale, row_elems);
  else
    hipLaunchKernelGGL(scale_rows_kernel<double>, dim3(gx, B), dim3(256), 0, stream,
                       (double*)grads, (const double*)scale, row_elems);
  E2E_HIP_CHECK(hipGetLastError(), "scale_rows_kernel launch");
  return E2E_OK;
}

}  // namespace e2e

using namespace e2e;

extern "C" {

int e2e_ctc_abi_version(void) { return E2E_CTC_ABI_VERSION; }
const char* e2e_last_error(void) { return g_err; }

static bool use_wide(int dtype, int T, int V, int Smax) {
  return !fast_supported(T, V, Smax, dtype) && wide_supported(T, V, Smax, dtype);
}

static int resolve_algo(int algo, int dtype, int T, int V, int Smax) {
  if (algo == E2E_ALGO_EXACT) return E2E_ALGO_EXACT;
  if (algo == E2E_ALGO_FAST) return E2E_ALGO_FAST;
  return (dtype == E2E_F32 && (fast_supported(T, V, Smax, dtype) || wide_supported(T, V, Smax, dtype))) ? E2E_ALGO_AUTO : E2E_ALGO_EXACT;
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

int e2e_ctc_loss_fwd_bwd(const void* x, int dtype, int input_is_logprobs,
                         int64_t sB, int64_t sT, int64_t sV,
                         const int64_t* targets, int64_t tgt_stride,
                         const int64_t* x_len, const int64_t* t_len,
                         int B, int T, int V, int Smax, int blank,
                         void* losses, void* grads,
                         void* workspace, size_t workspace_bytes,
                         int algo, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64) { set_error("dtype must be E2E_F32 or E2E_F64"); return E2E_ERR_ARG; }
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
  const int r = resolve_algo(algo, dtype, T, V, Smax);
  if (r == E2E_ALGO_EXACT) return launch_exact(a);
  if (use_wide(dtype, T, V, Smax)) return launch_wide(a, r == E2E_ALGO_AUTO);
  if (r == E2E_ALGO_FAST) {
    if (!fast_supported(T, V, Smax, dtype)) { set_error("fast CTC path does not support this shape/dtype"); return E2E_ERR_UNSUPPORTED; }
    return launch_fast(a, false);
  }
  return launch_fast(a, true);
}

int e2e_ctc_scale_grads(void* grads, int dtype, const void* scale, int B, int64_t row_elems, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64) { set_error("dtype must be E2E_F32 or E2E_F64"); return E2E_ERR_ARG; }
  if (B < 0 || row_elems < 0 || (B > 0 && row_elems > 0 && (!grads || !scale))) { set_error("bad argument"); return E2E_ERR_ARG; }
  return launch_scale(grads, dtype, scale, B, row_elems, (hipStream_t)stream);
}

int e2e_ctc_greedy(const void* x, int dtype, int64_t sB, int64_t sT, int64_t sV,
                   const int64_t* x_len, int B, int T, int V, int blank,
                   int64_t* out, int64_t* out_len, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64) { set_error("dtype must be E2E_F32 or E2E_F64"); return E2E_ERR_ARG; }
  if (B < 0 || T < 1 || V < 1) { set_error("bad sizes B=%d T=%d V=%d", B, T, V); return E2E_ERR_ARG; }
  if (B > 0 && (!x || !x_len || !out || !out_len)) { set_error("null pointer argument"); return E2E_ERR_ARG; }
  return launch_greedy(x, dtype, sB, sT, sV, x_len, B, T, V, blank, out, out_len, (hipStream_t)stream);
}

}  // extern "C"

// Shared helpers for the MI355X (gfx950) CTC library.  Wave size is 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#include "../../include/e2e_ctc.h"

namespace e2e {

constexpr int kWave = 64;

// thread-local error text behind e2e_last_error()
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// -- launchers implemented in the .hip files ---------------------------------
struct LossArgs {
  const void* x; int dtype; int logprobs;
  int64_t sB, sT, sV;
  const int64_t* targets; int64_t tgt_stride;
  const int64_t* x_len; const int64_t* t_len;
  int B, T, V, Smax, blank;
  void* losses; void* grads;
  void* ws; size_t ws_bytes;
  hipStream_t stream;
};

size_t exact_workspace_bytes(int B, int T, int V, int Smax);            // every utterance (algo EXACT)
size_t exact_fallback_workspace_bytes(int B, int T, int V, int Smax);   // flagged-utterance fallback of the fast path
int launch_exact(const LossArgs& a);
size_t fast_workspace_bytes(int B, int T, int V, int Smax);
int launch_fast(const LossArgs& a, bool fallback_to_exact);
bool fast_supported(int T, int V, int Smax, int dtype);
// wide alphabets: per-utterance compaction around the fast path (ctc_loss_wide.hip)
bool wide_supported(int T, int V, int Smax, int dtype);
size_t wide_workspace_bytes(int B, int T, int V, int Smax, bool with_exact);
int launch_wide(const LossArgs& a, bool fallback_to_exact);

}  // namespace e2e

#define E2E_HIP_CHECK(expr, what)                          \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return e2e::hip_fail(_e, what);  \
  } while (0)

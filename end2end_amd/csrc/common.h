// Shared helpers for the MI355X (gfx950) CTC library.  Wave size is 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#include "../../include/e2e_ctc.h"
#include "../../include/e2e_ctc_debug.h"

namespace e2e {

constexpr int kWave = 64;

// 16-bit I/O types of the kernels (E2E_F16 / E2E_BF16): plain arithmetic types of the compiler, converted on load / store
typedef _Float16 f16_t;
typedef __bf16 bf16_t;
inline bool dtype_is_16bit(int dtype) { return dtype == E2E_F16 || dtype == E2E_BF16; }
// losses / reduced of a call: the I/O dtype, except that 16-bit calls keep them in f32 (include/e2e_ctc.h)
template <typename IO> struct LossOf { typedef IO type; };
template <> struct LossOf<f16_t> { typedef float type; };
template <> struct LossOf<bf16_t> { typedef float type; };
// one element of a tensor whose dtype is known at run time only (wave-uniform `dt`): used where a kernel is not worth an
// instance per dtype
__device__ __forceinline__ float load_elem(const void* base, int64_t idx, int dt) {
  if (dt == E2E_F32) return reinterpret_cast<const float*>(base)[idx];
  if (dt == E2E_BF16) return (float)reinterpret_cast<const bf16_t*>(base)[idx];
  return (float)reinterpret_cast<const f16_t*>(base)[idx];
}
__device__ __forceinline__ void store_elem(void* base, size_t idx, float v, int dt) {
  if (dt == E2E_F32) reinterpret_cast<float*>(base)[idx] = v;
  else if (dt == E2E_BF16) reinterpret_cast<bf16_t*>(base)[idx] = (bf16_t)v;
  else reinterpret_cast<f16_t*>(base)[idx] = (f16_t)v;
}

// thread-local error text behind e2e_last_error()
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncAttributeMaxDynamicSharedMemorySize = `bytes` for kernel `fn`, raised when a launch needs more than the largest size asked for so far per (kernel, device) --
// not on every launch: a host call that an entry point meant to be captured into a graph should not repeat.  Returns a hipError_t.
hipError_t allow_dynamic_lds(const void* fn, int bytes);

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
  // e2e_ctc_loss_opts (include/e2e_ctc.h): every gradient element is multiplied by grad_scale as it is written;
  // `reduced` (one element of the I/O dtype, may be null) receives the sum / mean of the B losses
  double grad_scale = 1.0;
  void* reduced = nullptr;
  int reduction = 0;
  int chains = 0;          // E2E_CHAINS_*
  int scaled_exact = 0;    // the exact kernel may use its scaled probability-domain form (f32 I/O, algo AUTO; see ctc_exact_one)
};
// sum / mean of the losses by one small launch (paths that have no tail to fold it into)
int launch_reduce_losses(const LossArgs& a);

// Exponential tilt of the fast path's scaled lattice (rows hold alpha[j] r^j and beta[j] r^(L-1-j)); shared by the
// kernels that produce and that consume its checkpoints.  See LaneCells in ctc_loss_fast.hip.
// frame number -> (utterance, step) for the one-wave-per-frame kernels: a 32-bit division where the frame count allows it
// (a 64-bit one is ~100 instructions, more than some of those kernels' own work per frame)
__device__ __forceinline__ void split_frame(int64_t row, int T, int& b, int& t) {
  if (row < 0x7fffffffLL) { const unsigned r = (unsigned)row, q = r / (unsigned)T; b = (int)q; t = (int)(r - q * (unsigned)T); }
  else { b = (int)(row / T); t = (int)(row - (int64_t)b * T); }
}

__host__ __device__ inline float fast_tilt(int S, int T) {
  float rho = (float)S / (float)T;
  rho = fminf(fmaxf(rho, 1.f / 33.f), 0.8f);
  return S > 0 ? 2.f * rho / (1.f - rho) : 1.f;
}
constexpr int kFastSeg = 16;       // steps between two checkpoints of the fast path

// What the flagged-utterance launch needs to redo only the SECOND kernel of the fast path in f64 (an utterance whose
// f32 segment kernel ran out of range keeps its f64 chains' results: checkpoints, probabilities, loss).
struct FastRetry {
  int* ctl;                        // the fast path's control words (see FastParams::ctl)
  int ytab_segments;               // ytab is [B][NS][V][16] (the small alphabets' form) instead of [B][T][V]
  const float* ytab; const float* ckA; const float* ckQ; const short* ckE; const int* cumA; const int* cumB;
  const double* logz;              // [B][2] the chains' log Z (alpha side, beta side)
  const unsigned* segmask;         // [B][MW] bit s: segment s failed its range / self-check in the segment kernel (only those are redone)
  int NS, NB, CELLS, PPL, MW;
  // the extended-range redo (ctc_ext.h): per-cell exponents of the checkpoint rows it writes over ckA / ckQ for the
  // utterances it takes, and their partition sums
  int* ckXA; int* ckXQ;            // [B][NS][CELLS]
  double* extz;                    // [B][2]  Z = extz[0] * 2^extz[1]; behind them [B][2 sides][4]: the sides' cells of Z when alpha and beta ran on two workgroups
};

size_t exact_workspace_bytes(int B, int T, int V, int Smax);            // every utterance (algo EXACT)
size_t exact_fallback_workspace_bytes(int B, int T, int V, int Smax);   // flagged-utterance fallback of the fast path
int launch_exact(const LossArgs& a);
size_t fast_workspace_bytes(int B, int T, int V, int Smax);
int launch_fast(const LossArgs& a, bool fallback_to_exact);
bool fast_supported(int T, int V, int Smax, int dtype);
// wide alphabets: per-utterance compaction around the fast path (ctc_loss_wide.hip)
bool wide_supported(int T, int V, int Smax, int dtype);
bool wide_takes_fast_lattice(int T, int V, int Smax, int dtype);   // (false: the compact lattice is the exact kernel's)
size_t wide_workspace_bytes(int B, int T, int V, int Smax, bool with_exact);
int launch_wide(const LossArgs& a, bool fallback_to_exact);

}  // namespace e2e

#define E2E_HIP_CHECK(expr, what)                          \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return e2e::hip_fail(_e, what);  \
  } while (0)

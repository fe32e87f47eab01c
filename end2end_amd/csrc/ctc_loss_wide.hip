// Wide alphabets (V > 64, e.g. the V=8000 word-piece / OCR shape): the lattice only ever touches the utterance's
// own labels, so the alphabet is COMPACTED per utterance to its distinct labels + the blank (<= Smax+1 columns) and
// the small-alphabet lattice kernels (ctc_loss_fast.hip, with the exact kernel as their fallback) run unchanged on
// the compact log-probabilities.  Around them the streaming kernels carry all of the HBM traffic:
//   wide_rows_dense_kernel  (rows of <= 8192 aligned columns) one wave per frame holds the WHOLE row in registers
//                      (<= 32 float4 per lane): maximum, exp, sum, then grad[v] = softmax(x)[v] is written straight from
//                      the registers -- the logits are read once and the gradient written once, 2*V*4 bytes per frame,
//                      the algorithmic minimum -- plus the row's log-sum-exp and the <= Smax+1 compact log-probs
//   wide_fix_kernel    after the lattice: the <= Smax+1 label columns of every live frame are overwritten with
//                      (prob - posterior) from the compact gradient (scattered 4-byte writes, ~3 % extra traffic);
//                      utterances that turned out infeasible get their slab poisoned (quirk Q2)
//   wide_rows_kernel + wide_emit_kernel  (any stride / width): the two-pass form, logits read twice (3*V*4 bytes per
//                      frame): online max / sum-exp first, the dense gradient with the label columns afterwards
// Reference semantics: src/losses/ctc_loss.cpp:102-117 (gradient over the full (T,V) slab, quirks Q1/Q2).
#include <type_traits>

#include "common.h"

#pragma clang fp contract(fast)

namespace e2e {
namespace {

#ifndef E2E_WIDE_KWAVES
#define E2E_WIDE_KWAVES 4
#endif
constexpr int kWaves = E2E_WIDE_KWAVES;      // frames per workgroup (round 6, one process, B=512 T=256 V=8000: 1 / 2 / 4 / 8 frames -> f32 1825 / 1761 / 1715 / 1726 us,
                                             //  bf16 982 / 956 / 946 / 1049 us per call)

struct WideParams {
  const float* x; int64_t sB, sT, sV; int logprobs;   // (16-bit logits: the dense kernels' instances reinterpret x / grads)
  const int64_t* targets; int64_t tgt_stride; const int64_t* x_len; const int64_t* t_len;
  int B, T, V, Smax, VC, blank;
  float gscale;         // every gradient element is multiplied by this as it is written
  float* grads; float* losses;
  int64_t* targets_c;   // [B][Smax]  compact id of target i
  int* clabel;          // [B][VC]    original label of compact column k (-1: unused); column VC-1 is the blank
  float* lse;           // [B][T]
  float* shift;         // [B][T]     max of the frame's compact log-probs (<= 0), removed from the compact row
  float* xc;            // [B][T][VC] compact log-probs minus the frame's shift
  const float* gc;      // [B][T][VC] compact gradient from the lattice kernels
};

__device__ __forceinline__ float exp_acc(float x) {        // ~1 ulp, x <= ~88
  x = fmaxf(x, -200.f);
  const float t = x * 1.44269504088896340736f;
  const float n = rintf(t);
  float f = fmaf(x, 1.44269504088896340736f, -n);
  f = fmaf(x, 1.92596299112661746e-8f, f);
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// exp(x) for x <= 0 where 1e-6 relative is plenty (the dense rows' exp(x - max): the result is a probability <= 1 that ends up in
// a gradient held to 2e-6 absolute, or in a 16-bit number): the product's rounding, |x| log2(e) 2^-24, is the whole error.
// Two instructions instead of exp_acc's eight.  What it buys, in one process (tools/diag/ab_time.py, B=512, T=256, V=8000): nothing for f32
// and bf16 rows (1740 / 985 us per call either way: those kernels wait for HBM) and 1535 -> 995 us for f16 rows, whose kernel no longer
// spills (86 registers instead of 168 + 41 spilled).
#ifndef E2E_WIDE_FAST_EXP
#define E2E_WIDE_FAST_EXP 1
#endif
// exp(x - M) with mM = -M log2(e) worked out once per row: one multiply-add and v_exp_f32 (exp2(-inf) = 0: padding lanes need no clamp;
// the rounding of mM is common to the whole row and cancels in the normalisation)
__device__ __forceinline__ float exp_row(float x, float M, float mM) {
#if E2E_WIDE_FAST_EXP
  (void)M;
  return __builtin_amdgcn_exp2f(fmaf(x, 1.44269504088896340736f, mM));
#else
  (void)mM;
  return exp_acc(x - M);
#endif
}

__device__ __forceinline__ float wave_max_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_sum_f(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

// One workgroup per utterance: distinct labels in order of first appearance -> compact ids.
__global__ __launch_bounds__(256) void wide_compact_kernel(WideParams p) {
  extern __shared__ int sh[];
  int* lab = sh;                    // [Smax]
  int* rep = sh + p.Smax;           // [Smax] index of the first occurrence of target i's label
  int* cid = sh + 2 * p.Smax;       // [Smax] compact id (valid at representatives)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t Sq = p.t_len[b];
  const int S = Sq < 0 ? 0 : (Sq > p.Smax ? p.Smax : (int)Sq);
  const int64_t* tg = p.targets + (int64_t)b * p.tgt_stride;
  for (int i = tid; i < S; i += 256) lab[i] = (int)tg[i];
  for (int k = tid; k < p.VC; k += 256) p.clabel[(size_t)b * p.VC + k] = (k == p.VC - 1) ? p.blank : -1;
  __syncthreads();
  for (int i = tid; i < S; i += 256) {
    // (no early exit: the reads of lab[k] are then independent and pipeline -- with a break every iteration waited for its
    //  LDS read, 23 us for 200 labels)
    const int li = lab[i];
    int r = i;
    for (int k = i - 1; k >= 0; k--) r = lab[k] == li ? k : r;
    rep[i] = r;
  }
  __syncthreads();
  for (int i = tid; i < S; i += 256) {
    int c = 0;
    if (rep[i] == i) for (int k = 0; k < i; k++) c += rep[k] == k ? 1 : 0;
    cid[i] = c;
  }
  __syncthreads();
  for (int i = tid; i < p.Smax; i += 256) {
    int64_t out = p.VC - 1;                       // beyond the utterance's targets: never read by the lattice
    if (i < S) {
      const int li = lab[i];
      const int c = cid[rep[i]];
      // a target equal to the blank id, or outside the alphabet, keeps the reference's semantics by mapping to the
      // compact blank column: the lattice kernels then hand the utterance to the exact path (ctc_loss.cpp:53,109-113)
      out = li == p.blank ? p.VC - 1 : c;
      if (li < 0 || li >= p.V) out = p.VC;       // outside the alphabet: stays outside the compact one (-> NaN, e2e_ctc.h)
      if (rep[i] == i && out < p.VC - 1) p.clabel[(size_t)b * p.VC + c] = li;
    }
    p.targets_c[(size_t)b * p.Smax + i] = out;
  }
}

// log-sum-exp of every live frame + its compact log-probs.  One wave per frame.
// E: the logits' element type.  16-bit logits take the element-wise form (VEC4 = false) -- rows of more than 8192 or of unaligned
// columns, which the single-read kernels below do not hold: round 6; until then such a call was refused and the host up-cast it.
template <bool VEC4, typename E = float>
__global__ __launch_bounds__(64 * kWaves) void wide_rows_kernel(WideParams p) {
  static_assert(!VEC4 || sizeof(E) == 4, "16-byte accesses: f32 rows");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * kWaves + w;
  if (row >= (int64_t)p.B * p.T) return;
  int b, t; split_frame(row, p.T, b, t);
  const int64_t Tq = p.x_len[b];
  if (t >= Tq) return;
  const E* xr = reinterpret_cast<const E*>(p.x) + (int64_t)b * p.sB + (int64_t)t * p.sT;
  float lse = 0.f;
  if (!p.logprobs) {
    float m = -__builtin_huge_valf(), s = 0.f;
    if constexpr (VEC4) {
      const float4* x4 = reinterpret_cast<const float4*>(xr);
      const int n4 = p.V >> 2;
      int i = lane;
      for (; i + 192 < n4; i += 256) {              // 4 independent 16-byte loads in flight, one rescale test per 16 values
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = x4[i + 64 * u];
        float cm = v[0].x;
#pragma unroll
        for (int u = 0; u < 4; u++) cm = fmaxf(cm, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
        if (cm > m) { s *= exp_acc(m - cm); m = cm; }
#pragma unroll
        for (int u = 0; u < 4; u++) s += exp_acc(v[u].x - m) + exp_acc(v[u].y - m) + exp_acc(v[u].z - m) + exp_acc(v[u].w - m);
      }
      for (; i < n4; i += 64) {
        const float4 v = x4[i];
        const float cm = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        if (cm > m) { s *= exp_acc(m - cm); m = cm; }
        s += exp_acc(v.x - m) + exp_acc(v.y - m) + exp_acc(v.z - m) + exp_acc(v.w - m);
      }
      for (int i = (n4 << 2) + lane; i < p.V; i += 64) {
        const float v = xr[i];
        if (v > m) { s *= exp_acc(m - v); m = v; }
        s += exp_acc(v - m);
      }
    } else {
      for (int i = lane; i < p.V; i += 64) {
        const float v = (float)xr[(int64_t)i * p.sV];
        if (v > m) { s *= exp_acc(m - v); m = v; }
        s += exp_acc(v - m);
      }
    }
    const float M = wave_max_f(m);
    s = wave_sum_f(m > -__builtin_huge_valf() ? s * exp_acc(m - M) : 0.f);
    lse = M + logf(s);
    if (lane == 0) p.lse[row] = lse;
  }
  // Compact row, shifted so that its largest entry is 0.  A common per-frame factor cancels in the posteriors; it
  // keeps the lattice rows from decaying by ~V per step (2^-13 at V=8000), which the f32 segment kernel could not
  // bridge between two rescales.  The loss is corrected by sum_t shift_t afterwards (wide_loss_fix_kernel).
  float* xc = p.xc + (size_t)row * p.VC;
  const int* cl = p.clabel + (size_t)b * p.VC;
  float cm = -__builtin_huge_valf();
  for (int k = lane; k < p.VC; k += 64) {
    const int l = cl[k];
    const float v = l >= 0 ? (float)xr[(int64_t)l * p.sV] - lse : -__builtin_huge_valf();
    xc[k] = v;
    cm = fmaxf(cm, v);
  }
  cm = wave_max_f(cm);
  if (!(cm > -__builtin_huge_valf())) cm = 0.f;                  // every compact entry is log 0: nothing to shift
  for (int k = lane; k < p.VC; k += 64) xc[k] -= cm;             // same lanes, same addresses as above
  if (lane == 0) p.shift[row] = cm;
}

// loss_true = loss_shifted - sum_{t < T} shift_t   (log Z gains the sum of the removed per-frame factors)
__global__ __launch_bounds__(64) void wide_loss_fix_kernel(WideParams p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t Tq = p.x_len[b];
  if (Tq < 1 || Tq > p.T) return;
  double s = 0.0;
  for (int t = lane; t < (int)Tq; t += 64) s += (double)p.shift[(size_t)b * p.T + t];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) p.losses[b] = (float)((double)p.losses[b] - s);
}

// dense gradient rows.  One wave per frame.
template <bool VEC4, typename E = float>
__global__ __launch_bounds__(64 * kWaves) void wide_emit_kernel(WideParams p) {
  static_assert(!VEC4 || sizeof(E) == 4, "16-byte accesses: f32 rows");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * kWaves + w;
  if (row >= (int64_t)p.B * p.T) return;
  int b, t; split_frame(row, p.T, b, t);
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  const E* xr = reinterpret_cast<const E*>(p.x) + (int64_t)b * p.sB + (int64_t)t * p.sT;
  E* gr = reinterpret_cast<E*>(p.grads) + (size_t)row * p.V;
  const bool bad_len = Tq < 1 || Tq > p.T || Sq < 0 || Sq > p.Smax;
  const float loss = p.losses[b];
  const bool poison = bad_len || !(loss < __builtin_huge_valf());      // invalid lengths / infeasible (Q2) / NaN
  const bool live = !poison && t < Tq;
  const bool zero = !poison && !live && !p.logprobs;                   // padded frame, fused logits: 0 (else exp(lp), Q1)
  const float lse = (live && !p.logprobs) ? p.lse[row] : 0.f;
  const float qnan = __builtin_nanf("");
  // label columns (prob - posterior, with posterior = exp(xc) - gc in the shifted compact space): fetched and computed
  // up front, written after the dense row
  constexpr int kMaxFix = 2;            // the first 128 compact columns (the rest: after the row, below)
  float fix[kMaxFix]; int fixcol[kMaxFix];
#pragma unroll
  for (int u = 0; u < kMaxFix; u++) { fix[u] = 0.f; fixcol[u] = -1; }
  // (no alignment AND a target equal to the blank id: the reference leaves -inf in the columns that have a cell with a finite
  //  alpha + beta -- see ctc_exact_one -- and the compact gradient carries that pattern into the poisoned row)
  const bool inf_pattern = !bad_len && loss == __builtin_huge_valf() && t < Tq;
  if (live || inf_pattern) {
    const float* gc = p.gc + (size_t)row * p.VC;
    const int* cl = p.clabel + (size_t)b * p.VC;
    const float sh = live ? p.shift[row] : 0.f;
#pragma unroll
    for (int u = 0; u < kMaxFix; u++) {
      const int k = lane + 64 * u;
      if (k < p.VC) {
        const int l = cl[k];
        if (l >= 0 && live) {
          const float xl = (float)xr[(int64_t)l * p.sV] - lse;
          fix[u] = (exp_acc(xl) - (exp_acc(xl - sh) - gc[k])) * p.gscale;
          fixcol[u] = l;
        } else if (l >= 0 && gc[k] == -__builtin_huge_valf()) {
          fix[u] = -__builtin_huge_valf();
          fixcol[u] = l;
        }
      }
    }
  }
  if constexpr (VEC4) {
    typedef float vf4 __attribute__((ext_vector_type(4)));
    const vf4* x4 = reinterpret_cast<const vf4*>(xr);
    vf4* g4 = reinterpret_cast<vf4*>(gr);
    const int n4 = p.V >> 2;
    if (poison || zero) {
      const float f = poison ? qnan : 0.f;
      const vf4 o = {f, f, f, f};
      for (int i = lane; i < n4; i += 64) __builtin_nontemporal_store(o, &g4[i]);
    } else {
      // 8 independent 16-byte loads in flight per lane, then their exps and (streaming) stores
      int i = lane;
      for (; i + 448 < n4; i += 512) {
        vf4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(&x4[i + 64 * u]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const vf4 o = {exp_acc(v[u].x - lse), exp_acc(v[u].y - lse), exp_acc(v[u].z - lse), exp_acc(v[u].w - lse)};
          __builtin_nontemporal_store(o * p.gscale, &g4[i + 64 * u]);
        }
      }
      for (; i + 64 < n4; i += 128) {
        vf4 v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) v[u] = __builtin_nontemporal_load(&x4[i + 64 * u]);
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const vf4 o = {exp_acc(v[u].x - lse), exp_acc(v[u].y - lse), exp_acc(v[u].z - lse), exp_acc(v[u].w - lse)};
          __builtin_nontemporal_store(o * p.gscale, &g4[i + 64 * u]);
        }
      }
      for (; i < n4; i += 64) {
        const vf4 v = x4[i];
        const vf4 o = {exp_acc(v.x - lse), exp_acc(v.y - lse), exp_acc(v.z - lse), exp_acc(v.w - lse)};
        g4[i] = o * p.gscale;
      }
    }
    for (int i = (n4 << 2) + lane; i < p.V; i += 64) gr[i] = poison ? qnan : (zero ? 0.f : exp_acc(xr[i] - lse) * p.gscale);
  } else {
    for (int i = lane; i < p.V; i += 64) gr[i] = (E)(poison ? qnan : (zero ? 0.f : exp_acc((float)xr[(int64_t)i * p.sV] - lse) * p.gscale));
  }
  if (live || inf_pattern) {
    // the dense row above and these columns are written by different lanes of this wave: order them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int u = 0; u < kMaxFix; u++) if (fixcol[u] >= 0) gr[fixcol[u]] = (E)fix[u];
    // compact columns beyond the 128 fetched up front (targets of more than 127 distinct labels)
    const float* gc = p.gc + (size_t)row * p.VC;
    const int* cl = p.clabel + (size_t)b * p.VC;
    const float sh = live ? p.shift[row] : 0.f;
    for (int k = lane + 64 * kMaxFix; k < p.VC; k += 64) {
      const int l = cl[k];
      if (l >= 0 && live) {
        const float xl = (float)xr[(int64_t)l * p.sV] - lse;
        gr[l] = (E)((exp_acc(xl) - (exp_acc(xl - sh) - gc[k])) * p.gscale);
      } else if (l >= 0 && gc[k] == -__builtin_huge_valf()) {
        gr[l] = (E)(-__builtin_huge_valf());
      }
    }
  }
}

// ---- single-read form: the row stays in registers between the softmax's two passes ----
typedef float vf4 __attribute__((ext_vector_type(4)));
// E: the logits' (and the gradient's) element type -- float, or f16_t / bf16_t read and written 16 bytes (8 elements) at a
// time and converted in registers: 2*V*sizeof(E) bytes per frame.  NV4: float4-equivalents of registers per lane, rows of
// up to 256*NV4 columns either way.
// Cache policy of the dense row kernels (bit 0: non-temporal loads, bit 1: non-temporal stores).  Round 6, one process, B=512, T=256,
// V=8000 (tools/diag/ab_time.py): both non-temporal -- the form until then -- f32 1743 / bf16 990 us per call; neither 1859 / 1037;
// loads only 2007 / 1162; **stores only 1663 / 912**.  The row is read once, but not only once: the wave's tail gathers the
// utterance's <= S+1 label columns out of it, and behind a non-temporal load those are a second trip to HBM instead of an L2 hit.
#ifndef E2E_WIDE_NT
#define E2E_WIDE_NT 2
#endif
#if E2E_WIDE_NT & 1
#define DENSE_LD(p) __builtin_nontemporal_load(p)
#else
#define DENSE_LD(p) (*(p))
#endif
#if E2E_WIDE_NT & 2
#define DENSE_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define DENSE_ST(v, p) (*(p) = (v))
#endif
template <int NV4, typename E>
__device__ __forceinline__ void wide_rows_dense_body(const WideParams& p) {
  constexpr int EPC = 16 / (int)sizeof(E);            // elements per 16-byte chunk
  constexpr int NCH = NV4 * 4 / EPC;                  // chunks per lane
  constexpr bool PACKED = sizeof(E) == 2;             // 16-bit rows stay packed in registers between the passes (below)
  typedef E ev __attribute__((ext_vector_type(EPC)));
  typedef float fv __attribute__((ext_vector_type(EPC)));
  // (the wave's row is made visibly uniform: the row pointers then live in scalar registers and every load / store is
  // `scalar base + lane offset + immediate` -- with per-lane 64-bit addresses the 2*NCH of them cost more VGPRs than the row)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * kWaves + w;
  if (row >= (int64_t)p.B * p.T) return;
  int b, t; split_frame(row, p.T, b, t);
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  const E* xr = reinterpret_cast<const E*>(p.x) + (int64_t)b * p.sB + (int64_t)t * p.sT;
  ev* g4 = reinterpret_cast<ev*>(reinterpret_cast<E*>(p.grads) + (size_t)row * p.V);
  const ev* x4 = reinterpret_cast<const ev*>(xr);
  const int n4 = p.V / EPC;
  const bool bad_len = Tq < 1 || Tq > p.T || Sq < 0 || Sq > p.Smax;
  const bool live = !bad_len && t < Tq;
  if (bad_len || (!live && !p.logprobs)) {         // invalid lengths: NaN; padded frame of fused logits: 0 (else exp(lp), Q1)
    const float f = bad_len ? __builtin_nanf("") : 0.f;
    fv of;
#pragma unroll
    for (int e = 0; e < EPC; e++) of[e] = f;
    const ev o = __builtin_convertvector(of, ev);
    for (int i = lane; i < n4; i += 64) DENSE_ST(o, &g4[i]);
    return;
  }
  const float ninf = -__builtin_huge_valf();
  // Groups of 64 chunks: all but at most one are complete (unconditional accesses at `scalar base + lane + immediate`);
  // in the one partial group the lanes past the row re-read -- and later re-write, with the same value -- the row's last
  // chunk and are left out of the sum; groups past the row (a narrower alphabet than the instantiation holds) are
  // skipped by uniform branches.
  // f32 rows are held as they are.  16-bit rows are held PACKED (half the registers: four to five rows per SIMD in flight
  // instead of two, which a row of half the bytes needs to keep the same number of bytes in flight -- converted to f32 the
  // kernel ran 1.70 ms against the f32 kernel's 1.62 at V = 8000): every pass converts the chunk it works on, and the
  // exponentials exp(x - max) in [0, 1] go back into the same registers in the row's own 16-bit type (the sum is taken
  // from the unrounded values; the gradient is that value times 1/sum, rounded once more: <= 1.5 ulp of the output type).
  // (f16: the held exponentials are scaled by 2^14 -- unscaled, everything below 6e-5 would sit in f16's subnormals with an absolute
  //  error of 3e-8 that a fused grad_scale > 1, a loss-scaling factor, multiplies up beyond an ulp of what is written)
  typedef typename std::conditional<PACKED, ev, fv>::type held;
  constexpr float kHeld = (PACKED && std::is_same<E, f16_t>::value) ? 16384.f : 1.f;
  held v[NCH];
  const int last = n4 - 1;
  const int part_idx = min(lane + (n4 & ~63), last);      // this lane's chunk in the partial group
  const bool part_in = lane + (n4 & ~63) <= last;
  auto as_f32 = [](const held& h) -> fv { if constexpr (PACKED) return __builtin_convertvector(h, fv); else return h; };
  auto to_held = [](const fv& f) -> held { if constexpr (PACKED) return __builtin_convertvector(f, ev); else return f; };
#pragma unroll
  for (int u = 0; u < NCH; u++) {
    if (64 * u + 64 <= n4) { const ev r = DENSE_LD(&x4[64 * u + lane]); if constexpr (PACKED) v[u] = r; else v[u] = __builtin_convertvector(r, fv); }
    else if (64 * u < n4) { const ev r = DENSE_LD(&x4[part_idx]); if constexpr (PACKED) v[u] = r; else v[u] = __builtin_convertvector(r, fv); }
    else {
      fv none;
#pragma unroll
      for (int e = 0; e < EPC; e++) none[e] = ninf;
      v[u] = to_held(none);
    }
  }
  float lse = 0.f;
  if (!p.logprobs) {
    float m = ninf;
#pragma unroll
    for (int u = 0; u < NCH; u++) {
      const fv f = as_f32(v[u]);
#pragma unroll
      for (int e = 0; e < EPC; e++) m = fmaxf(m, f[e]);
    }
    const float M = wave_max_f(m), mM = -M * 1.44269504088896340736f;
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NCH; u++) {
      if (64 * u < n4) {
        fv f = as_f32(v[u]);
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; e++) { f[e] = exp_row(f[e], M, mM); part += f[e]; }
        v[u] = to_held(f * kHeld);
        sum += (64 * u + 64 <= n4 || part_in) ? part : 0.f;
      }
      if ((u * EPC / 4) & 1) __builtin_amdgcn_sched_barrier(0);       // (8 exps in flight are enough; interleaving all of them costs registers)
    }
    sum = wave_sum_f(sum);
    const float inv = 1.f / sum;
    const float invg = inv * p.gscale * (1.f / kHeld);
    lse = M + logf(sum);
#pragma unroll
    for (int u = 0; u < NCH; u++) {
      if (64 * u + 64 <= n4) DENSE_ST(__builtin_convertvector(as_f32(v[u]) * invg, ev), &g4[64 * u + lane]);
      else if (64 * u < n4) DENSE_ST(__builtin_convertvector(as_f32(v[u]) * invg, ev), &g4[part_idx]);
      if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (lane == 0) p.lse[row] = lse;
  } else {
#pragma unroll
    for (int u = 0; u < NCH; u++) {
      if (64 * u < n4) {
        const fv f = as_f32(v[u]);
        fv o;
#pragma unroll
        for (int e = 0; e < EPC; e++) o[e] = exp_acc(f[e]);
        if (64 * u + 64 <= n4) DENSE_ST(__builtin_convertvector(o * p.gscale, ev), &g4[64 * u + lane]);
        else DENSE_ST(__builtin_convertvector(o * p.gscale, ev), &g4[part_idx]);
      }
      if (u & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!live) return;
  // compact row, shifted so that its largest entry is 0 (see wide_rows_kernel)
  float* xc = p.xc + (size_t)row * p.VC;
  const int* cl = p.clabel + (size_t)b * p.VC;
  float cm = ninf;
  for (int k = lane; k < p.VC; k += 64) {
    const int l = cl[k];
    const float val = l >= 0 ? (float)xr[l] - lse : ninf;
    xc[k] = val;
    cm = fmaxf(cm, val);
  }
  cm = wave_max_f(cm);
  if (!(cm > ninf)) cm = 0.f;
  for (int k = lane; k < p.VC; k += 64) xc[k] -= cm;
  if (lane == 0) p.shift[row] = cm;
}

template <int NV4, typename E>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2)))      // (<= 256 VGPRs: two rows per SIMD in flight)
void wide_rows_dense_kernel(WideParams p) { wide_rows_dense_body<NV4, E>(p); }
// 16-bit rows (held packed: see the body): three rows per SIMD in flight (four: 76 bytes of scratch per lane at V > 4096)
template <int NV4, typename E>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(3)))
void wide_rows_dense_kernel_16(WideParams p) { wide_rows_dense_body<NV4, E>(p); }
// (Two 16-bit rows per wave -- both asked for before either is looked at: 384 KB in flight per CU instead of 192 -- were built in round 6
//  and are slower, 1131 against 985 us per call for bf16 in one process (tools/diag/ab_time.py with AB_DTYPE): removed.)

// after the lattice: the label columns of the live frames; the slab of an utterance that turned out infeasible
template <typename E>
__global__ __launch_bounds__(64 * kWaves) void wide_fix_kernel(WideParams p) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * kWaves + w;
  if (row >= (int64_t)p.B * p.T) return;
  int b, t; split_frame(row, p.T, b, t);
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  if (Tq < 1 || Tq > p.T || Sq < 0 || Sq > p.Smax) return;         // (poisoned by wide_rows_dense_kernel already)
  E* gr = reinterpret_cast<E*>(p.grads) + (size_t)row * p.V;
  const float loss_b = p.losses[b];
  if (t == 0) {
    // the utterance's first frame also corrects its loss (wide_loss_fix_kernel's job, folded in here in round 6: a launch of its own
    // was 5 us behind the lattice): loss_true = loss_shifted - sum_t shift_t.  The other frames of the utterance read the loss only to
    // tell a number from +inf / NaN, which the correction does not change.
    double sacc = 0.0;
    for (int u = lane; u < (int)Tq; u += 64) sacc += (double)p.shift[(size_t)b * p.T + u];
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
    if (lane == 0) p.losses[b] = (float)((double)loss_b - sacc);
  }
  if (!(loss_b < __builtin_huge_valf())) {                           // infeasible (Q2) / NaN: the whole slab
    constexpr int EPC = 16 / (int)sizeof(E);
    typedef E ev __attribute__((ext_vector_type(EPC)));
    ev o;
#pragma unroll
    for (int e = 0; e < EPC; e++) o[e] = (E)__builtin_nanf("");
    ev* g4 = reinterpret_cast<ev*>(gr);
    for (int i = lane; i < p.V / EPC; i += 64) __builtin_nontemporal_store(o, &g4[i]);
    if (t < Tq) {
      // (no alignment AND a target equal to the blank id: the reference leaves -inf in the columns that have a cell with a
      //  finite alpha + beta -- see ctc_exact_one -- and the compact gradient carries that pattern)
      const float* gc = p.gc + (size_t)row * p.VC;
      const int* cl = p.clabel + (size_t)b * p.VC;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the row's NaN stores have landed)
      for (int k = lane; k < p.VC; k += 64) {
        const int l = cl[k];
        if (l >= 0 && gc[k] == -__builtin_huge_valf()) gr[l] = (E)(-__builtin_huge_valf());
      }
    }
    return;
  }
  if (t >= Tq) return;
  // prob - posterior, the posterior in the shifted compact space: exp(xc) - gc.  Everything comes from the compact
  // workspace row -- the probability is exp(xc + shift) again -- so the column is WRITTEN, not read-modified: a 4-byte
  // store into a line that left the caches long ago needs no fill (the read half of the fix-up's traffic).
  const float* gc = p.gc + (size_t)row * p.VC;
  const float* xc = p.xc + (size_t)row * p.VC;
  const int* cl = p.clabel + (size_t)b * p.VC;
  const float sh = p.shift[row];
  // A column whose posterior is nothing in f32 -- the lattice's prob - posterior came back as the probability itself, bit for
  // bit: the label cannot be emitted at this frame, or so unlikely that the subtraction does not change the number -- already
  // holds its value (wide_rows_dense_kernel wrote the probability) and is left alone: a 4-byte store into a line that has
  // left the caches is a read-modify-write of the line in DRAM, and those stores are what this kernel's time consists of.
  for (int k = lane; k < p.VC; k += 64) {
    const int l = cl[k];
    const float xk = xc[k], gk = gc[k];
    const float yk = exp_acc(xk);
    if (l >= 0 && gk != yk) {
      const float pf = exp_acc(xk + sh);
      const E nv = (E)((pf - (yk - gk)) * p.gscale);
      // (16-bit gradients, round 6: a posterior that does not move the probability's 16-bit rounding leaves the column as the row
      //  kernel wrote it -- the same number to the type's resolution -- and saves the store: with bf16's eight bits that is every
      //  label column whose posterior is below ~0.2 % of its probability)
      if (sizeof(E) == 4 || nv != (E)(pf * p.gscale)) gr[l] = nv;
    }
  }
}

// More distinct labels than the lattice kernels' alphabet holds (targets of more than 95 word pieces): the compaction and the
// streaming kernels still carry the V-wide work -- 2 V 4 bytes per frame -- and the lattice on the <= S+1 compact columns is
// left to the reference's arithmetic (ctc_exact_kernel in log-prob mode, 256 utterances at a time on one set of alpha slabs).
// B=64, T=256, V=8000, S<=200: 6.7 ms when the exact kernel also had to walk the 8000 columns of every frame in f64.
constexpr int kWideExactChunk = 256;
static bool wide_inner_exact(int T, int Smax) { return !fast_supported(T, Smax + 1, Smax, E2E_F32); }

struct WideLayout { size_t targets_c, clabel, lse, shift, xc, gc, inner, total; int VC; };

WideLayout wide_layout(int B, int T, int V, int Smax, bool with_exact) {
  (void)V;
  WideLayout l;
  l.VC = Smax + 1;
  size_t o = 0;
  l.targets_c = o; o += align_up((size_t)B * (Smax > 0 ? Smax : 1) * sizeof(int64_t), 256);
  l.clabel = o; o += align_up((size_t)B * l.VC * sizeof(int), 256);
  l.lse = o; o += align_up((size_t)B * T * sizeof(float), 256);
  l.shift = o; o += align_up((size_t)B * T * sizeof(float), 256);
  l.xc = o; o += align_up((size_t)B * T * l.VC * sizeof(float), 256);
  l.gc = o; o += align_up((size_t)B * T * l.VC * sizeof(float), 256);
  l.inner = o;
  if (wide_inner_exact(T, Smax)) {
    o += exact_workspace_bytes(B < kWideExactChunk ? B : kWideExactChunk, T, l.VC, Smax);
  } else {
    o += fast_workspace_bytes(B, T, l.VC, Smax);
    if (with_exact) o += exact_fallback_workspace_bytes(B, T, l.VC, Smax);
  }
  l.total = o;
  return l;
}

}  // namespace

bool wide_supported(int T, int V, int Smax, int dtype) {
  if (!((dtype == E2E_F32 || dtype_is_16bit(dtype)) && V > 1 && Smax >= 0)) return false;
  // (with the lattice left to the exact kernel the compaction must at least halve the columns to be worth its passes)
  return fast_supported(T, Smax + 1, Smax, dtype) || V >= 2 * (Smax + 1);
}
bool wide_takes_fast_lattice(int T, int V, int Smax, int dtype) {
  return wide_supported(T, V, Smax, dtype) && !wide_inner_exact(T, Smax);
}

size_t wide_workspace_bytes(int B, int T, int V, int Smax, bool with_exact) {
  return wide_layout(B, T, V, Smax, with_exact).total;
}

int launch_wide(const LossArgs& a, bool fallback_to_exact) {
  const WideLayout l = wide_layout(a.B, a.T, a.V, a.Smax, fallback_to_exact);
  if (!a.ws || a.ws_bytes < l.total) { set_error("workspace too small: %zu < %zu", a.ws_bytes, l.total); return E2E_ERR_WORKSPACE; }
  if (a.B == 0) return E2E_OK;
  char* ws = reinterpret_cast<char*>(a.ws);
  WideParams p;
  p.x = reinterpret_cast<const float*>(a.x); p.sB = a.sB; p.sT = a.sT; p.sV = a.sV; p.logprobs = a.logprobs;
  p.targets = a.targets; p.tgt_stride = a.tgt_stride; p.x_len = a.x_len; p.t_len = a.t_len;
  p.B = a.B; p.T = a.T; p.V = a.V; p.Smax = a.Smax; p.VC = l.VC; p.blank = a.blank;
  p.grads = reinterpret_cast<float*>(a.grads); p.losses = reinterpret_cast<float*>(a.losses);
  p.gscale = (float)a.grad_scale;
  p.targets_c = reinterpret_cast<int64_t*>(ws + l.targets_c); p.clabel = reinterpret_cast<int*>(ws + l.clabel);
  p.lse = reinterpret_cast<float*>(ws + l.lse); p.shift = reinterpret_cast<float*>(ws + l.shift);
  p.xc = reinterpret_cast<float*>(ws + l.xc);
  p.gc = reinterpret_cast<const float*>(ws + l.gc);
  const bool io16 = dtype_is_16bit(a.dtype);
  const int esz = io16 ? 2 : 4, epc = 16 / esz;    // element size, elements per 16-byte access
  const bool vec4 = a.sV == 1 && (a.sT % epc == 0) && (a.sB % epc == 0) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0) &&
                    (a.V % epc == 0) && (reinterpret_cast<uintptr_t>(a.grads) % 16 == 0);
  const bool dense = vec4 && a.V <= 8192;          // the row fits a wave's registers: logits read once
  hipLaunchKernelGGL(wide_compact_kernel, dim3(a.B), dim3(256), sizeof(int) * 3 * (a.Smax > 0 ? a.Smax : 1), a.stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "wide_compact_kernel launch");

  // utterances [b0, b0 + nb): the row kernel on stream `s_rows`; the lattice on the compact alphabet, the loss
  // correction and the label-column fix-up on stream `s_lat`
  auto rows_part = [&](int b0, int nb, hipStream_t s_rows) -> int {
    WideParams q = p;
    q.B = nb; q.x_len = a.x_len + b0; q.t_len = a.t_len + b0;
    q.x = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.x) + (int64_t)b0 * a.sB * esz);
    q.grads = reinterpret_cast<float*>(reinterpret_cast<char*>(p.grads) + (size_t)b0 * a.T * a.V * esz); q.losses = p.losses + b0;
    q.clabel = p.clabel + (size_t)b0 * l.VC; q.lse = p.lse + (size_t)b0 * a.T; q.shift = p.shift + (size_t)b0 * a.T;
    q.xc = p.xc + (size_t)b0 * a.T * l.VC; q.gc = p.gc + (size_t)b0 * a.T * l.VC;
    const dim3 grid_rows((unsigned)(((int64_t)nb * a.T + kWaves - 1) / kWaves));
    if (dense) {
      auto rows = [&](auto elem_tag) {
        typedef decltype(elem_tag) E;
        if constexpr (sizeof(E) == 2) {
          if (a.V <= 2048) hipLaunchKernelGGL((wide_rows_dense_kernel_16<8, E>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
          else if (a.V <= 4096) hipLaunchKernelGGL((wide_rows_dense_kernel_16<16, E>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
          else hipLaunchKernelGGL((wide_rows_dense_kernel_16<32, E>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
        } else {
          if (a.V <= 2048) hipLaunchKernelGGL((wide_rows_dense_kernel<8, E>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
          else if (a.V <= 4096) hipLaunchKernelGGL((wide_rows_dense_kernel<16, E>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
          else hipLaunchKernelGGL((wide_rows_dense_kernel<32, E>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
        }
      };
      if (a.dtype == E2E_F16) rows(f16_t{}); else if (a.dtype == E2E_BF16) rows(bf16_t{}); else rows(float{});
    } else if (a.dtype == E2E_F16) hipLaunchKernelGGL((wide_rows_kernel<false, f16_t>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
    else if (a.dtype == E2E_BF16) hipLaunchKernelGGL((wide_rows_kernel<false, bf16_t>), grid_rows, dim3(64 * kWaves), 0, s_rows, q);
    else if (vec4) hipLaunchKernelGGL(wide_rows_kernel<true>, grid_rows, dim3(64 * kWaves), 0, s_rows, q);
    else hipLaunchKernelGGL(wide_rows_kernel<false>, grid_rows, dim3(64 * kWaves), 0, s_rows, q);
    E2E_HIP_CHECK(hipGetLastError(), "wide rows kernel launch");
    return E2E_OK;
  };
  auto lattice_part = [&](int b0, int nb, hipStream_t s_lat) -> int {
    WideParams q = p;
    q.B = nb; q.x_len = a.x_len + b0; q.t_len = a.t_len + b0;
    q.x = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.x) + (int64_t)b0 * a.sB * esz);
    q.grads = reinterpret_cast<float*>(reinterpret_cast<char*>(p.grads) + (size_t)b0 * a.T * a.V * esz); q.losses = p.losses + b0;
    q.clabel = p.clabel + (size_t)b0 * l.VC; q.lse = p.lse + (size_t)b0 * a.T; q.shift = p.shift + (size_t)b0 * a.T;
    q.xc = p.xc + (size_t)b0 * a.T * l.VC; q.gc = p.gc + (size_t)b0 * a.T * l.VC;
    const dim3 grid_rows((unsigned)(((int64_t)nb * a.T + kWaves - 1) / kWaves));
    // the lattice on the compact alphabet: log-probabilities in, (prob - posterior) out
    LossArgs c = a;
    c.B = nb; c.stream = s_lat;
    c.x = q.xc; c.dtype = E2E_F32; c.logprobs = 1;
    c.sB = (int64_t)a.T * l.VC; c.sT = l.VC; c.sV = 1;
    c.targets = p.targets_c + (size_t)b0 * (a.Smax > 0 ? a.Smax : 1); c.tgt_stride = a.Smax > 0 ? a.Smax : 1;
    c.x_len = q.x_len; c.t_len = q.t_len; c.losses = q.losses;
    c.V = l.VC; c.blank = l.VC - 1;
    c.grads = const_cast<float*>(q.gc);
    c.ws = ws + l.inner; c.ws_bytes = a.ws_bytes - l.inner;
    c.grad_scale = 1.0; c.reduced = nullptr; c.reduction = 0;      // (the compact gradient stays unscaled; the losses
                                                                    //  are corrected below, the caller reduces them after)
    if (wide_inner_exact(a.T, a.Smax)) {
      for (int c0 = 0; c0 < nb; c0 += kWideExactChunk) {
        LossArgs e = c;
        e.B = nb - c0 < kWideExactChunk ? nb - c0 : kWideExactChunk;
        e.x = q.xc + (size_t)c0 * a.T * l.VC; e.grads = const_cast<float*>(q.gc) + (size_t)c0 * a.T * l.VC;
        e.targets = c.targets + (size_t)c0 * c.tgt_stride; e.x_len = c.x_len + c0; e.t_len = c.t_len + c0;
        e.losses = reinterpret_cast<float*>(c.losses) + c0;
        const int rc = launch_exact(e);
        if (rc != E2E_OK) return rc;
      }
    } else {
      const int rc = launch_fast(c, fallback_to_exact);
      if (rc != E2E_OK) return rc;
    }
    if (!dense) {                      // (the dense path's fix-up corrects the losses itself)
      hipLaunchKernelGGL(wide_loss_fix_kernel, dim3(nb), dim3(64), 0, s_lat, q);
      E2E_HIP_CHECK(hipGetLastError(), "wide_loss_fix_kernel launch");
    }
    if (dense) {
      if (a.dtype == E2E_F16) hipLaunchKernelGGL(wide_fix_kernel<f16_t>, grid_rows, dim3(64 * kWaves), 0, s_lat, q);
      else if (a.dtype == E2E_BF16) hipLaunchKernelGGL(wide_fix_kernel<bf16_t>, grid_rows, dim3(64 * kWaves), 0, s_lat, q);
      else hipLaunchKernelGGL(wide_fix_kernel<float>, grid_rows, dim3(64 * kWaves), 0, s_lat, q);
    }
    else if (a.dtype == E2E_F16) hipLaunchKernelGGL((wide_emit_kernel<false, f16_t>), grid_rows, dim3(64 * kWaves), 0, s_lat, q);
    else if (a.dtype == E2E_BF16) hipLaunchKernelGGL((wide_emit_kernel<false, bf16_t>), grid_rows, dim3(64 * kWaves), 0, s_lat, q);
    else if (vec4) hipLaunchKernelGGL(wide_emit_kernel<true>, grid_rows, dim3(64 * kWaves), 0, s_lat, q);
    else hipLaunchKernelGGL(wide_emit_kernel<false>, grid_rows, dim3(64 * kWaves), 0, s_lat, q);
    E2E_HIP_CHECK(hipGetLastError(), "wide emit kernel launch");
    return E2E_OK;
  };

  // (Chunking the batch -- lattice + fix-up of chunk k on an internal second stream under the row kernel of chunk k+1, so
  // that the fix-up would find its gradient lines in the Infinity Cache -- was built and measured at B=512, T=256, V=8000:
  // 16 utterances per chunk 3.37 ms, 32: 2.29, 64: 2.09, 128: 2.05 against 2.00 ms unchunked.  A chunk's lattice is
  // latency-bound (~100 us of launches and serial steps whatever its size) and the fix-up is no faster behind it.  Round 5, the
  // second stream at the highest queue priority, two chunks: f32 1.76 against 1.70 ms, bf16 1.075 against 1.053 --
  // tools/diag/experiments/wide_fork_r05.patch.)
  int rc = rows_part(0, a.B, a.stream);
  if (rc != E2E_OK) return rc;
  return lattice_part(0, a.B, a.stream);
}

}  // namespace e2e

// Exact CTC forward-backward: f64 log-domain lattice, the reference's own
// arithmetic (src/losses/ctc_loss.cpp:15-118), one workgroup per utterance.
//
// This is the always-correct path: it takes f64 inputs (gradcheck), -inf
// log-probs, infeasible alignments (loss=+inf, grads=NaN, quirk Q2) and any
// utterance the fast scaled path flags as out of range.  It is not the
// throughput path -- f64 exp/log dominate -- see ctc_loss_fast.hip.
//
// HBM layout: alpha rows go to the workspace as [b][t][j] doubles (row stride
// Lmax = 2*Smax+1) so that a time step is one coalesced row write in the alpha
// sweep and one coalesced row read in the beta sweep.  LDS holds the extended
// label row, the previous lattice row (double-buffered), the label-sorted cell
// order with one posterior sum per distinct label, and one BIT per alphabet column
// (is it one of the utterance's labels) -- nothing V-wide in doubles, so no alphabet
// is too wide for the kernel (word-piece vocabularies: 32 000 columns = 4 KB).
#include <string.h>

#include "common.h"

namespace e2e {
namespace {

constexpr int kThreads = 512;   // 8 waves: L <= 512 cells take one cell per lane

struct ExactParams {
  const void* x; int64_t sB, sT, sV; int logprobs;
  const int64_t* targets; int64_t tgt_stride;
  const int64_t* x_len; const int64_t* t_len;
  int B, T, V, Smax, Lmax, blank;
  void* losses; void* grads;
  double* ws_alpha;   // [B][T][Lmax]
  double* ws_lse;     // [B][T] row log-sum-exp (logits mode)
  int* ws_exp;        // [B][T] scaled form: exponent removed from alpha row t
  int scaled;         // 1: scaled probability-domain lattice where allowed (see ctc_exact_one), 0: the reference's log domain only
  int* flags;         // per-utterance "redo me" words written by the fast path (mode != 0); mode 1 adds bit 512: the f64
                      // redo of a segment could not settle the utterance
  int mode;           // 0: every utterance; 1: only flagged ones; 2: poison flagged ones, compute nothing
  int nslabs;         // alpha slabs in the workspace (mode 1: workgroups 0 .. nslabs-1 run the full recomputation)
  int redo_waves;     // waves of a workgroup that take part in the f64 redo of the segments (their LDS must fit: <= 8)
  int* ctl;           // flagged modes: the fast path's control words (0: tickets of this launch's workgroups)
  double gscale;      // every gradient element is multiplied by this as it is written
  void* reduced; int reduction;   // flagged modes: optional sum / mean of the losses, written by the last workgroup
  int has_retry; FastRetry retry;   // mode 1: the fast path's checkpoints, for the f64 redo of its second kernel
  int has_ext;        // mode 1: utterances whose numbers -- not whose inputs -- defeated the fast path are redone in extended range (ctc_ext.h)
};

__device__ __forceinline__ double neg_inf() { return -__builtin_huge_val(); }

// src/utils/math_utils.h:8-16 (log(1.0 + x), not log1p)
__device__ __forceinline__ double lse2(double a, double b) {
  if (a == neg_inf()) return b;
  if (b == neg_inf()) return a;
  if (a > b) return a + log(1.0 + exp(b - a));
  return b + log(1.0 + exp(a - b));
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// Reductions over the wave by DPP (a dozen cycles a stage instead of an LDS permute's hundred): the result is in LANE 63.
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_or0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false); }
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_or0(double v) {
  return __hiloint2double(dpp_or0<CTRL, ROWS>(__double2hiint(v)), dpp_or0<CTRL, ROWS>(__double2loint(v)));
}
__device__ __forceinline__ int wave_max_nonneg_lane63(int v) {
  v = max(v, dpp_or0<0xB1, 0xf>(v)); v = max(v, dpp_or0<0x4E, 0xf>(v));         // quads
  v = max(v, dpp_or0<0x141, 0xf>(v)); v = max(v, dpp_or0<0x140, 0xf>(v));       // rows of 16 (half mirror, mirror)
  v = max(v, dpp_or0<0x142, 0xa>(v)); v = max(v, dpp_or0<0x143, 0xc>(v));       // lane 15 -> next row, lane 31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
  v += dpp_or0<0xB1, 0xf>(v); v += dpp_or0<0x4E, 0xf>(v);
  v += dpp_or0<0x141, 0xf>(v); v += dpp_or0<0x140, 0xf>(v);
  v += dpp_or0<0x142, 0xa>(v); v += dpp_or0<0x143, 0xc>(v);
  return v;
}

// ------------------------------------------------------------------------------------------------------------
// f64 redo of ONE 16-step segment of the fast path's segment kernel (flag bits 8 / 16: the f32 recompute left its range).
// The chains' work stands -- loss, probabilities, alpha / beta checkpoints every 16 steps, all from f64 state -- so only
// the segments are done again, in doubles, ONE WAVE PER (flagged utterance, segment): every wave of the launch takes the
// segments whose running index is its own modulo the number of waves, so a batch with 73 flagged utterances is 4 600
// independent pieces of ~10 us on the whole chip instead of 73 workgroups walking 63 segments each (0.75 ms).
// Nothing is staged in HBM: a wave keeps the alpha rows entering steps 4, 8, 12 from one forward sweep (and the
// checkpoint for step 0), then takes the segment four rows at a time from the back -- recompute the four alpha rows from
// the kept one, walk beta through them -- 28 alpha steps for 16 rows.  Per-label sums by LDS f64 atomics, one row at a time.
// Returns false if a row sum is not a positive finite number or does not reproduce the chains' log Z (-> exact path).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_shift_up(double v) {      // lane n <- lane n-1, lane 0 <- 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_shift_down(double v) {    // lane n <- lane n+1, lane 63 <- 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

constexpr int kRetryGroup = 4;      // rows recomputed and held at a time

// per-wave LDS of the redo: the segment's probability rows [16][V] (floats), post[V] per-label sums of the row at hand,
// and the three kept alpha rows [3][8 cells][64 lanes] (registers hold the four rows in work, beta and the row being
// stepped: with the kept rows on top the kernel spilled 82 registers into its hot loops)
// (ppl: label pairs per lane of the segment kernel whose rows are redone -- the kept rows hold max(8, 2 ppl) cells per lane)
__host__ __device__ inline int retry_keep_cells(int ppl) { return 2 * ppl > 8 ? 2 * ppl : 8; }
__host__ __device__ inline size_t retry_wave_lds_bytes(int V, int ppl) {
  return ((sizeof(float) * kFastSeg * (size_t)V + sizeof(double) * ((size_t)V + 2) + 15) & ~(size_t)15) + sizeof(double) * 3 * retry_keep_cells(ppl) * 64;
}

template <typename IO, int PPL>
__device__ __forceinline__ bool retry_segment_f64(const ExactParams& p, unsigned char* wsmem, int b, int seg, int lane) {
  constexpr int NC = 2 * PPL, kSeg = kFastSeg, G = kRetryGroup, kKeep = NC > 8 ? NC : 8;
  const FastRetry& rt = p.retry;
  const int V = p.V, blank = p.blank, Tmax = p.T;
  const int T = (int)p.x_len[b], S = (int)p.t_len[b], L = 2 * S + 1;
  const int t0 = seg * kSeg, n = min(kSeg, T - t0);
  const bool cond = (T > 1 || L == 1);
  const double rr = (double)fast_tilt(S, T);
  IO* grads = reinterpret_cast<IO*>(p.grads) + (size_t)b * (size_t)Tmax * (size_t)V;
  float* ys = reinterpret_cast<float*>(wsmem);                                  // [16][V]
  double* post = reinterpret_cast<double*>(wsmem + ((sizeof(float) * kSeg * (size_t)V + 7) & ~(size_t)7));   // [V], then row sum, blank sum

  // this lane's pairs i = PPL*lane + r: label, skip permissions (ctc_loss.cpp:53-57, 91-96) -- as LaneCells::load
  int lab[PPL]; float skp[PPL], skn[PPL];      // (1 where the skip is allowed: its weight is r^2)
  const double rr2 = rr * rr;
  const int64_t* tg = p.targets + (int64_t)b * p.tgt_stride;
#pragma unroll
  for (int r = 0; r < PPL; r++) {
    const int i = PPL * lane + r;
    const int li = i < S ? (int)tg[i] : -1;
    const int lp_ = (i >= 1 && i - 1 < S) ? (int)tg[i - 1] : -1;
    const int ln = (i + 1 < S) ? (int)tg[i + 1] : -1;
    lab[r] = (i < S && li >= 0 && li < V) ? li : -1;
    skp[r] = (i < S && i >= 1 && li != blank && lp_ != li) ? 1.f : 0.f;
    skn[r] = (i + 1 < S && li != blank && ln != li) ? 1.f : 0.f;
  }
  {
    // (the segment's probabilities as F1 left them: [label][16 steps] for the small alphabets, [step][label] beyond)
    const float* src = rt.ytab_segments ? rt.ytab + ((size_t)b * rt.NS + seg) * V * kSeg : rt.ytab + ((size_t)b * Tmax + t0) * V;
    const int cnt = rt.ytab_segments ? kSeg * V : n * V;
    for (int i = lane; i < cnt; i += 64) ys[i] = src[i];
  }
  const int ys_t = rt.ytab_segments ? 1 : V, ys_v = rt.ytab_segments ? kSeg : 1;
  auto y = [&](int tt, int v) -> double { return v >= 0 ? (double)ys[tt * ys_t + v * ys_v] : 0.0; };
  // the exponents the chains had removed around this segment (see FastParams in ctc_loss_fast.hip): blocks i .. i+2 of 8
  // steps, i = t0 / 8 -- read once (a load per row on the rows' critical path cost more than the arithmetic)
  int cA[3], cB[3];
  {
    const int* cumA = rt.cumA + (size_t)b * rt.NB + (t0 >> 3);
    const int* cumB = rt.cumB + (size_t)b * rt.NB + (t0 >> 3);
#pragma unroll
    for (int k = 0; k < 3; k++) { cA[k] = cumA[k]; cB[k] = cumB[k]; }
  }
  auto cumA_at = [&](int blk) -> int { const int k = blk - (t0 >> 3); return k <= 0 ? cA[0] : k == 1 ? cA[1] : cA[2]; };
  auto cumB_at = [&](int blk) -> int { const int k = blk - (t0 >> 3); return k <= 0 ? cB[0] : k == 1 ? cB[1] : cB[2]; };
  // Every row must reproduce the chains' log Z: sum_j alpha_t[j] beta_t[j] = Z r^(L-1) 2^-(EA(t) + EB(t)), with EA / EB
  // the exponents the chains had removed by then.  It does not if the f32 checkpoints could not hold what mattered (their
  // cells share one exponent per lane: a cell 2^-149 below its lane's largest is stored as zero) -- then the exact kernel.
  // (in log2, split like the segment kernel's self-check: the integer part is compared exactly, the fraction through an f32
  //  logarithm of the row sum's mantissa -- a double-precision log per row was a fifth of the redo's instructions)
  const double logz = rt.logz[2 * b];
  const double z2 = (logz + (double)(L - 1) * log(rr)) * 1.4426950408889634074;
  const double z2i = floor(z2);
  const float z2f = (float)(z2 - z2i);
  const bool in_lattice = lane * NC < L;               // (the chains store no cells past the lattice's 2S+1)
  bool bad = false;

  // one alpha step into row t = t0 + tt (tilted cells, F1's units and rescales)
  auto alpha_step = [&](double (&a)[NC], int tt) {
    const int t = t0 + tt;
    const double yb = y(tt, blank);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NC; k++) a[k] = 0.0;
      if (lane == 0) { a[0] = cond ? yb : 0.0; a[1] = rr * y(0, lab[0]); }
    } else {
      double pl = lane_shift_up(a[NC - 1]);
#pragma unroll
      for (int r = 0; r < PPL; r++) {
        const double ob = a[2 * r], ol = a[2 * r + 1];
        a[2 * r] = (ob + rr * pl) * yb;
        a[2 * r + 1] = (ol + rr * ob + (skp[r] != 0.f ? rr2 : 0.0) * pl) * y(tt, lab[r]);
        pl = ol;
      }
    }
    if ((t & 7) == 7) {
      const int e = cumA_at((t >> 3) + 1) - cumA_at(t >> 3);
      if (e != 0) {
#pragma unroll
        for (int k = 0; k < NC; k++) a[k] = ldexp(a[k], -e);
      }
    }
  };

  // ---- forward sweep: the rows entering steps 0 (the checkpoint), 4, 8, 12 ----
  // kept rows 1..3 live in LDS as [row][cell][lane] (the checkpoint, row 0, is read again when its turn comes)
  double* keep = reinterpret_cast<double*>(wsmem + (((sizeof(float) * kSeg * (size_t)V + sizeof(double) * ((size_t)V + 2)) + 15) & ~(size_t)15));
  auto checkpoint = [&](double (&a)[NC]) {
    if (seg == 0) {
#pragma unroll
      for (int k = 0; k < NC; k++) a[k] = 0.0;
    } else {
      const float* src = rt.ckA + ((size_t)b * rt.NS + seg) * rt.CELLS + lane * NC;
      const int own = in_lattice ? rt.ckE[(((size_t)b * rt.NS + seg) * 2 + 0) * 64 + lane] : -30000;
#pragma unroll
      for (int k = 0; k < NC; k++) a[k] = own > -30000 ? ldexp((double)src[k], own) : 0.0;
    }
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    double a[NC];
    checkpoint(a);
#pragma unroll 1                     // (rolled: fully unrolled the redo was ~10 k instructions per segment and no longer fitted
                                     //  the instruction cache)
    for (int g = 1; g < kSeg / G; g++) {
#pragma unroll
      for (int i = 0; i < G; i++) if ((g - 1) * G + i < n) alpha_step(a, (g - 1) * G + i);
#pragma unroll
      for (int k = 0; k < NC; k++) keep[((g - 1) * kKeep + k) * 64 + lane] = a[k];
    }
  }
  // ---- beta back through the segment, four rows at a time ----
  double q[NC];
  const bool last_seg = (t0 + n == T);
  if (!last_seg) {
    const float* src = rt.ckQ + ((size_t)b * rt.NS + seg + 1) * rt.CELLS + lane * NC;
    const int own = in_lattice ? rt.ckE[(((size_t)b * rt.NS + seg + 1) * 2 + 1) * 64 + lane] : -30000;
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = own > -30000 ? ldexp((double)src[k], own) : 0.0;
  } else {
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = 0.0;
  }
#pragma unroll 1
  for (int g = kSeg / G - 1; g >= 0; g--) {
    if (g * G >= n) continue;
    double A[G][NC];
    {
      double a[NC];
      if (g == 0) checkpoint(a);
      else {
#pragma unroll
        for (int k = 0; k < NC; k++) a[k] = keep[((g - 1) * kKeep + k) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < G; i++) {
        if (g * G + i < n) alpha_step(a, g * G + i);
#pragma unroll
        for (int k = 0; k < NC; k++) A[i][k] = a[k];
      }
    }
#pragma unroll
    for (int i = G - 1; i >= 0; i--) {
      const int tt = g * G + i;
      if (tt >= n) continue;
      const int t = t0 + tt;
      for (int v = lane; v < V + 2; v += 64) post[v] = 0.0;
      double bs[NC];
      if (t == T - 1) {
#pragma unroll
        for (int r = 0; r < PPL; r++) {
          const int pi = PPL * lane + r;
          bs[2 * r] = (2 * pi == L - 1 && cond) ? 1.0 : 0.0;
          bs[2 * r + 1] = (2 * pi + 1 == L - 2) ? rr : 0.0;
        }
      } else {
        double nb = lane_shift_down(q[0]), nl = lane_shift_down(q[1]);
#pragma unroll
        for (int r = PPL - 1; r >= 0; r--) {
          bs[2 * r + 1] = q[2 * r + 1] + rr * nb + (skn[r] != 0.f ? rr2 : 0.0) * nl;
          bs[2 * r] = q[2 * r] + rr * q[2 * r + 1];
          nb = q[2 * r]; nl = q[2 * r + 1];
        }
      }
      double mine = 0.0, myblank = 0.0;
#pragma unroll
      for (int r = 0; r < PPL; r++) {
        const double pa = A[i][2 * r] * bs[2 * r];
        const double pb = A[i][2 * r + 1] * bs[2 * r + 1];
        myblank += pa; mine += pa + pb;
        if (lab[r] >= 0 && pb != 0.0) atomicAdd(&post[lab[r]], pb);
      }
      mine = wave_sum(mine); myblank = wave_sum(myblank);
      const double yb = y(tt, blank);
#pragma unroll
      for (int r = 0; r < PPL; r++) {
        q[2 * r] = bs[2 * r] * yb;
        q[2 * r + 1] = bs[2 * r + 1] * y(tt, lab[r]);
      }
      if ((t & 7) == 0) {
        const int e = cumB_at(t >> 3) - cumB_at((t >> 3) + 1);
        if (e != 0) {
#pragma unroll
          for (int k = 0; k < NC; k++) q[k] = ldexp(q[k], -e);
        }
      }
      // ---- the row: check against the chains' log Z, then y - posterior ----
      const double st = mine;
      // (a row sum near the end of f64 -- rows sink by up to ~110 bits per step between the chains' rescales, nine steps apart --
      //  has lost cells that matter to denormals, and its reciprocal overflows: until round 5 such rows passed the check below
      //  and were written as NaN (inf - inf in the Newton step) while the redo reported success.  They fail now: the
      //  extended-range redo, or the exact kernel, takes the utterance.)
      if (!(st > 0x1p-960) || !(st < __builtin_huge_val())) bad = true;
      {
        const int EA = cumA_at((t + 1) >> 3), EB = cumB_at((t + 8) >> 3);
        int ex;
        const double mant = frexp(st, &ex);                       // st = mant 2^ex, mant in [0.5, 1)
        const float dev = __builtin_amdgcn_logf((float)mant) - z2f + (float)((double)(ex + EA + EB) - z2i);
        // (a row sum off by more than 2e-6 relative means cells that mattered were stored with too few bits)
        if (!(fabsf(dev) <= 3e-6f)) bad = true;
      }
      double inv = __builtin_amdgcn_rcp(st);
      inv = fma(fma(-st, inv, 1.0), inv, inv);                    // (one Newton step: 2^-40 relative)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the atomics have landed: one wave, in-order LDS)
      for (int v = lane; v < V; v += 64) {
        double pv = post[v];
        if (v == blank) pv += myblank;
        grads[(size_t)t * V + v] = (IO)((y(tt, v) - pv * inv) * p.gscale);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  return !__any(bad);
}

#ifndef E2E_EXT_ON                   // (tools/diag: 0 compiles the extended-range redo out of the flagged kernel)
#define E2E_EXT_ON 1
#endif
#include "ctc_ext.h"

// One utterance b, with the alpha slab `slot` of the workspace.
// (forced inline: out of line the parameter block is handed over through scratch memory and every pointer in it becomes
// a generic one -- flat loads behind scratch loads)
// SCALED: the same lattice in the probability domain -- f64 cells, every row divided by a power of two taken from the row
// before it (the wave maxima ride on the step's own barrier), alpha rows in the same slab with their exponents beside
// them -- instead of the reference's log(1.0 + exp(x)) per addition: ~1.7 instead of ~7 us per frame.  Not the reference's
// arithmetic (results agree to ~1e-12 relative), so it only stands in where this kernel is the *fallback or the tail of
// an f32 path*: utterances the fast path hands over, the compact lattice of the wide path, targets beyond the fast
// kernels' width.  f64 inputs, E2E_ALGO_EXACT and anything whose scaled partition sum is not a positive finite number
// (infeasible, or probabilities below f64's range) take the log-domain code below.
template <typename IO, bool SCALED = false>
__device__ __forceinline__ void ctc_exact_one(const ExactParams& p, unsigned char* smem, int b, int slot) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int V = p.V, Tmax = p.T, Lmax = p.Lmax, Smax = p.Smax, blank = p.blank;

  double* buf0 = reinterpret_cast<double*>(smem);          // [Lmax]
  double* buf1 = buf0 + Lmax;                              // [Lmax]
  double* psorted = buf1 + Lmax;                           // [2][Smax] label-sorted posteriors (second half: scaled form)
  double* red = psorted + 2 * (Smax > 0 ? Smax : 1);       // [40]: sums; the scaled form's wave sums and maxima
  int* ext = reinterpret_cast<int*>(red + 40);             // [Lmax]
  int* rank = ext + Lmax;                                  // [Smax] rank of target i in label order
  int* sorted_lab = rank + (Smax > 0 ? Smax : 1);          // [Smax] label at sorted position r
  unsigned* is_label = reinterpret_cast<unsigned*>(sorted_lab + (Smax > 0 ? Smax : 1));   // [(V+31)/32] bit v: column v is a target label

  const IO* x = reinterpret_cast<const IO*>(p.x) + (int64_t)b * p.sB;
  IO* grads = reinterpret_cast<IO*>(p.grads) + (size_t)b * (size_t)Tmax * (size_t)V;
  typedef typename LossOf<IO>::type LT;
  LT* losses = reinterpret_cast<LT*>(p.losses);
  double* wa = p.ws_alpha + (size_t)slot * (size_t)Tmax * (size_t)Lmax;
  double* wl = p.ws_lse + (size_t)slot * (size_t)Tmax;

  if (p.mode == 2 && p.flags[b] == 0) return;
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  if (p.mode == 2 || Tq < 1 || Tq > Tmax || Sq < 0 || Sq > Smax) {   // invalid lengths: poison, do not crash
    const double qnan = __builtin_nan("");
    if (tid == 0) losses[b] = (LT)qnan;
    for (size_t i = tid; i < (size_t)Tmax * V; i += kThreads) grads[i] = (IO)qnan;
    return;
  }
  const int T = (int)Tq, S = (int)Sq, L = 2 * S + 1;

  // ---- P0: extended targets (ctc_loss.cpp:25-31), label order, row lse ----
  int bad_label = 0;
  for (int j = tid; j < L; j += kThreads) {
    int64_t lab = blank;
    if (j & 1) {
      lab = p.targets[(int64_t)b * p.tgt_stride + (j >> 1)];
      bad_label |= (lab < 0) | (lab >= V);
    }
    ext[j] = (int)lab;
  }
  for (int v = tid; v < (V + 31) / 32; v += kThreads) is_label[v] = 0u;
  if (tid == 0) red[8] = 0.0;
  if (__syncthreads_or(bad_label)) {
    // a target outside [0,V) would index the row and the per-label sums out of bounds (the reference reads garbage
    // there): poison like invalid lengths, do not crash
    const double qnan = __builtin_nan("");
    if (tid == 0) losses[b] = (LT)qnan;
    for (size_t i = tid; i < (size_t)Tmax * V; i += kThreads) grads[i] = (IO)qnan;
    __syncthreads();
    return;
  }
  // Too few frames for the targets -- T < S + (adjacent equal labels), with no target equal to the blank id -- has no
  // alignment whatever the emissions: the reference walks the whole lattice to find log Z = -inf, then its
  // exp(-inf - (-inf)) poisons the slab (Q2: loss +inf, grads NaN).  Same result here without the walk: a batch with one
  // such utterance cost the fast path's caller a full 7 ms recomputation per step.
  bool blank_valued_label = false;     // a target equal to the blank id (the reference shares the blank column, :109-113)
  {
    int need = 0, blank_label = 0;
    for (int i = tid; i < S; i += kThreads) {
      const int li = ext[2 * i + 1];
      need += 1 + ((i > 0 && ext[2 * i - 1] == li) ? 1 : 0);
      blank_label |= li == blank;
    }
    need = (int)wave_sum((double)need);
    if (lane == 0) red[wid] = (double)need;
    const int any_blank = __syncthreads_or(blank_label);
    int total = 0;
    for (int w2 = 0; w2 < kThreads / 64; w2++) total += (int)red[w2];
    __syncthreads();
    blank_valued_label = any_blank != 0;
    if (!any_blank && T < total) {
      const double qnan = __builtin_nan("");
      if (tid == 0) losses[b] = (LT)__builtin_huge_val();
      for (size_t i = tid; i < (size_t)Tmax * V; i += kThreads) grads[i] = (IO)qnan;
      return;
    }
  }
  // stable rank of target i among the S targets by label value (same-label cells keep increasing j,
  // so the per-label sums below run in the reference's order, ctc_loss.cpp:109-114)
  for (int i = tid; i < S; i += kThreads) {
    const int li = ext[2 * i + 1];
    int r = 0;
    for (int k = 0; k < S; k++) {
      const int lk = ext[2 * k + 1];
      r += (lk < li) || (lk == li && k < i);
    }
    rank[i] = r;
    sorted_lab[r] = li;
    if (li != blank) atomicOr(&is_label[li >> 5], 1u << (li & 31));       // (after the barrier above that cleared the map)
  }
  if (!p.logprobs) {
    for (int t = wid; t < Tmax; t += kThreads / 64) {
      const IO* row = x + (int64_t)t * p.sT;
      double m = neg_inf();
      for (int v = lane; v < V; v += 64) m = fmax(m, (double)row[(int64_t)v * p.sV]);
      m = wave_max(m);
      double s = 0.0;
      for (int v = lane; v < V; v += 64) s += exp((double)row[(int64_t)v * p.sV] - m);
      s = wave_sum(s);
      if (lane == 0) wl[t] = m + log(s);
    }
  }
  __syncthreads();
  // make the row-lse values written by other waves visible (global memory, same workgroup)
  __threadfence_block();

  auto lp = [&](int t, int v) -> double {
    double r = (double)x[(int64_t)t * p.sT + (int64_t)v * p.sV];
    return p.logprobs ? r : r - wl[t];
  };

  if (SCALED && L <= 2 * kThreads) {
    // A step is a chain -- barrier, LDS reads, a handful of f64 operations, LDS writes, the row maximum, barrier -- that every
    // wave walks alone, so it is written for latency: no branches around the cells (dead neighbours are multiplied by 0,
    // dead threads read a clamped address and do not write), all LDS reads of a step issued together with exp() of the
    // emissions in their shadow, reductions by DPP, the barrier waits for LDS only, and every global value a step needs is
    // asked for kAhead steps earlier (register ring; the sweeps are unrolled by kAhead).  Cells outside the reference's
    // [start, end) window are not forced to zero: above it they are zero by themselves, below it alpha is not but beta is
    // (and the other way round), so their posteriors are exact zeros all the same.
    constexpr int kAhead = 4;
    int* wexp = p.ws_exp + (size_t)slot * (size_t)Tmax;
    double* psort2 = psorted;                                // [2][S1] (the log-domain code uses the first half)
    double* redb = red + 10;                                 // [2][8] blank-cell sums of the waves
    int* rmax = reinterpret_cast<int*>(red + 26);            // [2][8] wave maxima (high words) of the row just written
    int* blank_is_label = reinterpret_cast<int*>(red + 34);
    const int S1 = Smax > 0 ? Smax : 1;
    const double gscale = p.gscale;
    const bool lpin = p.logprobs != 0;
    // two cells per thread, j = tid and tid + NT, on the first NT threads: half of the waves (one per SIMD, its two cells
    // interleaved) when that covers the row -- the other waves only keep the barriers (and the blank column) company
    const int NT = L > kThreads ? kThreads : kThreads / 2;
    const bool act = tid < NT;
    bool live[2]; int64_t off[2]; int rk[2];
    int jc[2], jm1[2], jm2[2], jp1[2], jp2[2];               // clamped LDS indices of the cell and its neighbours
    double f0[2], f1[2], f2[2], g1[2], g2[2], binit[2];       // 1.0 where that neighbour counts
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int j = tid + k * NT;
      live[k] = act && j < L;
      jc[k] = min(j, L - 1);
      const int c = ext[jc[k]];
      jm1[k] = max(jc[k] - 1, 0); jm2[k] = max(jc[k] - 2, 0); jp1[k] = min(jc[k] + 1, L - 1); jp2[k] = min(jc[k] + 2, L - 1);
      f0[k] = live[k] ? 1.0 : 0.0;
      f1[k] = (live[k] && j > 0) ? 1.0 : 0.0;
      f2[k] = (live[k] && c != blank && j >= 2 && ext[jm2[k]] != c) ? 1.0 : 0.0;
      g1[k] = (live[k] && j < L - 1) ? 1.0 : 0.0;
      g2[k] = (live[k] && c != blank && j + 2 < L && ext[jp2[k]] != c) ? 1.0 : 0.0;
      binit[k] = (live[k] && ((j == L - 1 && (T > 1 || L == 1)) || j == L - 2)) ? 1.0 : 0.0;
      off[k] = (int64_t)c * p.sV;
      rk[k] = (live[k] && (j & 1)) ? rank[j >> 1] : 0;
    }
    // this thread's column of the gradient row: its label (sorted position tid) if it is the first of its run; the last
    // thread has the blank column
    const bool leader = tid < S && (tid == 0 || sorted_lab[tid - 1] != sorted_lab[tid]);
    const int llab = leader ? sorted_lab[tid] : blank;
    int lrun = 0;
    if (leader) { lrun = 1; while (tid + lrun < S && sorted_lab[tid + lrun] == llab) lrun++; }
    if (tid == 0) *blank_is_label = 0;
    __syncthreads();
    if (leader && llab == blank) *blank_is_label = 1;
    __syncthreads();
    const bool blank_leader = leader && llab == blank;
    const bool blank_thread = tid == kThreads - 1 && !*blank_is_label;
    const bool hascol = leader || blank_thread;
    const int64_t coloff = (int64_t)llab * p.sV;
    auto row_exp = [&](int which) -> int {                   // exponent of the largest cell of that row (0 for an all-zero row)
      const int4 lo = *reinterpret_cast<const int4*>(rmax + which * 8), hi = *reinterpret_cast<const int4*>(rmax + which * 8 + 4);
      const int m = max(max(max(lo.x, lo.y), max(lo.z, lo.w)), max(max(hi.x, hi.y), max(hi.z, hi.w)));
      return m > 0 ? ((m >> 20) & 0x7ff) - 1023 : 0;         // (non-negative doubles order like their high words)
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- alpha ----
    struct ASlot { IO xr[2]; double rl; };
    ASlot ar[kAhead];
    // (fetches and steps are unconditional -- rows clamped, the steps past the end with their stores switched off: a path
    //  through the loop that issues fewer loads makes the compiler's wait-count bookkeeping drain the ring at every use)
    auto afetch = [&](ASlot& q, int t) {
      t = min(t, T - 1);
      const IO* row = x + (int64_t)t * p.sT;
      q.xr[0] = row[off[0]]; q.xr[1] = row[off[1]];
      q.rl = lpin ? 0.0 : wl[t];
    };
    int E = 0;
    auto astep = [&](int t, ASlot& q) {
      const bool valid = t < T;
      const ASlot c = q;
      afetch(q, t + kAhead);
      if (act) {
        const double* prev = (t & 1) ? buf0 : buf1;
        double* cur = (t & 1) ? buf1 : buf0;
        double* warow = wa + (size_t)t * Lmax;
        const int e = row_exp((t - 1) & 1);
        const double p00 = prev[jc[0]], p10 = prev[jm1[0]], p20 = prev[jm2[0]];
        const double p01 = prev[jc[1]], p11 = prev[jm1[1]], p21 = prev[jm2[1]];
        const double y0 = exp((double)c.xr[0] - c.rl), y1 = exp((double)c.xr[1] - c.rl);
        E += valid ? e : 0;
        const double a0 = ldexp((p00 * f0[0] + p10 * f1[0] + p20 * f2[0]) * y0, -e);
        const double a1 = ldexp((p01 * f0[1] + p11 * f1[1] + p21 * f2[1]) * y1, -e);
        if (live[0] && valid) { cur[tid] = a0; warow[tid] = a0; }
        if (live[1] && valid) { cur[tid + NT] = a1; warow[tid + NT] = a1; }
        // the row's scale comes from the cells INSIDE the reference's window only (here: cells that can still reach the end,
        // j >= L - 2 (T - t)).  A doomed cell below it -- the all-blank path at j = 0 when T is barely long enough -- may be
        // far larger than everything feasible; scaled by it, feasible cells would lose bits or flush to zero mid-sequence.
        // Doomed mass never enters the window; if it overflows, the NaN it leaves ends in the log-domain walk below.
        const int wlo = L - 2 * (T - t);
        int mt = max((live[0] && tid >= wlo) ? __double2hiint(a0) : 0, (live[1] && tid + NT >= wlo) ? __double2hiint(a1) : 0);
        mt = wave_max_nonneg_lane63(mt);
        if (lane == 63 && valid) rmax[(t & 1) * 8 + wid] = mt;
      }
      if (tid == 0 && valid) wexp[t] = E;
      lds_barrier();
    };
    {
      int mt = 0;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int j = tid + k * NT;
        if (live[k]) {
          double a = 0.0;
          if ((j == 0 && (T > 1 || L == 1)) || j == 1) a = exp(lp(0, ext[j]));
          buf0[j] = a; wa[j] = a;
          if (j >= L - 2 * T) mt = max(mt, __double2hiint(a));
        }
      }
      mt = wave_max_nonneg_lane63(mt);
      if (lane == 63) { rmax[wid] = mt; rmax[8 + wid] = 0; redb[wid] = 0.0; redb[8 + wid] = 0.0; }   // (waves without cells never write these again)
      if (tid == 0) wexp[0] = 0;
#pragma unroll
      for (int i = 0; i < kAhead; i++) afetch(ar[(1 + i) % kAhead], 1 + i);
      lds_barrier();
    }
    for (int t = 1; t < T; t += kAhead) { astep(t, ar[1]); astep(t + 1, ar[2]); astep(t + 2, ar[3]); astep(t + 3, ar[0]); }
    const double* last = ((T - 1) & 1) ? buf1 : buf0;
    const double z = (L > 1) ? last[L - 1] + last[L - 2] : last[L - 1];
    __syncthreads();                                          // (also: wexp[] and the alpha rows are in memory)
    bool ok = z > 1e-250 && z < __builtin_huge_val();
    if (ok) {
      const double logZ = log(z) + (double)E * 0.693147180559945309417;
      const int ET = E;
      const double invz = 1.0 / z;
      // ---- beta, posteriors, the label and blank columns of the gradient: row t leaves in the step that makes beta's row t ----
      struct BSlot { IO xr[2]; IO xcol; double rl; double war[2]; int ex; };
      BSlot br[kAhead];
      auto bfetch = [&](BSlot& q, int t) {
        t = max(t, 0);
        const IO* row = x + (int64_t)t * p.sT;
        const double* warow = wa + (size_t)t * Lmax;
        q.xr[0] = row[off[0]]; q.war[0] = warow[jc[0]];
        q.xr[1] = row[off[1]]; q.war[1] = warow[jc[1]];
        q.xcol = row[coloff];
        q.rl = lpin ? 0.0 : wl[t];
        q.ex = wexp[t];
      };
      int F = 0;                                              // exponent taken out of the beta row t+1 (with its emissions)
      int bad = 0;
      auto bstep = [&](int t, BSlot& q) {
        const bool valid = t >= 0;
        const BSlot c = q;
        bfetch(q, t - kAhead);
        double* ps = psort2 + (t & 1) * S1;
        const double xcol = (double)c.xcol - c.rl;
        double ycol = 0.0;
        if (act) {
          const double* be_next = (t & 1) ? buf0 : buf1;
          double* be_cur = (t & 1) ? buf1 : buf0;
          const bool first = t == T - 1;
          const int e = first ? 0 : row_exp((t + 1) & 1);
          const double b00 = be_next[jc[0]], b10 = be_next[jp1[0]], b20 = be_next[jp2[0]];
          const double b01 = be_next[jc[1]], b11 = be_next[jp1[1]], b21 = be_next[jp2[1]];
          const double y0 = exp((double)c.xr[0] - c.rl), y1 = exp((double)c.xr[1] - c.rl);
          ycol = exp(xcol);
          // alpha_t beta_t / Z in the rows' units: 2^(E_t + F - E_T) / z
          const double unit = ldexp(invz, c.ex + F - ET);
          const double war0 = c.war[0] * unit, war1 = c.war[1] * unit;
          const double bt0 = first ? binit[0] : b00 * f0[0] + b10 * g1[0] + b20 * g2[0];
          const double bt1 = first ? binit[1] : b01 * f0[1] + b11 * g1[1] + b21 * g2[1];
          const double bc0 = ldexp(bt0 * y0, -e), bc1 = ldexp(bt1 * y1, -e);
          const double pj0 = war0 * bt0, pj1 = war1 * bt1;
          bad |= valid & (!(pj0 <= 2.0) | !(pj1 <= 2.0));
          double blank_part = 0.0;
          if (live[0] && valid) {
            be_cur[tid] = bc0;
            if (tid & 1) ps[rk[0]] = pj0; else blank_part = pj0;
          }
          if (live[1] && valid) {
            be_cur[tid + NT] = bc1;
            if (tid & 1) ps[rk[1]] = pj1; else blank_part += pj1;          // (NT is even: same parity as tid)
          }
          F += e;
          const int whi = 2 * t + 2;                          // (beta's side of the window: cells the start can reach, j < 2t + 2)
          int mt = max((live[0] && tid < whi) ? __double2hiint(bc0) : 0, (live[1] && tid + NT < whi) ? __double2hiint(bc1) : 0);
          mt = wave_max_nonneg_lane63(mt);
          blank_part = wave_sum_lane63(blank_part);
          if (lane == 63 && valid) { redb[(t & 1) * 8 + wid] = blank_part; rmax[(t & 1) * 8 + wid] = mt; }
        }
        lds_barrier();
        if (hascol && valid) {
          double s2 = 0.0;
          if (leader) {
            // the run's posteriors in increasing-j order (ctc_loss.cpp:109-114), four LDS reads in flight at a time
            s2 = ps[tid];
            int q2 = 1;
            for (; q2 + 3 < lrun; q2 += 4) {
              const double u0 = ps[tid + q2], u1 = ps[tid + q2 + 1], u2 = ps[tid + q2 + 2], u3 = ps[tid + q2 + 3];
              s2 = (((s2 + u0) + u1) + u2) + u3;
            }
            for (; q2 < lrun; q2++) s2 += ps[tid + q2];
          }
          if (blank_leader || blank_thread) {
            // (a target equal to the blank id shares the blank column, ctc_loss.cpp:109-113)
            const double* rb = redb + (t & 1) * 8;
            const double2 r0 = *reinterpret_cast<const double2*>(rb), r1 = *reinterpret_cast<const double2*>(rb + 2),
                          r2 = *reinterpret_cast<const double2*>(rb + 4), r3 = *reinterpret_cast<const double2*>(rb + 6);
            s2 += ((((((r0.x + r0.y) + r1.x) + r1.y) + r2.x) + r2.y) + r3.x) + r3.y;
          }
          if (!act) ycol = exp(xcol);                       // (the blank column's thread when its wave has no cells)
          grads[(size_t)t * V + llab] = (IO)((ycol - s2) * gscale);
        }
        // (no barrier here: the next step writes the other halves of ps / redb / rmax and the beta buffer this one read
        //  before its barrier.  Folding this column pass into the next step's stretch of code -- one LDS round trip for
        //  both -- measured no faster: 1.90 against 1.94 ms on 24 handed-over C2 utterances, 3.67 against 3.28 ms at S=500.)
      };
#pragma unroll
      for (int i = 0; i < kAhead; i++) bfetch(br[i], T - 1 - i);
      for (int t = T - 1; t >= 0; t -= kAhead) { bstep(t, br[0]); bstep(t - 1, br[1]); bstep(t - 2, br[2]); bstep(t - 3, br[3]); }
      ok = !__syncthreads_or(bad);
      if (ok) {
        if (tid == 0) losses[b] = (LT)(-logZ);
        // columns that are neither a label nor the blank: y itself, every row at once (nothing of the lattice in them)
        for (int t = wid; t < T; t += kThreads / 64) {
          const IO* row = x + (int64_t)t * p.sT;
          const double rl = lpin ? 0.0 : wl[t];
          for (int v = lane; v < V; v += 64) {
            if (v == blank || ((is_label[v >> 5] >> (v & 31)) & 1u)) continue;
            grads[(size_t)t * V + v] = (IO)(exp((double)row[(int64_t)v * p.sV] - rl) * gscale);
          }
        }
        for (size_t i = (size_t)T * V + tid; i < (size_t)Tmax * V; i += kThreads) {
          const int t = (int)(i / V), v = (int)(i % V);
          grads[i] = lpin ? (IO)(exp((double)x[(int64_t)t * p.sT + (int64_t)v * p.sV]) * gscale) : (IO)0;
        }
        return;
      }
    }
    // (no positive finite partition sum -- no alignment, or probabilities beyond f64's range -- or a posterior that is not
    //  a number: the log-domain walk below decides, and rewrites everything)
  }
  // ---- P1: alpha sweep, ctc_loss.cpp:33-61 ----
  for (int j = tid; j < L; j += kThreads) {
    double a = neg_inf();
    if (j == 0 && (T > 1 || L == 1)) a = lp(0, ext[0]);
    if (j == 1) a = lp(0, ext[1]);
    buf0[j] = a;
    wa[j] = a;
  }
  __syncthreads();
  for (int t = 1; t < T; t++) {
    const double* prev = (t & 1) ? buf0 : buf1;
    double* cur = (t & 1) ? buf1 : buf0;
    const int start = max(0, L - 2 * (T - t)), end = min(2 * t + 2, L);
    double* warow = wa + (size_t)t * Lmax;
    for (int j = tid; j < L; j += kThreads) {
      double a = neg_inf();
      if (j >= start && j < end) {
        const int cl = ext[j];
        a = prev[j];
        if (j > 0) {
          a = lse2(a, prev[j - 1]);
          if (cl != blank && j >= 2 && ext[j - 2] != cl) a = lse2(a, prev[j - 2]);
        }
        a += lp(t, cl);
      }
      cur[j] = a;
      warow[j] = a;
    }
    __syncthreads();
  }

  // ---- P2: loss, ctc_loss.cpp:63-70 ----
  const double* last = ((T - 1) & 1) ? buf1 : buf0;
  const double logZ = (L > 1) ? lse2(last[L - 1], last[L - 2]) : last[L - 1];
  if (tid == 0) losses[b] = (LT)(-logZ);
  __syncthreads();

  const double qnan = __builtin_nan("");
  const bool infeasible = logZ == neg_inf();
  if (infeasible && !blank_valued_label) {
    // infeasible alignment: the reference's exp(-inf - (-inf)) poisons the whole slab (Q2)
    for (size_t i = tid; i < (size_t)Tmax * V; i += kThreads) grads[i] = (IO)qnan;
    return;
  }
  // (infeasible WITH a blank-valued target: the reference's alpha refuses the skip into such a label, its beta takes it, so
  //  cells can have a finite alpha + beta although log Z is -inf; exp(log_post - logZ) is then +inf for a column with such a
  //  cell and NaN for the others.  The sweep below runs with "is the cell finite" in place of the posterior.)

  // ---- P3: beta sweep (ctc_loss.cpp:72-100) fused with the gradient (:102-117) ----
  // be[j] holds beta[j][t+1] + lp[t+1][ext[j]], the quantity the three-way sum reads.
  for (int t = T - 1; t >= 0; t--) {
    const double* be_next = (t & 1) ? buf0 : buf1;   // written at step t+1
    double* be_cur = (t & 1) ? buf1 : buf0;
    const int start = max(0, L - 2 * (T - t)), end = min(2 * t + 2, L);
    const double* warow = wa + (size_t)t * Lmax;
    double blank_part = 0.0;
    for (int j = tid; j < L; j += kThreads) {
      const int cl = ext[j];
      double bt = neg_inf();
      if (t == T - 1) {
        if (j == L - 1 && (T > 1 || L == 1)) bt = 0.0;
        if (j == L - 2) bt = 0.0;
      } else if (j >= start && j < end) {
        bt = be_next[j];
        if (j < L - 1) {
          bt = lse2(bt, be_next[j + 1]);
          if (cl != blank && j + 2 < L && ext[j + 2] != cl) bt = lse2(bt, be_next[j + 2]);
        }
      }
      be_cur[j] = bt + lp(t, cl);
      // posterior of cell (j,t); exp(-inf)=0   (infeasible: 1 for a finite cell)
      const double pj = infeasible ? ((warow[j] + bt > neg_inf()) ? 1.0 : 0.0) : exp(warow[j] + bt - logZ);
      if (j & 1) psorted[rank[j >> 1]] = pj; else blank_part += pj;
    }
    blank_part = wave_sum(blank_part);
    if (lane == 0) red[wid] = blank_part;
    __syncthreads();
    // per-label sums in increasing-j order
    for (int r = tid; r < S; r += kThreads) {
      const int lab = sorted_lab[r];
      if (r == 0 || sorted_lab[r - 1] != lab) {
        double s = psorted[r];
        for (int q = r + 1; q < S && sorted_lab[q] == lab; q++) s += psorted[q];
        // a target equal to the blank id shares the blank column (ctc_loss.cpp:109-113 keys on the label)
        if (lab != blank) {
          // the label's column of the gradient row, written here by the thread that holds its sum (the dense pass below
          // skips the columns that are labels)
          const double rl_ = p.logprobs ? 0.0 : wl[t];
          if (infeasible) s = s > 0.0 ? __builtin_huge_val() : qnan;
          grads[(size_t)t * V + lab] = (IO)((exp((double)x[(int64_t)t * p.sT + (int64_t)lab * p.sV] - rl_) - s) * p.gscale);
        } else red[8] = s;
      }
    }
    if (tid == kThreads - 1) {
      double s = 0.0;
      for (int w = 0; w < kThreads / 64; w++) s += red[w];
      red[9] = s;
    }
    __syncthreads();
    {
      IO* grow = grads + (size_t)t * V;
      const IO* xrow = x + (int64_t)t * p.sT;
      const double rl = p.logprobs ? 0.0 : wl[t];
      for (int v = tid; v < V; v += kThreads) {
        // blank column = even cells (+ target cells whose label equals the blank id, red[8]); every other column that is
        // not a label has posterior 0
        if ((is_label[v >> 5] >> (v & 31)) & 1u) continue;
        double pv = (v == blank) ? red[9] + red[8] : 0.0;
        if (infeasible) pv = pv > 0.0 ? __builtin_huge_val() : qnan;
        grow[v] = (IO)((exp((double)xrow[(int64_t)v * p.sV] - rl) - pv) * p.gscale);
      }
    }
    // next iteration's writes to psorted/red happen after its own first barrier or touch
    // buffers nobody reads here (be_cur of t-1 is be_next of t, last read before the barrier above)
    __syncthreads();
  }

  // ---- padded frames t >= T: exp(lp) in log-prob mode (Q1), 0 for fused logits; infeasible: NaN (-inf - -inf) ----
  for (size_t i = (size_t)T * V + tid; i < (size_t)Tmax * V; i += kThreads) {
    const int t = (int)(i / V), v = (int)(i % V);
    grads[i] = infeasible ? (IO)qnan : p.logprobs ? (IO)(exp((double)x[(int64_t)t * p.sT + (int64_t)v * p.sV]) * p.gscale) : (IO)0;
  }
}

constexpr int kFlagCache = 2048;    // utterances whose flag words and segment counts a workgroup keeps in LDS
constexpr int kRedoFailed = 512;    // flag bit set by the segment redo
#ifndef E2E_EXT_NOSPLIT             // (tools/diag A/B: 1 = both directions of a flagged utterance always on one workgroup)
#define E2E_EXT_NOSPLIT 0
#endif

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
  return v;
}

// Every utterance (mode 0): workgroups stride over the batch, one alpha slab each.
// Flagged modes (the tail of every fast-path call): in the usual case of nothing flagged the launch only writes the
// optional sum / mean of the losses.  Otherwise (mode 1)
//   1. utterances flagged for range only (bits 8 / 16) have their segments redone in f64, one wave per segment, spread over
//      all waves of the launch (retry_segment_f64);
//   2. utterances flagged for anything else are recomputed by the reference's arithmetic, one workgroup each, on the
//      first `nslabs` workgroups (the alpha slabs of the workspace: 24 at most -- 77 MB at B=256, T=1000, S=200 -- instead
//      of one per workgroup);
//   3. the last workgroup to finish recomputes what step 1 could not settle (bit 512; rare) and writes the reduction.
template <typename IO, bool SCALED>
__global__ __launch_bounds__(kThreads) void ctc_exact_all_kernel(ExactParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    ctc_exact_one<IO, SCALED>(p, smem, b, blockIdx.x);
    __syncthreads();                       // LDS is reused by the next utterance
  }
}

// P8: the instance whose f64 redo of single segments is the one for eight label pairs per segment-kernel lane (targets of
// 256..447 labels, alphabets beyond 224 columns) -- an instance of its own because that redo needs ~250 registers and
// costs the others their allocation (the headline's fallback regime: 0.205 -> 0.217 ms with all four widths in one kernel)
template <typename IO, bool SCALED, bool P8 = false>
__global__ __launch_bounds__(kThreads) void ctc_exact_kernel(ExactParams p_in) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // the whole flag vector in ONE round trip per workgroup (every workgroup looks at all of it, so that all of them agree
  // on what is flagged); the flag words are kept in LDS for the walks below
  __shared__ int any;
  __shared__ unsigned short s_flag[kFlagCache];
  if (tid == 0) any = 0;
  __syncthreads();
  for (int b = tid; b < p_in.B; b += kThreads) {
    const int f = p_in.flags[b] & (kRedoFailed - 1);
    if (f != 0) any = 1;
    if (b < kFlagCache) s_flag[b] = (unsigned short)f;
  }
  __syncthreads();
  const bool reduce = p_in.reduced && p_in.reduction != E2E_REDUCE_NONE;
  auto write_reduction = [&](bool coherent) {
    if (reduce && tid < 64) {
      typedef typename LossOf<IO>::type LT;
      const LT* losses = reinterpret_cast<const LT*>(p_in.losses);
      double s = 0.0;
      for (int b = tid; b < p_in.B; b += 64)
        s += coherent ? (double)__hip_atomic_load(&losses[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (double)losses[b];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (tid == 0) *reinterpret_cast<LT*>(p_in.reduced) = (LT)(p_in.reduction == E2E_REDUCE_MEAN ? s / (double)p_in.B : s);
    }
  };
  if (!any) {
    // nothing flagged: the losses the fast path wrote are final; workgroup 0 writes their sum / mean (fixed order)
    if (blockIdx.x == 0) write_reduction(false);
    return;
  }
  // From here on the parameter block is read through a pointer the compiler cannot trace back to the kernel's arguments.  A load
  // of a kernel argument is always safe, so it hoists ALL of them -- for every path below -- into the entry block: 23 scalar
  // loads in dependent batches and, for want of SGPRs, 75 writes to VGPR lanes, in front of the test above that ends the launch in
  // the usual case (300 instructions up to the first s_endpgm; 198 now, with 12 loads).  Headline call 126.05 -> 125.5 us in one
  // process; the flagged regimes gain 1-2 % as well (fewer live values across everything below).
  typedef const ExactParams __attribute__((address_space(4))) KParams;
  KParams* pk = (KParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pk));
  const ExactParams& p = *(const ExactParams*)pk;
  // (diagnostics, flagged calls only: workgroup 0 leaves the 100 MHz clock at the end of every phase in ctl[16 ..], two ints each)
  auto stamp = [&](int k) {
    if (blockIdx.x == 0 && tid == 0) {
      const unsigned long long t = wall_clock64();
      p.ctl[16 + 2 * k] = (int)(unsigned)t; p.ctl[17 + 2 * k] = (int)(unsigned)(t >> 32);
    }
  };
  stamp(0);
  auto flag_of = [&](int b) -> int { return b < kFlagCache ? (int)s_flag[b] : (p.flags[b] & (kRedoFailed - 1)); };
  // (any_ext0, set below: some utterance of the call goes to the extended-range redo for its chains' sake -- then the ones whose
  //  f32 segment rows alone gave out (8 / 16) join it instead of the f64 redo of single segments: the chains' 0.25 ms is paid
  //  anyway, theirs run beside it, and nobody has to wait to learn which redos failed)
  bool any_ext0 = false;
  auto range_only = [&](int f) -> bool { return p.mode == 1 && p.has_retry && !any_ext0 && f != 0 && (f & ~(8 | 16)) == 0; };
  auto range_bits_only = [&](int f) -> bool { return f != 0 && (f & ~(8 | 16)) == 0; };
  // flagged for its numbers, not for its inputs (1: lengths, 2: blank inside the targets, 128: protocol, 256: probabilities the f32
  // table cannot hold; 64 alone -- probabilities below 2^-100 but still normal f32 numbers -- is a matter of range)
  auto ext_candidate = [&](int f) -> bool {
    return E2E_EXT_ON && p.mode == 1 && p.has_ext && ((f & (4 | 32 | 64)) != 0 || (!p.has_retry && (f & (8 | 16)) != 0)) && (f & (1 | 2 | 128 | 256)) == 0;
  };

  bool last_by_arrival = false;       // (set where step 1's bounded wait already told which workgroup is the last: see step 3)
  // (this workgroup wrote what the last workgroup reads or may write again: losses, for its reduction; rows of an f64 segment redo
  //  that failed, when no wait-and-release follows step 1 -- the last workgroup recomputes such an utterance, and rows left dirty in
  //  another XCD's L2 would be written back over its own.  It owes a release before it takes its ticket: see step 3)
  bool owes_release = false;
  if (p.mode == 2) {
    owes_release = true;
    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
      ctc_exact_one<IO, false>(p, smem, b, 0);      // (poisons flagged utterances, touches no slab)
      __syncthreads();
    }
  } else {
    // ---- X. extended-range redo (ctc_ext.h) of what is flagged for its NUMBERS.  Round 0, known from the start: alpha / beta
    //      log Z mismatch (32), a partition sum out of range (4), probabilities below 2^-100 (64).  Round 1, known once every
    //      workgroup has reported the end of step 1: what the f64 redo of single segments could not settle (512).
    //      Chains: utterance i of a round's list on workgroup i mod grid; then every workgroup takes segments. ----
    __shared__ int s_dec, s_next, s_go, s_lastarr;
    __shared__ int s_xb[kExtMaxList];                   // a round's list: utterance numbers ...
    __shared__ int s_xoff[kExtMaxList + 1];             // ... and the running count of their segments
    // the list of a round, in utterance order, the same in every workgroup (s_flag holds what the round looks at)
    auto build_list = [&](int round) {
      if (wid == 0) {
        int nx = 0, off = 0;
        for (int c0 = 0; c0 < p.B && nx < kExtMaxList; c0 += 64) {
          const int bb = c0 + lane;
          int f = 0;
          if (bb < p.B) f = bb < kFlagCache ? (int)s_flag[bb] : (__hip_atomic_load(&p.flags[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (2 * kRedoFailed - 1));
          const bool take = bb < p.B && (round == 0 ? (ext_candidate(f) || (any_ext0 && range_bits_only(f)))
                                                    : ((f & kRedoFailed) != 0 && range_only(f & (kRedoFailed - 1))));
          const int nseg = take ? ((int)p.x_len[bb] + kFastSeg - 1) / kFastSeg : 0;
          const unsigned long long m = __ballot(take);
          const int before = __builtin_popcountll(m & ((1ull << lane) - 1));
          const int incl = wave_incl_scan(nseg, lane);
          const int cnt = __builtin_popcountll(m);
          if (nx + cnt > kExtMaxList) break;              // (a list that overflows is cut at a chunk boundary: the rest is left to step 3)
          if (take) { s_xb[nx + before] = bb; s_xoff[nx + before] = off + incl - nseg; }
          nx += cnt; off += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane == 0) { s_next = nx; s_xoff[nx] = off; }
      }
      __syncthreads();
    };
    auto ext_segments_of_list = [&](int nx) {
      // A contiguous run of the list's items per workgroup (item j = segment j - s_xoff[i] of the list's i-th utterance): mostly
      // segments of ONE utterance, whose flag is waited for, whose Z / lengths / labels are fetched and behind whose flag the
      // acquire fence is paid once per run instead of once per item (ext_segment's header says what an item cost).
      const int nitems = s_xoff[nx];
      const int G = (int)gridDim.x;
      const int j0 = (int)((long long)blockIdx.x * nitems / G), j1 = (int)((long long)(blockIdx.x + 1) * nitems / G);
      if (j0 >= j1) return;
      {
        double* post = reinterpret_cast<double*>(smem);              // ext_segment's rows: all zero between items
        for (int i = tid; i < 16 * (p.V + 1); i += kThreads) post[i] = 0.0;
      }
      int lo = 0, hi = nx - 1;                                      // the utterance of item j0: last i with s_xoff[i] <= j0
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_xoff[mid] <= j0) lo = mid; else hi = mid - 1; }
      XSegUtt utt; utt.b = -1;
      bool go = false;
      for (int j = j0; j < j1; j++) {
        while (lo + 1 < nx && s_xoff[lo + 1] <= j) lo++;
        const int ub = s_xb[lo], seg = j - s_xoff[lo];
        if (ub != utt.b) {
          if (tid == 0) {
            int f, spins = 0;
            while (((f = __hip_atomic_load(&p.flags[ub], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & kExtDone) == 0 && ++spins < (1 << 16))
              __builtin_amdgcn_s_sleep(8);
            if ((f & kExtDone) == 0) {
              // patience ran out: the utterance is left to step 3 -- but only if its chains have STILL not reported.  kExtBad goes in
              // by compare-and-swap against a word without kExtDone, so that it can never follow kExtDone: rows that a segment wrote
              // behind a clean kExtDone are then final, nobody rewrites them, and they need no release (with every
              // workgroup releasing behind its segments the launch's end was 256 L2 write-backs at once, ~25 us of label_noise)
              for (;;) {
                if (f & kExtDone) break;                            // (they made it after all)
                const int old = atomicCAS(&p.flags[ub], f, f | kExtBad);
                if (old == f) { f |= kExtBad; atomicAdd(&p.ctl[4], 1); break; }
                f = old;
              }
            }
            s_go = (f & kExtDone) != 0 && (f & kExtBad) == 0;
          }
          __syncthreads();
          go = s_go != 0;
          if (go) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ext_segment_setup<IO>(p, ub, utt);
          } else utt.b = ub;
        }
        __syncthreads();                 // (s_go has been read by everybody; the rows of the item before are cleared)
        if (go) ext_segment<IO>(p, utt, seg);       // (rows behind a clean kExtDone are final: see the wait above)
      }
      __syncthreads();
    };
    if (E2E_EXT_ON && p.has_ext) {
      bool mine = false;
      for (int b = tid; b < p.B; b += kThreads) mine |= ext_candidate(flag_of(b));
      any_ext0 = __syncthreads_or(mine ? 1 : 0) != 0;
    }
    // ---- 1. segments of the utterances flagged for range only: item i of the running list goes to wave i mod NW.  (Only when no
    //      utterance needs the extended-range chains: otherwise the range-flagged ones join those, and this step has nothing to do.) ----
    const int NW = (int)gridDim.x * p.redo_waves;
    const int me = (int)blockIdx.x * p.redo_waves + wid;
    unsigned char* wsmem = smem + (size_t)wid * retry_wave_lds_bytes(p.V, p.retry.PPL);
    int base = 0;
    bool failed_here = false, wrote_here = false;
    for (int c0 = 0; c0 < p.B && wid < p.redo_waves; c0 += 64) {
      const int bb = c0 + lane;
      const int f = bb < p.B ? flag_of(bb) : 0;
      // (only the segments that failed in the segment kernel: the others' rows passed their self-check and stay)
      int ns = 0;
      if (range_only(f)) {
        const unsigned* mk = p.retry.segmask + (size_t)bb * p.retry.MW;
        for (int i = 0; i < p.retry.MW; i++) ns += __builtin_popcount(mk[i]);
      }
      const int incl = wave_incl_scan(ns, lane);
      const int tot = __builtin_amdgcn_readlane(incl, 63);
      if (tot != 0) {
        const int start = base + incl - ns;
        int first = (me - start) % NW; if (first < 0) first += NW;       // my first segment of this utterance, if < ns
        unsigned long long work = __ballot(first < ns);
        while (work) {
          const int l = __builtin_ctzll(work); work &= work - 1;
          const int ub = c0 + l;
          const int sfirst = __builtin_amdgcn_readlane(first, l), uns = __builtin_amdgcn_readlane(ns, l);
          for (int item = sfirst; item < uns; item += NW) {
            // the item-th failed segment of the utterance
            int seg = -1;
            {
              const unsigned* mk = p.retry.segmask + (size_t)ub * p.retry.MW;
              int left = item;
              for (int i = 0; i < p.retry.MW && seg < 0; i++) {
                unsigned m = mk[i];
                const int c = __builtin_popcount(m);
                if (left >= c) { left -= c; continue; }
                for (; left > 0; left--) m &= m - 1;
                seg = 32 * i + __builtin_ctz(m);
              }
            }
            if (seg < 0) continue;
            bool ok;
            if constexpr (P8) ok = retry_segment_f64<IO, 8>(p, wsmem, ub, seg, lane);
            else if (p.retry.PPL == 1) ok = retry_segment_f64<IO, 1>(p, wsmem, ub, seg, lane);
            else if (p.retry.PPL == 2) ok = retry_segment_f64<IO, 2>(p, wsmem, ub, seg, lane);
            else ok = retry_segment_f64<IO, 4>(p, wsmem, ub, seg, lane);
            if (!ok) { failed_here = true; if (lane == 0) { atomicOr(&p.flags[ub], kRedoFailed); atomicAdd(&p.ctl[5], 1); } }
            else wrote_here = true;
          }
        }
      }
      base += tot;
    }
    const bool wg_failed = __syncthreads_or(failed_here ? 1 : 0) != 0;      // (and: the waves' LDS is taken over by what follows)
    if (wg_failed) owes_release = true;
    // Rows of a successful redo have no reader inside the launch -- unless ANOTHER segment of the same utterance failed and nothing
    // but step 3 is left to settle it (no extended-range redo in this launch, or its wait ran out: below): then the last workgroup
    // rewrites those rows from its own XCD, and the first writer's dirty lines must have left this one's L2 by then.
    if (wrote_here && !(E2E_EXT_ON && p.has_ext)) owes_release = true;
    stamp(1);
    if (E2E_EXT_ON && p.has_ext) {
      // Which round?  0 (known from the start): some utterance needs the chains -- alpha / beta log Z mismatch, a partition sum out
      // of range, probabilities below 2^-100 -- and everything flagged for its numbers goes with it.  1: step 1 ran, and some redo
      // could not settle its utterance.  To learn that, every workgroup has to have finished step 1: a bounded wait (a grid
      // that is not resident at once -- a partitioned or shared GPU -- must not hang).  A workgroup arrives with ONE atomic that
      // carries its own outcome (ctl[2]: arrivals in the low 16 bits, workgroups with a failed redo above), and all workgroups
      // adopt ONE decision (ctl[3]): 1 = everybody arrived and some redo failed, 2 = some wait ran out (then what step 1 could not
      // settle is left to step 3), 3 = everybody arrived, nothing failed (the usual case).  No wait when step 1 had nothing to do.
      // (No release before the arrival: nothing a workgroup wrote in step 1 is read by another one, and the write-back of the
      //  L2 -- 256 of them at once -- was 17 us of the usual case.  Round 1 REWRITES rows that step 1 wrote, possibly from another
      //  XCD: there, and only there, every workgroup releases first, and the chains start when all have: see below.)
      int round = any_ext0 ? 0 : -1;
      if (!any_ext0) {
        bool any_range = false;
        for (int b = tid; b < p.B; b += kThreads) any_range |= range_only(flag_of(b));
        if (__syncthreads_or(any_range ? 1 : 0)) {
          if (tid == 0) {
            const int G = (int)gridDim.x;
            s_lastarr = (atomicAdd(&p.ctl[2], 1 + (wg_failed ? 0x10000 : 0)) & 0xffff) == G - 1;
            int spins = 0;
            while ((__hip_atomic_load(&p.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffff) < G &&
                   __hip_atomic_load(&p.ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 15))
              __builtin_amdgcn_s_sleep(8);
            const int now = __hip_atomic_load(&p.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int mine = (now & 0xffff) >= G ? ((now >> 16) != 0 ? 1 : 3) : 2;
            const int was = atomicCAS(&p.ctl[3], 0, mine);
            s_dec = was != 0 ? was : mine;
            if (was == 0 && mine == 2) atomicAdd(&p.ctl[4], 1);       // (diagnostics: a wait ran out)
          }
          __syncthreads();
          if (s_dec == 2 && wrote_here) owes_release = true;         // (step 3 may rewrite this workgroup's rows: see above)
          if (s_dec == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // step 1's rows leave this XCD's L2 before round 1 writes them again
            __syncthreads();
            if (tid == 0) atomicAdd(&p.ctl[6], 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int b = tid; b < p.B && b < kFlagCache; b += kThreads)     // the flag words as they are now, in one round trip
              s_flag[b] = (unsigned short)(__hip_atomic_load(&p.flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (2 * kRedoFailed - 1));
            __syncthreads();
            round = 1;
          } else if (s_dec == 3) {
            // Everybody is here and nothing is left for the extended-range redo.  If nothing is flagged for its inputs either
            // (step 2), the launch is over but for step 3 -- and the last workgroup is known: the one that arrived last.  The
            // others leave now, without a second release + ticket (256 workgroups released at the same moment by the wait and
            // all writing back their L2 at once: 18 us at the headline shape).
            bool hard = false;
            for (int b = tid; b < p.B; b += kThreads) { const int f = flag_of(b); hard |= f != 0 && !range_only(f) && !ext_candidate(f); }
            if (!__syncthreads_or(hard ? 1 : 0)) {
              if (!s_lastarr) return;
              last_by_arrival = true;
            }
          }
        }
      }
      stamp(2);
      if (round >= 0) {
        build_list(round);
        const int nx = s_next;
        // (few utterances: alpha and beta of an utterance on two workgroups, item 2 i + side)
        const bool split = 2 * nx <= (int)gridDim.x && !E2E_EXT_NOSPLIT;
        for (int i2 = blockIdx.x; i2 < (split ? 2 * nx : nx); i2 += gridDim.x) {
          const int i = split ? i2 >> 1 : i2;
          if (round == 1) {
            // (every workgroup has released what step 1 wrote before any chain of this round reports itself done -- and only then
            //  does a segment of the round write a row; the chains take > 100 us, the releases a few: nobody waits here in practice)
            if (tid == 0) {
              int spins = 0;
              while (__hip_atomic_load(&p.ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x && ++spins < (1 << 15))
                __builtin_amdgcn_s_sleep(8);
              s_go = __hip_atomic_load(&p.ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)gridDim.x;
              if (!s_go) { atomicOr(&p.flags[s_xb[i]], kExtDone | kExtBad); atomicAdd(&p.ctl[4], 1); }      // left to step 3
            }
            __syncthreads();
            const bool go_now = s_go;
            __syncthreads();                 // (everybody has read s_go before thread 0 writes it for the next utterance)
            if (!go_now) continue;
          }
          ext_chains<IO>(p, s_xb[i], split ? (i2 & 1) : -1); owes_release = true;
        }
        stamp(3);
        if (nx > 0) ext_segments_of_list(nx);
        if (round == 1) {
          for (int b = tid; b < p.B && b < kFlagCache; b += kThreads) s_flag[b] &= (unsigned short)(kRedoFailed - 1);
          __syncthreads();
        }
      }
    }
    stamp(4);
    // ---- 2. what is flagged for its INPUTS (bad lengths, a blank inside the targets, probabilities at the end of f32, a
    //      protocol error): the reference's arithmetic, utterance h on workgroup h mod nslabs ----
    int h = 0;
    for (int c0 = 0; c0 < p.B; c0 += 64) {
      const int bb = c0 + lane;
      const int f = bb < p.B ? flag_of(bb) : 0;
      unsigned long long hard = __ballot(f != 0 && !range_only(f) && !ext_candidate(f) && !(any_ext0 && range_bits_only(f)));
      while (hard) {
        const int l = __builtin_ctzll(hard); hard &= hard - 1;
        if ((int)blockIdx.x < p.nslabs && h % p.nslabs == (int)blockIdx.x) {
          ctc_exact_one<IO, SCALED>(p, smem, c0 + l, blockIdx.x);
          __syncthreads();
          owes_release = true;
        }
        h++;
      }
    }
  }
  // ---- 3. the last workgroup to get here: what the segment redo could not settle, then the reduction ----
  // (release / acquire at agent scope: what other workgroups stored must have left their XCD's L2)
  __shared__ int s_last;
  if (!last_by_arrival) {
    // (the release is an L2 write-back; what the last workgroup reads of the others' work is their losses -- gradient rows need no
    //  reader inside the launch -- so only a workgroup that wrote losses owes it: with all 256 released by the same event and
    //  writing back at once it was 18 us of the launch)
    if (__syncthreads_or(owes_release ? 1 : 0)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&p.ctl[0], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (tid == 0) { const unsigned long long t = wall_clock64(); p.ctl[26] = (int)(unsigned)t; p.ctl[27] = (int)(unsigned)(t >> 32); }
  if (p.mode == 1) {
    for (int c0 = 0; c0 < p.B; c0 += 64) {
      const int bb = c0 + lane;
      const int f = bb < p.B ? __hip_atomic_load(&p.flags[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      // what nothing has settled: a failed segment redo or an extended-range candidate without good extended-range chains
      const bool settled = (f & kExtDone) != 0 && (f & kExtBad) == 0;
      const int f0 = f & (kRedoFailed - 1);
      unsigned long long failed = __ballot(((f & kRedoFailed) != 0 || ext_candidate(f0) || (any_ext0 && range_bits_only(f0))) && !settled);
      while (failed) {
        const int l = __builtin_ctzll(failed); failed &= failed - 1;
        if (tid == 0) atomicAdd(&p.ctl[1], 1);          // (diagnostics: redone in full although only the segments' range gave out)
        ctc_exact_one<IO, SCALED>(p, smem, c0 + l, 0);
        __syncthreads();
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
  }
  write_reduction(true);
  if (tid == 0) { const unsigned long long t = wall_clock64(); p.ctl[28] = (int)(unsigned)t; p.ctl[29] = (int)(unsigned)(t >> 32); }
}

size_t exact_lds_bytes(int V, int Smax) {
  const size_t Lmax = 2 * (size_t)Smax + 1, S1 = Smax > 0 ? Smax : 1;
  return sizeof(double) * (2 * Lmax + 2 * S1 + 40) + sizeof(int) * (Lmax + 2 * S1 + ((size_t)V + 31) / 32);
}
// waves per workgroup that fit the redo's LDS beside the kernel's static 12.5 KB (wide alphabets: fewer than 8)
int retry_waves(int V, int ppl) {
  const size_t n = (160 * 1024 - 14 * 1024) / retry_wave_lds_bytes(V, ppl);
  return n < 1 ? 0 : n > (size_t)(kThreads / 64) ? kThreads / 64 : (int)n;
}
size_t retry_lds_bytes(int V, int ppl) { return retry_waves(V, ppl) * retry_wave_lds_bytes(V, ppl); }

}  // namespace

constexpr int kFallbackSlabs = 24;     // alpha slabs (3.2 MB each at C2) of the flagged-utterance launch: the workgroups that may run the full
                                       // recomputation at the same time.  (256 of them -- one per workgroup, 821 MB -- until the f64 redo
                                       // of the segments moved to one wave per segment and needed none.)
constexpr int kFallbackGrid = 256;     // workgroups of that launch: 2 048 waves for the redo of the segments

static size_t exact_bytes_for(int slabs, int T, int Smax) {
  const size_t Lmax = 2 * (size_t)Smax + 1;
  return align_up((size_t)slabs * T * Lmax * sizeof(double), 256) + align_up((size_t)slabs * T * sizeof(double), 256) +
         align_up((size_t)slabs * T * sizeof(int), 256);
}

size_t exact_workspace_bytes(int B, int T, int V, int Smax) {
  (void)V;
  return exact_bytes_for(B, T, Smax);
}

size_t exact_fallback_workspace_bytes(int B, int T, int V, int Smax) {
  (void)V;
  return exact_bytes_for(B < kFallbackSlabs ? B : kFallbackSlabs, T, Smax);
}

int launch_exact_flagged(const LossArgs& a, int* flags, int mode, const FastRetry* retry);

int launch_exact(const LossArgs& a) { return launch_exact_flagged(a, nullptr, 0, nullptr); }

int launch_exact_flagged(const LossArgs& a, int* flags, int mode, const FastRetry* retry) {
  size_t lds = exact_lds_bytes(a.V, a.Smax);
  if (mode == 1 && retry && retry_lds_bytes(a.V, retry->PPL) > lds) lds = retry_lds_bytes(a.V, retry->PPL);
  // the extended-range redo stands in where the scaled form may (f32 lattice behind an AUTO call); its LDS must fit beside the rest
  const bool ext_ok = mode == 1 && retry && a.scaled_exact && (a.dtype == E2E_F32 || dtype_is_16bit(a.dtype)) &&
                      retry->PPL >= 1 && 2 * a.Smax + 2 <= retry->CELLS && ExtLds::bytes(a.V) <= 120 * 1024 && getenv("E2E_NO_EXT") == nullptr;
  if (ext_ok && ExtLds::bytes(a.V) > lds) lds = ExtLds::bytes(a.V);
  // (the flagged launch keeps 12.5 KB of static LDS -- flag cache, extended-range lists -- beside the dynamic part; the
  //  all-utterances kernel of mode 0 has none and may take the whole 160 KiB)
  const size_t lds_cap = mode == 0 ? 160 * 1024 : 146 * 1024;
  if (lds > lds_cap) {
    set_error("exact CTC kernel: V=%d, Smax=%d need %zu B of LDS (> %zu KiB%s)", a.V, a.Smax, lds, lds_cap / 1024,
              mode == 0 ? "" : " beside its static 12.5");
    return E2E_ERR_UNSUPPORTED;
  }
  const int slabs = mode == 0 ? a.B : (a.B < kFallbackSlabs ? a.B : kFallbackSlabs);
  const size_t need = mode == 2 ? 0 : exact_bytes_for(slabs, a.T, a.Smax);
  if (mode != 2 && (a.ws_bytes < need || !a.ws)) {
    set_error("workspace too small: %zu < %zu", a.ws_bytes, need);
    return E2E_ERR_WORKSPACE;
  }
  ExactParams p;
  p.x = a.x; p.sB = a.sB; p.sT = a.sT; p.sV = a.sV; p.logprobs = a.logprobs;
  p.targets = a.targets; p.tgt_stride = a.tgt_stride; p.x_len = a.x_len; p.t_len = a.t_len;
  p.B = a.B; p.T = a.T; p.V = a.V; p.Smax = a.Smax; p.Lmax = 2 * a.Smax + 1; p.blank = a.blank;
  p.losses = a.losses; p.grads = a.grads; p.flags = flags; p.mode = mode; p.nslabs = slabs;
  p.ctl = retry ? retry->ctl : nullptr;
  p.gscale = a.grad_scale; p.reduced = mode != 0 ? a.reduced : nullptr; p.reduction = a.reduction;
  if (mode != 0 && !p.ctl) { set_error("internal: flagged exact launch without control words"); return E2E_ERR_ARG; }
  // (eight pairs per lane -- targets beyond 255 labels -- have no f64 redo of the segments: the full recomputation takes them)
  const bool f32_lattice = a.dtype == E2E_F32 || dtype_is_16bit(a.dtype);      // (16-bit I/O: the fast path's f32 lattice behind it)
  p.has_retry = (mode == 1 && retry && f32_lattice && retry_waves(a.V, retry->PPL) > 0 &&
                 (retry->PPL <= 4 || (retry->PPL == 8 && a.scaled_exact))) ? 1 : 0;      // (eight pairs per lane: the scaled instances only)
  p.redo_waves = retry ? retry_waves(a.V, retry->PPL) : 0;
  p.has_ext = ext_ok ? 1 : 0;
  if (p.has_retry || p.has_ext) p.retry = *retry; else memset(&p.retry, 0, sizeof(p.retry));
  p.ws_alpha = reinterpret_cast<double*>(a.ws);
  p.ws_lse = reinterpret_cast<double*>(reinterpret_cast<char*>(a.ws) +
                                       align_up((size_t)slabs * a.T * p.Lmax * sizeof(double), 256));
  p.ws_exp = reinterpret_cast<int*>(reinterpret_cast<char*>(p.ws_lse) + align_up((size_t)slabs * a.T * sizeof(double), 256));
  const bool scaled = a.scaled_exact && f32_lattice;
  p.scaled = scaled ? 1 : 0;
  const int grid = mode == 0 ? slabs : (a.B < kFallbackGrid ? a.B : kFallbackGrid);
  if (a.B == 0) return E2E_OK;
  auto go = [&](auto kernel) -> int {
    E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds),
                  "hipFuncSetAttribute");
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds, a.stream, p);
    return E2E_OK;
  };
  int rc;
  if (dtype_is_16bit(a.dtype)) {
    // 16-bit I/O exists behind the fast / wide paths only (flagged utterances of an AUTO or FAST call)
    if (mode == 0) { set_error("the exact kernel takes f32 / f64 logits (16-bit logits: algo AUTO)"); return E2E_ERR_UNSUPPORTED; }
    const bool p8 = p.has_retry && p.retry.PPL == 8;
    if (a.dtype == E2E_F16) rc = p8 ? go(&ctc_exact_kernel<f16_t, true, true>) : scaled ? go(&ctc_exact_kernel<f16_t, true>) : go(&ctc_exact_kernel<f16_t, false>);
    else rc = p8 ? go(&ctc_exact_kernel<bf16_t, true, true>) : scaled ? go(&ctc_exact_kernel<bf16_t, true>) : go(&ctc_exact_kernel<bf16_t, false>);
  } else if (mode == 0) {
    if (a.dtype != E2E_F32) rc = go(&ctc_exact_all_kernel<double, false>);
    else if (scaled) rc = go(&ctc_exact_all_kernel<float, true>);
    else rc = go(&ctc_exact_all_kernel<float, false>);
  } else if (a.dtype != E2E_F32) rc = go(&ctc_exact_kernel<double, false>);
  else if (p.has_retry && p.retry.PPL == 8) rc = go(&ctc_exact_kernel<float, true, true>);
  else if (scaled) rc = go(&ctc_exact_kernel<float, true>);
  else rc = go(&ctc_exact_kernel<float, false>);
  if (rc != E2E_OK) return rc;
  E2E_HIP_CHECK(hipGetLastError(), "ctc_exact_kernel launch");
  return E2E_OK;
}

}  // namespace e2e

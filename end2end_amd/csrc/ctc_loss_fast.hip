// Fast CTC forward-backward for MI355X: scaled linear-domain lattice, no MFMA.
//
// Why not the log domain: a log-sum-exp cell costs ~4 quarter-rate transcendentals; at
// B=256 (one utterance per CU) that alone is ~60 us of VALU time.  In the probability domain a
// cell is add + fma + mul.  Why not store alpha: L*T cells per utterance (1.6 MB in f32 at
// T=1000, S=200) is 14x the algorithmic bytes once written and re-read.  So:
//
//  kernel F1  (one 4-wave workgroup per utterance, one wave per SIMD)
//     wave 0  alpha chain, t = 0..T-1      | f64 state in registers, 2*PPL lattice cells per lane
//     wave 1  beta  chain, t = T-1..0      | (cells 2i = blank, 2i+1 = label i), neighbours by DPP
//     wave 2/3 softmax rows for wave 0/1   -> LDS ring of probability rows (f64), hand-off by
//                                             per-block sequence words in LDS, no barriers
//     Every 8 steps a chain rescales its row by a power of two (exponent of the row maximum),
//     every 16 steps it stores the row as an f32 checkpoint.  Outputs: loss, checkpoints,
//     per-8-step exponents, the probability rows (for F2), alpha-side and beta-side log Z.
//  kernel F2  (one wave per (utterance, 16-step segment): thousands of independent waves)
//     recomputes the 16 alpha rows of its segment from the checkpoint into registers, walks
//     beta backwards through the segment, forms the posteriors alpha*beta/sum, accumulates them
//     per label in LDS and writes the gradient rows (prob - posterior), coalesced.
//  exact kernel (ctc_loss_exact.hip) re-does the utterances F1/F2 flag: infeasible alignments,
//     range underflow (sum alpha*beta below 2^-90 of the row scales), alpha/beta log Z mismatch,
//     targets that contain the blank id.
//
// Reference semantics restated: src/losses/ctc_loss.cpp:33-117 (recurrences, loss, gradient).
#include "common.h"

// the scaled lattice is not bit-pinned to the reference: let the compiler fuse multiply-adds here
#pragma clang fp contract(fast)

namespace e2e {
namespace {

constexpr int kSeg = 16;        // steps per F2 segment == checkpoint spacing
constexpr int kBlk = 8;         // prep -> chain hand-off block and rescale period
constexpr int kRingBlks = 8;    // ring depth (blocks)
constexpr int kMaxSmallV = 64;  // probability row fits one lane group

struct FastParams {
  const float* x; int64_t sB, sT, sV; int logprobs;
  const int64_t* targets; int64_t tgt_stride;
  const int64_t* x_len; const int64_t* t_len;
  int B, T, V, Smax, blank;
  float* losses; float* grads;
  float* ytab;     // [B][T][V]  probabilities y_t[v]
  float* ckA;      // [B][NS][CELLS]  row k: alpha row at t = 16k-1 (k >= 1)
  float* ckQ;      // [B][NS][CELLS]  row k: beta-with-emission row at t = 16k (k >= 1)
  short* ckE;      // [B][NS][2][64]  per-lane exponent of checkpoint row k (0: alpha, 1: beta); -30000 = all zero
  short* escA;     // [B][NB]   exponent removed from the alpha row at step 8n+7
  short* escB;     // [B][NB]   exponent removed from the beta row at step 8n
  double* logz;    // [B][2]    alpha-side / beta-side log Z
  int* flags;      // [B]       != 0: redo with the exact kernel
  int NS, NB, CELLS;
};

// ---- cross-lane helpers (wave64) ------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);
}
// lane n <- lane n-1 (lane 0 keeps 0)
__device__ __forceinline__ double from_prev_lane(double v) {
  const int lo = dpp_i<0x138>(0, __double2loint(v)), hi = dpp_i<0x138>(0, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float from_prev_lane(float v) {
  return __int_as_float(dpp_i<0x138>(0, __float_as_int(v)));
}
// lane n <- lane n+1 (lane 63 keeps 0)
__device__ __forceinline__ double from_next_lane(double v) {
  const int lo = dpp_i<0x130>(0, __double2loint(v)), hi = dpp_i<0x130>(0, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float from_next_lane(float v) {
  return __int_as_float(dpp_i<0x130>(0, __float_as_int(v)));
}

__device__ __forceinline__ float wave_sum(float v) {
  v += __int_as_float(dpp_i<0xB1>(0, __float_as_int(v)));    // quad_perm [1,0,3,2]
  v += __int_as_float(dpp_i<0x4E>(0, __float_as_int(v)));    // quad_perm [2,3,0,1]
  v += __int_as_float(dpp_i<0x141>(0, __float_as_int(v)));   // row_half_mirror
  v += __int_as_float(dpp_i<0x140>(0, __float_as_int(v)));   // row_mirror
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
  v = max(v, dpp_i<0xB1>(0, v));
  v = max(v, dpp_i<0x4E>(0, v));
  v = max(v, dpp_i<0x141>(0, v));
  v = max(v, dpp_i<0x140>(0, v));
  v = max(v, __shfl_xor(v, 16, 64));
  v = max(v, __shfl_xor(v, 32, 64));
  return v;
}
// reductions inside aligned groups of G lanes (G = 2..64, power of two)
template <int G>
__device__ __forceinline__ float group_max(float v) {
  for (int o = 1; o < G; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void spin_until(volatile int* p, int want) {
  while (*p != want) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void publish(volatile int* p, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  *p = v;
}

// per-lane lattice description shared by F1 and F2: lane holds pairs i = PPL*lane + r
template <int PPL>
struct LaneCells {
  int lab[PPL];        // label of pair r, or V (the always-zero column) when i >= S
  float skp[PPL];      // r^2 if the alpha skip (i-1) -> i is allowed (ctc_loss.cpp:53-57), else 0
  float skn[PPL];      // r^2 if the beta skip i -> (i+1) is allowed (ctc_loss.cpp:91-96), else 0
  float r;             // tilt: the rows hold alpha[j]*r^j and beta[j]*r^(L-1-j); their product is unchanged
  bool has_blank_label;
  // Exponential tilting.  Untilted, alpha_t favours j ~ t (most alignments) and beta_t favours L-j ~ T-t, so for
  // uninformative emissions their masses sit hundreds of cells apart and sum_j alpha*beta is ~2^-250 of the row
  // maxima: out of f32 range.  Weighting a step of one cell by r = sqrt(rho/(1-rho)), rho = S/T (the change of
  // measure under which a frame starts a new label with probability rho), centres both rows on the diagonal.
  // It costs nothing: the recurrences keep their shape with multipliers (1, r, r^2) instead of (1, 1, 1).
  __device__ void load(const int64_t* tg, int S, int T, int V, int blank, int lane) {
    has_blank_label = false;
    {
      // the number of alignments of t frames to i labels grows by ~((t-i)/(2i))^2 per extra label, so the
      // untilted maximum sits at i = t/3; r = 2*rho/(1-rho) moves it to i = rho*t
      float rho = (float)S / (float)T;
      rho = fminf(fmaxf(rho, 1.f / 33.f), 0.8f);
      r = S > 0 ? 2.f * rho / (1.f - rho) : 1.f;
    }
#pragma unroll
    for (int q = 0; q < PPL; q++) {
      const int i = PPL * lane + q;
      const int li = i < S ? (int)tg[i] : -1;
      const int lp = (i >= 1 && i - 1 < S) ? (int)tg[i - 1] : -1;
      const int ln = (i + 1 < S) ? (int)tg[i + 1] : -1;
      lab[q] = i < S ? li : V;
      skp[q] = (i < S && i >= 1 && li != blank && lp != li) ? r * r : 0.f;
      skn[q] = (i + 1 < S && li != blank && ln != li) ? r * r : 0.f;
      if (i < S && (li == blank || li < 0 || li >= V)) has_blank_label = true;
      if (i < S && (li < 0 || li >= V)) lab[q] = V;
    }
  }
};

// ============================================================================================
// F1: the two serial chains
// ============================================================================================
template <int PPL>
__global__ __launch_bounds__(256) void ctc_fast_chain_kernel(FastParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NC = 2 * PPL;                     // cells per lane
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V, blank = p.blank, ROW = V + 1;   // column V of every row is 0
  double* ring = reinterpret_cast<double*>(smem);                  // [2][kRingBlks*kBlk][ROW]
  volatile int* filled = reinterpret_cast<volatile int*>(ring + 2 * kRingBlks * kBlk * ROW);  // [2][kRingBlks]
  volatile int* freed = filled + 2 * kRingBlks;                    // [2][kRingBlks]

  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  const bool bad = Tq < 1 || Tq > p.T || Sq < 0 || Sq > p.Smax;
  if (bad) {                       // the exact kernel poisons this utterance
    if (tid == 0) { p.flags[b] = 1; p.losses[b] = __builtin_nanf(""); }   // reason bit 0: bad lengths
    return;
  }
  const int T = (int)Tq, S = (int)Sq, L = 2 * S + 1;
  if (tid < 2 * 2 * kRingBlks) const_cast<int*>(filled)[tid] = 0;
  __syncthreads();

  const int dir = wid & 1;                                  // 0 alpha (forward), 1 beta (backward)
  const int nblk = (T + kBlk - 1) / kBlk;
  double* myring = ring + (size_t)dir * kRingBlks * kBlk * ROW;
  volatile int* myfilled = filled + dir * kRingBlks;
  volatile int* myfreed = freed + dir * kRingBlks;
  const float* x = p.x + (int64_t)b * p.sB;

  if (wid >= 2) {
    // ---------------- prep wave: probability rows into the ring ----------------
    // VP lanes per row (next power of two >= V), RPP rows per pass
    int VP = 2; while (VP < V) VP <<= 1;
    const int RPP = 64 / VP, grp = lane / VP, v = lane - grp * VP;
    float* ytab = p.ytab + (size_t)b * p.T * V;
    for (int n = 0; n < nblk; n++) {
      const int slot = n % kRingBlks;
      if (n >= kRingBlks) spin_until(&myfreed[slot], n - kRingBlks + 1);
      for (int tt = grp; tt < kBlk; tt += RPP) {
        const int sidx = n * kBlk + tt;
        const int t = dir == 0 ? sidx : T - 1 - sidx;
        const bool live = sidx < T && v < V;
        const float xv = live ? x[(int64_t)t * p.sT + (int64_t)v * p.sV] : -__builtin_huge_valf();
        float y;
        if (p.logprobs) {
          y = expf(xv);
        } else {
          float m = xv, e;
          switch (VP) {   // group-wide softmax
            case 2: m = group_max<2>(m); e = expf(xv - m); y = e / group_sum<2>(e); break;
            case 4: m = group_max<4>(m); e = expf(xv - m); y = e / group_sum<4>(e); break;
            case 8: m = group_max<8>(m); e = expf(xv - m); y = e / group_sum<8>(e); break;
            case 16: m = group_max<16>(m); e = expf(xv - m); y = e / group_sum<16>(e); break;
            case 32: m = group_max<32>(m); e = expf(xv - m); y = e / group_sum<32>(e); break;
            default: m = group_max<64>(m); e = expf(xv - m); y = e / group_sum<64>(e); break;
          }
        }
        if (live) {
          myring[(size_t)(slot * kBlk + tt) * ROW + v] = (double)y;
          if (dir == 0) ytab[(size_t)t * V + v] = y;
        }
        if (sidx < T && v == 0) myring[(size_t)(slot * kBlk + tt) * ROW + V] = 0.0;
      }
      if (lane == 0) publish(&myfilled[slot], n + 1); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    return;
  }

  // ---------------- chain wave ----------------
  __builtin_amdgcn_s_setprio(3);
  LaneCells<PPL> lc;
  lc.load(p.targets + (int64_t)b * p.tgt_stride, S, T, V, blank, lane);
  const double rr = (double)lc.r;
  if (dir == 0 && __any(lc.has_blank_label)) { if (lane == 0) atomicOr(&p.flags[b], 2); }
  double sk[PPL];
#pragma unroll
  for (int r = 0; r < PPL; r++) sk[r] = dir == 0 ? (double)lc.skp[r] : (double)lc.skn[r];
  const bool cond = (T > 1 || L == 1);            // ctc_loss.cpp:39,76

  double c[NC];                                    // the row: c[2r] blank cell 2i, c[2r+1] label cell 2i+1
#pragma unroll
  for (int k = 0; k < NC; k++) c[k] = 0.0;
  int e_pending = 0;                               // exponent measured one step earlier
  int e_total = 0;                                 // sum of removed exponents
  float* ck = (dir == 0 ? p.ckA : p.ckQ) + (size_t)b * p.NS * p.CELLS;
  short* esc = (dir == 0 ? p.escA : p.escB) + (size_t)b * p.NB;

  for (int n = 0; n < nblk; n++) {
    const int slot = n % kRingBlks;
    spin_until(&myfilled[slot], n + 1);
    const double* rows = myring + (size_t)slot * kBlk * ROW;
#pragma unroll
    for (int tt = 0; tt < kBlk; tt++) {
      const int sidx = n * kBlk + tt;
      if (sidx < T) {
        const int t = dir == 0 ? sidx : T - 1 - sidx;
        const double* row = rows + tt * ROW;
        const double yb = row[blank];
        double e[PPL];
#pragma unroll
        for (int r = 0; r < PPL; r++) e[r] = row[lc.lab[r]];
        if (dir == 0) {
          // alpha_t[j] = (alpha[j] + alpha[j-1] + skip*alpha[j-2]) * y_t[l_j], ctc_loss.cpp:47-60
          if (sidx == 0) {
#pragma unroll
            for (int k = 0; k < NC; k++) c[k] = 0.0;
            if (lane == 0) { c[0] = cond ? yb : 0.0; c[1] = rr * e[0]; }   // ctc_loss.cpp:39-42 (e[0]=0 when S=0)
          } else {
            double pl = from_prev_lane(c[NC - 1]);        // label cell just below this lane's first blank
#pragma unroll
            for (int r = 0; r < PPL; r++) {
              const double ob = c[2 * r], ol = c[2 * r + 1];
              c[2 * r] = (ob + rr * pl) * yb;
              c[2 * r + 1] = (ol + rr * ob + sk[r] * pl) * e[r];
              pl = ol;
            }
          }
        } else {
          // q_t[j] = (q[j] + q[j+1] + skipn*q[j+2]) * y_t[l_j]; q = beta * emission, ctc_loss.cpp:84-99
          if (sidx == 0) {
#pragma unroll
            for (int k = 0; k < NC; k++) c[k] = 0.0;
#pragma unroll
            for (int r = 0; r < PPL; r++) {
              const int i = PPL * lane + r;
              if (2 * i == L - 1 && cond) c[2 * r] = yb;             // ctc_loss.cpp:76
              if (2 * i + 1 == L - 2) c[2 * r + 1] = rr * e[r];      // ctc_loss.cpp:78
            }
          } else {
            double nb = from_next_lane(c[0]), nl = from_next_lane(c[1]);   // next lane's first blank / label
#pragma unroll
            for (int r = PPL - 1; r >= 0; r--) {
              const double ob = c[2 * r], ol = c[2 * r + 1];
              c[2 * r + 1] = (ol + rr * nb + sk[r] * nl) * e[r];
              c[2 * r] = (ob + rr * ol) * yb;
              nb = ob; nl = ol;
            }
          }
        }
        // power-of-two rescale: measure at phase 6 (alpha: t%8==6, beta: t%8==1), remove at phase 7
        const int ph = dir == 0 ? (t & 7) : 7 - (t & 7);
        if (ph == 7) {
          if (e_pending != 0) {
#pragma unroll
            for (int k = 0; k < NC; k++) c[k] = ldexp(c[k], -e_pending);
          }
          if (lane == 0) esc[t >> 3] = (short)e_pending;
          e_total += e_pending;
          e_pending = 0;
          const int kk = dir == 0 ? (t + 1) : t;            // alpha row 16k-1 / beta row 16k -> slot k
          if ((kk & (kSeg - 1)) == 0 && kk > 0 && kk < T) {
            // block floating point: each lane stores its cells scaled by its own exponent (f32 keeps every
            // lane's cells however far apart the lanes' magnitudes are)
            int m = 0;
#pragma unroll
            for (int k = 0; k < NC; k++) m = max(m, __double2hiint(c[k]));
            const int own = m > 0 ? ((m >> 20) & 0x7ff) - 1023 : -30000;
            float* dst = ck + (size_t)(kk / kSeg) * p.CELLS + lane * NC;
#pragma unroll
            for (int k = 0; k < NC; k++) dst[k] = m > 0 ? (float)ldexp(c[k], -own) : 0.f;
            p.ckE[(((size_t)b * p.NS + kk / kSeg) * 2 + dir) * 64 + lane] = (short)own;
          }
        } else if (ph == 6) {
          int hi = 0;
#pragma unroll
          for (int k = 0; k < NC; k++) hi = max(hi, __double2hiint(c[k]));   // positive doubles order like ints
          hi = wave_max(hi);
          hi = __builtin_amdgcn_readfirstlane(hi);
          e_pending = hi > 0 ? ((hi >> 20) & 0x7ff) - 1023 : 0;
          if (e_pending < -1000) e_pending = -1000;
        }
      }
    }
    if (lane == 0) publish(&myfreed[slot], n + 1);
  }

  // ---- log Z from this side ----
  double z = 0.0;
  if (dir == 0) {
#pragma unroll
    for (int k = 0; k < NC; k++) {
      const int j = NC * lane + k;
      if (j == L - 1) z += c[k];                          // ctc_loss.cpp:63-70, un-tilted relative to cell L-1
      if (j == L - 2) z += rr * c[k];
    }
  } else if (lane == 0) {
    z = (cond ? c[0] : 0.0) + rr * c[1];                  // sum_j alpha_0[j]*beta_0[j]
  }
  // wave sum of a double through two xor-butterflies on the halves is overkill: only <= 2 lanes hold data
  for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
  if (lane == 0) {
    const double lz = log(z) + (double)e_total * 0.693147180559945309417 - (double)(L - 1) * log(rr);
    p.logz[2 * b + dir] = lz;
    if (dir == 0) {
      p.losses[b] = (float)(-lz);
      if (!(z > 0.0) || !(z < __builtin_huge_val())) atomicOr(&p.flags[b], 4);     // infeasible or out of range
    }
  }
}

// ============================================================================================
// F2: one wave per (utterance, segment)
// ============================================================================================
template <int PPL>
__global__ __launch_bounds__(64) void ctc_fast_segment_kernel(FastParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NC = 2 * PPL;
  constexpr int kSlope = 3 * NC;    // exponent drop allowed per lane (see the alpha load below)
  const int b = blockIdx.y, seg = blockIdx.x, lane = threadIdx.x;
  const int V = p.V, blank = p.blank, ROW = V + 1, Tmax = p.T;
  float* ys = reinterpret_cast<float*>(smem);          // [kSeg][ROW]  probabilities, column V = 0
  float* post = ys + kSeg * ROW;                       // [kSeg][ROW]  per-label posteriors
  float* postb = post + kSeg * ROW;                    // [kSeg][16]   partial sums of the blank cells
  const int t0 = seg * kSeg;
  float* grads = p.grads + (size_t)b * Tmax * V;
  const float* x = p.x + (int64_t)b * p.sB;

  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  if (Tq < 1 || Tq > Tmax || Sq < 0 || Sq > p.Smax) return;    // flagged by F1, the exact kernel poisons it
  const int T = (int)Tq, S = (int)Sq, L = 2 * S + 1;
  const int tend = min(t0 + kSeg, Tmax);

  // frames past the utterance's end: exp(lp) in log-prob mode (quirk Q1), zero for fused logits
  for (int idx = max(t0, T) * V + lane; idx < tend * V; idx += 64) {
    const int t = idx / V, v = idx - t * V;
    grads[idx] = p.logprobs ? expf(x[(int64_t)t * p.sT + (int64_t)v * p.sV]) : 0.f;
  }
  if (t0 >= T) return;
  const int t1 = min(t0 + kSeg, T), n = t1 - t0;

  // stage the segment's probability rows, clear the accumulators
  const float* ytab = p.ytab + ((size_t)b * Tmax + t0) * V;
  for (int idx = lane; idx < n * V; idx += 64) { const int tt = idx / V; ys[tt * ROW + (idx - tt * V)] = ytab[idx]; }
  for (int tt = lane; tt < kSeg; tt += 64) ys[tt * ROW + V] = 0.f;
  for (int idx = lane; idx < kSeg * ROW; idx += 64) post[idx] = 0.f;
  for (int idx = lane; idx < kSeg * 16; idx += 64) postb[idx] = 0.f;

  LaneCells<PPL> lc;
  lc.load(p.targets + (int64_t)b * p.tgt_stride, S, T, V, blank, lane);
  const float rr = lc.r;
  const bool cond = (T > 1 || L == 1);
  const short* escA = p.escA + (size_t)b * p.NB;
  const short* escB = p.escB + (size_t)b * p.NB;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);   // LDS staging visible to this (single) wave

  // ---- alpha rows of the segment, kept in registers ----
  float A[kSeg][NC];
  float a[NC];
  int eA = 0;                       // this lane's exponent for its alpha cells during the segment
  if (seg == 0) {
#pragma unroll
    for (int k = 0; k < NC; k++) a[k] = 0.f;
  } else {
    const float* src = p.ckA + ((size_t)b * p.NS + seg) * p.CELLS + lane * NC;
#pragma unroll
    for (int k = 0; k < NC; k++) a[k] = src[k];
    const int own = p.ckE[(((size_t)b * p.NS + seg) * 2 + 0) * 64 + lane];
    // Mass flows from lane n-1 to lane n and can cross 2 cells per step, i.e. 32/NC lanes within the segment;
    // every lane crossed multiplies the stored value by the hand-over factor 2^(eA[n-1]-eA[n]).  Limiting the
    // exponent drop to 3 bits per cell (kSlope per lane) bounds the worst growth over a segment by 2^96.  A lane
    // whose own cells lie further below its left neighbour than that takes the neighbour's unit minus kSlope and
    // keeps its cells as small numbers (still exact down to 2^-126 of that unit).
    eA = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(eA, d, 64);
      if (lane >= d) eA = max(eA, v - kSlope * d);
    }
    const int sh = max(own - eA, -200);
#pragma unroll
    for (int k = 0; k < NC; k++) a[k] = ldexpf(a[k], sh);
  }
  float fA, fB;                     // hand-over factors: alpha from lane-1, beta from lane+1
  {
    const int ep = __shfl_up(eA, 1, 64), en = __shfl_down(eA, 1, 64);
    fA = lane > 0 ? ldexpf(1.f, max(min(ep - eA, 126), -126)) : 0.f;
    fB = lane < 63 ? ldexpf(1.f, max(min(eA - en, 126), -126)) : 0.f;
  }
#pragma unroll
  for (int tt = 0; tt < kSeg; tt++) {
    if (tt < n) {
      const int t = t0 + tt;
      const float* row = ys + tt * ROW;
      const float yb = row[blank];
      if (t == 0) {
#pragma unroll
        for (int k = 0; k < NC; k++) a[k] = 0.f;
        if (lane == 0) { a[0] = cond ? yb : 0.f; a[1] = rr * row[lc.lab[0]]; }
      } else {
        float pl = from_prev_lane(a[NC - 1]) * fA;
#pragma unroll
        for (int r = 0; r < PPL; r++) {
          const float ob = a[2 * r], ol = a[2 * r + 1];
          a[2 * r] = (ob + rr * pl) * yb;
          a[2 * r + 1] = (ol + rr * ob + lc.skp[r] * pl) * row[lc.lab[r]];
          pl = ol;
        }
      }
      if ((t & 7) == 7) {
        const int e = escA[t >> 3];
        if (e != 0) {
#pragma unroll
          for (int k = 0; k < NC; k++) a[k] = ldexpf(a[k], -e);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) A[tt][k] = a[k];
  }

  // ---- beta backwards through the segment, posteriors, per-label accumulation ----
  // beta rows live in the reciprocal units of the alpha lanes (lane n: 2^(emax - eA_n)), so that alpha*beta is
  // in one common unit across the wave; the hand-over factor from lane n+1 is then 2^(eA_n - eA_{n+1}) <= 2^kSlope
  float q[NC];
  const bool last_seg = (t1 == T);
  const int lane_last = (L - 1) / NC;                 // lane holding cell L-1
  const int eA_ref = __shfl(eA, lane_last, 64);
  const float end_unit = ldexpf(1.f, max(min(eA - eA_ref, 126), -126));
  if (last_seg) {
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = 0.f;
  } else {
    const float* src = p.ckQ + ((size_t)b * p.NS + seg + 1) * p.CELLS + lane * NC;
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = src[k];
    const int ownB = p.ckE[(((size_t)b * p.NS + seg + 1) * 2 + 1) * 64 + lane];
    const int E = eA + ownB;
    const int emax = wave_max(E);
    const int sh = max(E - emax, -200);
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = ldexpf(q[k], sh);
  }
  float smin = __builtin_huge_valf();
  bool finite_ok = true;
#pragma unroll
  for (int tt = kSeg - 1; tt >= 0; tt--) {
    if (tt < n) {
      const int t = t0 + tt;
      const float* row = ys + tt * ROW;
      const float yb = row[blank];
      float bs[NC];                 // beta_t[j] (no emission at t)
      if (t == T - 1) {
#pragma unroll
        for (int r = 0; r < PPL; r++) {
          const int i = PPL * lane + r;
          bs[2 * r] = (2 * i == L - 1 && cond) ? end_unit : 0.f;
          bs[2 * r + 1] = (2 * i + 1 == L - 2) ? rr * end_unit : 0.f;
        }
      } else {
        float nb = from_next_lane(q[0]) * fB, nl = from_next_lane(q[1]) * fB;
#pragma unroll
        for (int r = PPL - 1; r >= 0; r--) {
          bs[2 * r + 1] = q[2 * r + 1] + rr * nb + lc.skn[r] * nl;
          bs[2 * r] = q[2 * r] + rr * q[2 * r + 1];
          nb = q[2 * r]; nl = q[2 * r + 1];
        }
      }
      float pj[NC], part = 0.f, pblank = 0.f;
#pragma unroll
      for (int k = 0; k < NC; k++) { pj[k] = A[tt][k] * bs[k]; part += pj[k]; }
      const float s = wave_sum(part);
      smin = fminf(smin, s);
      finite_ok = finite_ok && (s < __builtin_huge_valf());
      const float inv = 1.0f / s;
#pragma unroll
      for (int r = 0; r < PPL; r++) {
        pblank += pj[2 * r];
        const float pv = pj[2 * r + 1] * inv;
        if (pv != 0.f) atomicAdd(&post[tt * ROW + lc.lab[r]], pv);     // ds_add_f32, no return
      }
      pblank *= inv;
      if (pblank != 0.f) atomicAdd(&postb[tt * 16 + (lane & 15)], pblank);
      // q_t = beta_t * y_t
#pragma unroll
      for (int r = 0; r < PPL; r++) {
        q[2 * r] = bs[2 * r] * yb;
        q[2 * r + 1] = bs[2 * r + 1] * row[lc.lab[r]];
      }
      if ((t & 7) == 0) {
        const int e = escB[t >> 3];
        if (e != 0) {
#pragma unroll
          for (int k = 0; k < NC; k++) q[k] = ldexpf(q[k], -e);
        }
      }
    }
  }
  // range / consistency check: everything that carries posterior mass was representable
  if (!(smin >= 0x1p-90f) || !finite_ok) { if (lane == 0) atomicOr(&p.flags[b], finite_ok ? 8 : 16); }
  if (seg == 0 && lane == 0) {
    const double za = p.logz[2 * b], zb = p.logz[2 * b + 1];
    if (!(fabs(za - zb) <= 1e-6 * fabs(za) + 1e-4)) atomicOr(&p.flags[b], 32);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  // ---- gradient rows: y - posterior (d loss/d logits in fused mode; exp(lp) - posterior otherwise) ----
  for (int idx = lane; idx < n * V; idx += 64) {
    const int tt = idx / V, v = idx - tt * V;
    float pv = post[tt * ROW + v];
    if (v == blank) {
      float sb = 0.f;
#pragma unroll
      for (int g = 0; g < 16; g++) sb += postb[tt * 16 + g];
      pv += sb;
    }
    grads[(size_t)t0 * V + idx] = ys[tt * ROW + v] - pv;
  }
}

template <int PPL>
int launch_fast_ppl(const FastParams& p, hipStream_t stream) {
  const size_t lds1 = sizeof(double) * 2 * kRingBlks * kBlk * (p.V + 1) + sizeof(int) * 4 * kRingBlks;
  const size_t lds2 = sizeof(float) * (2 * kSeg * (p.V + 1) + kSeg * 16);
  hipLaunchKernelGGL(ctc_fast_chain_kernel<PPL>, dim3(p.B), dim3(256), lds1, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_kernel launch");
  hipLaunchKernelGGL(ctc_fast_segment_kernel<PPL>, dim3(p.NS, p.B), dim3(64), lds2, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_segment_kernel launch");
  return E2E_OK;
}

int ppl_for(int Smax) {
  // pairs per lane: 64*PPL pairs plus the final blank must fit 128*PPL cells
  if (Smax <= 63) return 1;
  if (Smax <= 127) return 2;
  if (Smax <= 255) return 4;
  return 0;
}

struct FastLayout {
  size_t ytab, ckA, ckQ, ckE, escA, escB, logz, flags, total;
  int NS, NB, CELLS;
};

FastLayout fast_layout(int B, int T, int V, int Smax) {
  FastLayout l;
  const int ppl = ppl_for(Smax);
  l.CELLS = 128 * (ppl > 0 ? ppl : 1);
  l.NS = (T + kSeg - 1) / kSeg;
  l.NB = (T + kBlk - 1) / kBlk + 1;
  size_t o = 0;
  l.ytab = o; o += align_up((size_t)B * T * V * sizeof(float), 256);
  l.ckA = o; o += align_up((size_t)B * l.NS * l.CELLS * sizeof(float), 256);
  l.ckQ = o; o += align_up((size_t)B * l.NS * l.CELLS * sizeof(float), 256);
  l.ckE = o; o += align_up((size_t)B * l.NS * 2 * 64 * sizeof(short), 256);
  l.escA = o; o += align_up((size_t)B * l.NB * sizeof(short), 256);
  l.escB = o; o += align_up((size_t)B * l.NB * sizeof(short), 256);
  l.logz = o; o += align_up((size_t)B * 2 * sizeof(double), 256);
  l.flags = o; o += align_up((size_t)B * sizeof(int), 256);
  l.total = o;
  return l;
}

}  // namespace

bool fast_supported(int T, int V, int Smax, int dtype) {
  (void)T;
  return dtype == E2E_F32 && V >= 2 && V <= kMaxSmallV && ppl_for(Smax) != 0;
}

size_t fast_workspace_bytes(int B, int T, int V, int Smax) {
  return fast_layout(B, T, V, Smax).total;
}

int launch_exact_flagged(const LossArgs& a, const int* flags, int mode);

int launch_fast(const LossArgs& a, bool fallback_to_exact) {
  const FastLayout l = fast_layout(a.B, a.T, a.V, a.Smax);
  size_t need = l.total;
  if (fallback_to_exact) need += exact_workspace_bytes(a.B, a.T, a.V, a.Smax);
  if (!a.ws || a.ws_bytes < need) { set_error("workspace too small: %zu < %zu", a.ws_bytes, need); return E2E_ERR_WORKSPACE; }
  if (a.B == 0) return E2E_OK;
  char* ws = reinterpret_cast<char*>(a.ws);
  FastParams p;
  p.x = reinterpret_cast<const float*>(a.x); p.sB = a.sB; p.sT = a.sT; p.sV = a.sV; p.logprobs = a.logprobs;
  p.targets = a.targets; p.tgt_stride = a.tgt_stride; p.x_len = a.x_len; p.t_len = a.t_len;
  p.B = a.B; p.T = a.T; p.V = a.V; p.Smax = a.Smax; p.blank = a.blank;
  p.losses = reinterpret_cast<float*>(a.losses); p.grads = reinterpret_cast<float*>(a.grads);
  p.ytab = reinterpret_cast<float*>(ws + l.ytab);
  p.ckA = reinterpret_cast<float*>(ws + l.ckA); p.ckQ = reinterpret_cast<float*>(ws + l.ckQ);
  p.ckE = reinterpret_cast<short*>(ws + l.ckE);
  p.escA = reinterpret_cast<short*>(ws + l.escA); p.escB = reinterpret_cast<short*>(ws + l.escB);
  p.logz = reinterpret_cast<double*>(ws + l.logz); p.flags = reinterpret_cast<int*>(ws + l.flags);
  p.NS = l.NS; p.NB = l.NB; p.CELLS = l.CELLS;
  E2E_HIP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)a.B * sizeof(int), a.stream), "hipMemsetAsync(flags)");
  int rc;
  switch (ppl_for(a.Smax)) {
    case 1: rc = launch_fast_ppl<1>(p, a.stream); break;
    case 2: rc = launch_fast_ppl<2>(p, a.stream); break;
    case 4: rc = launch_fast_ppl<4>(p, a.stream); break;
    default: set_error("fast CTC path: Smax=%d too long", a.Smax); return E2E_ERR_UNSUPPORTED;
  }
  if (rc != E2E_OK) return rc;
  LossArgs e = a;
  e.ws = ws + l.total; e.ws_bytes = a.ws_bytes - l.total;
  // mode 1: redo flagged utterances exactly; mode 2: no fallback requested -> poison them
  return launch_exact_flagged(e, p.flags, fallback_to_exact ? 1 : 2);
}

}  // namespace e2e

// Diagnostics (not part of include/e2e_ctc.h): copy the fast path's per-utterance flag words and both log Z
// values out of a workspace that the last e2e_ctc_loss_fwd_bwd(ALGO_FAST/AUTO) call used.  Synchronises.
extern "C" int e2e_debug_fast_state(const void* workspace, int B, int T, int V, int Smax, int* flags_host, double* logz_host) {
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const char* ws = reinterpret_cast<const char*>((base + 255) & ~(uintptr_t)255);
  const e2e::FastLayout l = e2e::fast_layout(B, T, V, Smax);
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  if (hipMemcpy(flags_host, ws + l.flags, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  if (hipMemcpy(logz_host, ws + l.logz, sizeof(double) * 2 * (size_t)B, hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  return E2E_OK;
}

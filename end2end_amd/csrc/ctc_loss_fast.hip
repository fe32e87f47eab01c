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
//     cumulative removed exponents per 8-step block, the probability rows (for F2), alpha-side
//     and beta-side log Z, log2 of the tilted partition sum (what every row of F2 must reproduce).
//  kernel F2  (one wave per (utterance, 16-step segment): thousands of independent waves)
//     recomputes the 16 alpha rows of its segment from the checkpoint into registers, walks
//     beta backwards through the segment, forms the posteriors alpha*beta/sum, accumulates them
//     per label in LDS and writes the gradient rows (prob - posterior), coalesced.
//     Every row checks sum_j alpha*beta against the chains' partition sum (finish_rows).
//  exact kernel (ctc_loss_exact.hip) re-does the utterances F1/F2 flag: infeasible alignments,
//     rows that fail the self-check or leave f32's range (segments redone in f64 from the
//     checkpoints first), alpha/beta log Z mismatch, targets that contain the blank id.
//
// Reference semantics restated: src/losses/ctc_loss.cpp:33-117 (recurrences, loss, gradient).
#include "fast_common.h"

namespace e2e {
using namespace fastk;
namespace {

// One serial chain (DIR 0: alpha forward, DIR 1: beta-with-emission backward).
//
// A lone wave issues at most one instruction every ~4.5 cycles whatever its kind, so the chain's speed is its
// instruction count; LDS and flag latencies must not add to it.  Hence: (a) the hand-over words are polled one
// block ahead and looked at late; (b) the block's probabilities sit in ONE register set that is refilled in halves
// while the other half is in use (steps 4..7 of block n at the start of block n, steps 0..3 of block n+1 at step 4
// of block n), every read being issued four steps before its first use; (c) the steady blocks have a loop of their
// own, one straight-line body without liveness tests.
// (d) the row is kept in a form that needs 4 instead of 5 multiply-adds per label pair: a blank cell is held BEFORE its
// emission (B~ = B / y_t[blank]) and a label cell pre-multiplied by the tilt (L^ = r L), so that
//     B~' = B~ * yb + L^prev                         (yb: blank probability of the frame just processed)
//     L^' = (L^ + (r^2 yb) * B~ + skip * L^prev) * y'[label]
// -- the same recurrence, with the blank emission applied one step late.  Checkpoints and log Z convert back.
template <int PPL, int DIR, int RB = kRingBlks>
__device__ __forceinline__ void chain_wave(const FastParams& p, int b, int T, int S, const F1Lds& lds, int lane) {
  constexpr int NC = 2 * PPL;
  const int V = p.V, blank = p.blank, L = 2 * S + 1;
  const int nblk = (T + kBlk - 1) / kBlk;
  const double* myring = lds.ring + (size_t)DIR * RB * lds.blk_elems;
  volatile int* myfilled = lds.filled + DIR * kRingBlks;
  __builtin_amdgcn_s_setprio(3);
  unsigned long long prof_spin = 0, prof_t0 = __builtin_amdgcn_s_memtime(), prof_steps = 0;
  (void)prof_spin; (void)prof_t0; (void)prof_steps;
  LaneCells<PPL> lc;
  lc.load(p.targets + (int64_t)b * p.tgt_stride, S, T, V, blank, lane);
  const double rr = (double)lc.r;
  if (DIR == 0 && __any(lc.has_blank_label)) { if (lane == 0) atomicOr(&p.flags[b], 2); }
  double sk[PPL];
#pragma unroll
  for (int r = 0; r < PPL; r++) sk[r] = DIR == 0 ? (double)lc.skp[r] : (double)lc.skn[r];
  const bool cond = (T > 1 || L == 1);            // ctc_loss.cpp:39,76

  double c[NC];                                    // the row: c[2r] = B~ of blank cell 2i, c[2r+1] = L^ of label cell 2i+1
#pragma unroll
  for (int k = 0; k < NC; k++) c[k] = 0.0;
  double yb_prev = 0.0;                            // blank probability of the frame processed last
  const double rr2 = rr * rr, inv_rr = 1.0 / rr;
  int e_pending = 0;                               // exponent measured one step earlier
  int e_total = 0;                                 // sum of removed exponents
  float* ck = (DIR == 0 ? p.ckA : p.ckQ) + (size_t)b * p.NS * p.CELLS;
  int* cum = (DIR == 0 ? p.cumA : p.cumB) + (size_t)b * p.NB;
  if (lane == 0) {
    if (DIR == 0) cum[0] = 0;
    else { cum[((T - 1) >> 3) + 1] = 0; cum[((T - 1) >> 3) + 2] = 0; }
  }

  // the probabilities of a block: per label cell (and for the blank) four wide reads of 2 steps each
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 eraw[PPL][4], braw[4];
  auto load_half = [&](int n, auto half_tag) {
    constexpr int H = decltype(half_tag)::value;
    const double* blk = myring + (size_t)(n % RB) * lds.blk_elems;
#pragma unroll
    for (int r = 0; r < PPL; r++) {
      const d2* src = reinterpret_cast<const d2*>(blk + lc.lab[r] * kRow);
      eraw[r][2 * H] = src[2 * H]; eraw[r][2 * H + 1] = src[2 * H + 1];
    }
    const d2* srcb = reinterpret_cast<const d2*>(blk + blank * kRow);
    braw[2 * H] = srcb[2 * H]; braw[2 * H + 1] = srcb[2 * H + 1];
  };

  // One block of 8 steps.  STEADY: all 8 rows are live and none is the chain's first row -- no per-step tests.
  auto run_block = [&](int n, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    load_half(n, std::integral_constant<int, 1>{});
    // the LDS runs this wave's operations in order, so the producers' next writes cannot overtake the reads above
    publish(lds.took + DIR, n + 1);
    const bool want_next = n + 1 < nblk;
    int next_filled = 0;
    if (want_next) next_filled = peek(&myfilled[(n + 1) % RB]);
    double yb[kBlk], e[kBlk][PPL];
    const int tbase = block_time(DIR, n, 0, T);          // t of tt = 0; t = tbase +/- tt
#ifdef E2E_FAST_PROFILE
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int tt = 0; tt < kBlk; tt++) {
      const int t = DIR == 0 ? tbase + tt : tbase - tt;
      yb[tt] = braw[tt >> 1][tt & 1];
#pragma unroll
      for (int r = 0; r < PPL; r++) e[tt][r] = eraw[r][tt >> 1][tt & 1];
      if (tt == 4) {
        // the halves for steps 0..3 are dead by now.  The producers normally run several blocks ahead; if not, the
        // wave waits here.  (Behind the last block the read fetches a stale slot that nobody uses.)
        if (want_next && __builtin_amdgcn_readfirstlane(next_filled) != n + 2) {
          PROF_SPIN_BEGIN spin_until(&myfilled[(n + 1) % RB], n + 2); PROF_SPIN_END(prof_spin)
        }
        load_half(n + 1, std::integral_constant<int, 0>{});
      }
      if (STEADY || t < T) {
        const bool first = !STEADY && (DIR == 0 ? t == 0 : t == T - 1);
        if (DIR == 0) {
          // alpha_t[j] = (alpha[j] + r*alpha[j-1] + r^2*skip*alpha[j-2]) * y_t[l_j], ctc_loss.cpp:47-60
          if (first) {
            if (lane == 0) { c[0] = cond ? 1.0 : 0.0; c[1] = rr2 * e[tt][0]; }     // ctc_loss.cpp:39-42
          } else {
            double pl = from_prev_lane(c[NC - 1]);        // label cell just below this lane's first blank
            const double w = rr2 * yb_prev;
#pragma unroll
            for (int r = 0; r < PPL; r++) {
              const double ob = c[2 * r], ol = c[2 * r + 1];
              c[2 * r] = ob * yb_prev + pl;
              c[2 * r + 1] = (ol + w * ob + sk[r] * pl) * e[tt][r];
              pl = ol;
            }
          }
        } else {
          // q_t[j] = (q[j] + r*q[j+1] + r^2*skipn*q[j+2]) * y_t[l_j]; q = beta * emission, ctc_loss.cpp:84-99
          if (first) {
#pragma unroll
            for (int r = 0; r < PPL; r++) {
              const int i = PPL * lane + r;
              if (2 * i == L - 1 && cond) c[2 * r] = 1.0;                 // ctc_loss.cpp:76
              if (2 * i + 1 == L - 2) c[2 * r + 1] = rr2 * e[tt][r];      // ctc_loss.cpp:78
            }
          } else {
            double nb = from_next_lane(c[0]), nl = from_next_lane(c[1]);   // next lane's first blank / label
            const double w = rr2 * yb_prev;
#pragma unroll
            for (int r = PPL - 1; r >= 0; r--) {
              const double ob = c[2 * r], ol = c[2 * r + 1];
              c[2 * r + 1] = (ol + w * nb + sk[r] * nl) * e[tt][r];
              c[2 * r] = ob * yb_prev + ol;
              nb = ob; nl = ol;
            }
          }
        }
        yb_prev = yb[tt];
        // power-of-two rescale: measure the row's exponent at position 6 of the block, remove it at position 7
        // (alpha: t%8 == 6 / 7, beta: t%8 == 1 / 0)
        if (tt == 7) {
          if (e_pending != 0) {
#pragma unroll
            for (int k = 0; k < NC; k++) c[k] = ldexp(c[k], -e_pending);
          }
          e_total += e_pending;
          if (lane == 0) cum[(t >> 3) + (DIR == 0 ? 1 : 0)] = e_total;
          e_pending = 0;
          const int kk = DIR == 0 ? (t + 1) : t;            // alpha row 16k-1 / beta row 16k -> slot k
          if ((kk & (kSeg - 1)) == 0 && kk > 0 && kk < T) {
            // block floating point: each lane stores its cells scaled by its own exponent (f32 keeps every
            // lane's cells however far apart the lanes' magnitudes are)
            double cell[NC];                       // the true cells: blank with its emission, label without the tilt
#pragma unroll
            for (int r = 0; r < PPL; r++) { cell[2 * r] = c[2 * r] * yb_prev; cell[2 * r + 1] = c[2 * r + 1] * inv_rr; }
            int m = 0;
#pragma unroll
            for (int k = 0; k < NC; k++) m = max(m, __double2hiint(cell[k]));
            const int own = m > 0 ? ((m >> 20) & 0x7ff) - 1023 : -30000;
            if (lane * NC < L) {                   // (lanes past the lattice's 2S+1 cells: the segment kernel does not read them)
              float* dst = ck + (size_t)(kk / kSeg) * p.CELLS + lane * NC;
#pragma unroll
              for (int k = 0; k < NC; k++) dst[k] = m > 0 ? (float)ldexp(cell[k], -own) : 0.f;
              p.ckE[(((size_t)b * p.NS + kk / kSeg) * 2 + DIR) * 64 + lane] = (short)own;
            }
          }
        } else if (tt == 6) {
          int hi = 0;
#pragma unroll
          for (int k = 0; k < NC; k++) hi = max(hi, __double2hiint(c[k]));   // positive doubles order like ints
          hi = wave_max(hi);
          e_pending = hi > 0 ? ((hi >> 20) & 0x7ff) - 1023 : 0;
          if (e_pending < -1000) e_pending = -1000;
        }
      }
    }
#ifdef E2E_FAST_PROFILE
    prof_steps += __builtin_amdgcn_s_memtime() - ts0;
#endif
  };
  // alpha: block 0 holds the first row, the last block may be short; beta: block 0 holds the first row (and
  // possibly dead rows), every later block is whole
  {
    { PROF_SPIN_BEGIN spin_until(&myfilled[0], 1); PROF_SPIN_END(prof_spin) }
    load_half(0, std::integral_constant<int, 0>{});
    const int steady_end = DIR == 0 ? T / kBlk : nblk;         // blocks [1, steady_end) are steady
    run_block(0, std::false_type{});
    int n = 1;
    for (; n < steady_end; n++) run_block(n, std::true_type{});
    for (; n < nblk; n++) run_block(n, std::false_type{});
  }

  // ---- log Z from this side ----
  double z = 0.0;
  if (DIR == 0) {
#pragma unroll
    for (int k = 0; k < NC; k++) {
      const int j = NC * lane + k;
      if (j == L - 1) z += c[k] * yb_prev;                // ctc_loss.cpp:63-70, un-tilted relative to cell L-1
      if (j == L - 2) z += c[k];                          // (= r * the label cell)
    }
  } else if (lane == 0) {
    z = (cond ? c[0] * yb_prev : 0.0) + c[1];             // sum_j alpha_0[j]*beta_0[j]
  }
#ifdef E2E_FAST_PROFILE
  if (lane == 0 && b < 256) { g_prof[(b * 4 + DIR) * 4 + 0] = __builtin_amdgcn_s_memtime() - prof_t0; g_prof[(b * 4 + DIR) * 4 + 1] = prof_spin;
    g_prof[(b * 4 + DIR) * 4 + 2] = 0; g_prof[(b * 4 + DIR) * 4 + 3] = prof_steps; }
#endif
  for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
  if (lane == 0) {
    const double lz = log(z) + (double)e_total * 0.693147180559945309417 - (double)(L - 1) * log(rr);
    p.logz[2 * b + DIR] = lz;
    if (DIR == 0) p.zt2[b] = log2(z) + (double)e_total;
    if (DIR == 0) {
      p.losses[b] = (float)(-lz);
      if (!(z > 0.0) || !(z < __builtin_huge_val())) atomicOr(&p.flags[b], 4);     // infeasible or out of range
    }
  }
}

// ============================================================================================
// F1: the two serial chains
// ============================================================================================
// Waves of a workgroup land on the SIMDs in the order 0,2,1,3,0,2,1,3: waves 0/1 (the chains) get SIMDs 0 and 2 to
// themselves, waves 2,6 (alpha rows) share SIMD 1, waves 3,7 (beta rows) share SIMD 3; wave 4 writes the lattice
// description for F2 and retires, wave 5 retires at once.
// (Splitting a chain's lanes over two pipelined waves was tried and does not pay: the per-block bookkeeping does not
// shrink with the cells, and a lone wave's speed is its instruction count.)
// RB: the probability ring's depth.  Eight blocks by default; FOUR where eight would keep a second workgroup off the CU although the
// batch has more utterances than the chip has CUs (round 6: one GPU's share of BASELINE configs[4], 512 utterances on 65 compact
// columns -- 84.5 KB of ring per workgroup, i.e. two ROUNDS of 256 workgroups, 61 us for 256 steps; with 42 KB the two rounds
// run side by side on SIMDs that a lone chain wave leaves half idle).
template <int PPL, int RB = kRingBlks>
__global__ E2E_KERNEL_ALIGN __launch_bounds__(512) void ctc_fast_chain_kernel(FastParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V;
  const F1Lds lds(smem, V, RB);

  if (b == 0 && tid < 32) p.ctl[tid] = 0;     // (this kernel ends before the fallback launch, which counts there, starts)
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  const bool bad = Tq < 1 || Tq > p.T || Sq < 0 || Sq > p.Smax;
  if (bad) {                       // the exact kernel poisons this utterance
    if (tid == 0) { p.flags[b] = 1; p.losses[b] = __builtin_nanf(""); }   // reason bit 0: bad lengths
    return;
  }
  const int T = (int)Tq, S = (int)Sq;
  if (tid == 0) p.flags[b] = 0;    // (the barrier below orders this before the waves' atomicOr; saves a memset launch)
  for (int i = tid; i < p.MW; i += blockDim.x) p.segmask[(size_t)b * p.MW + i] = 0u;
  if (tid < F1Lds::kSyncInts) lds.filled[tid] = 0;
  for (int i = tid; i < 2 * RB * kBlk; i += blockDim.x)      // the zero rows (label index V) of every block
    lds.ring[(size_t)(i / kBlk) * lds.blk_elems + V * kRow + (i % kBlk)] = 0.0;
  __syncthreads();

  // (RB == 4, two workgroups per CU: giving the launch's second half its roles two waves on -- its chains on SIMDs 1 and 3 instead
  //  of 0 and 2 again -- was measured in round 6 and is not faster, 70.2 against 68.8 us per call for configs[4]'s compact lattice)
  const int wave = __builtin_amdgcn_readfirstlane(wid);
  if (wave == 0) chain_wave<PPL, 0, RB>(p, b, T, S, lds, lane);
  else if (wave == 1) chain_wave<PPL, 1, RB>(p, b, T, S, lds, lane);
  else if (wave == 4) cellinfo_wave<PPL>(p, b, T, S, lds.sortcnt, lane);
  else if (wave == 5) return;
  else {
    const int d = (wave == 2 || wave == 6) ? 0 : 1;     // waves 2,6 -> alpha rows, waves 3,7 -> beta rows
    const int first = wave >= 6 ? 1 : 0;                 // the two producers of a direction take alternate blocks
    unsigned char* ring = reinterpret_cast<unsigned char*>(lds.ring + (size_t)d * RB * lds.blk_elems);
    const int bb = lds.blk_elems * 8;
    volatile int* fl = lds.filled + d * kRingBlks;
    volatile int* tk = lds.took + d;
    if (V <= 16) prep_wave<2, 0, RB>(p, b, T, d, first, 2, ring, bb, fl, tk, lane);
    else if (V <= 32) prep_wave<4, 0, RB>(p, b, T, d, first, 2, ring, bb, fl, tk, lane);
    else if (V <= 48) prep_wave<6, 0, RB>(p, b, T, d, first, 2, ring, bb, fl, tk, lane);
    else if (V <= 64) prep_wave<8, 0, RB>(p, b, T, d, first, 2, ring, bb, fl, tk, lane);
    else prep_wave<12, 0, RB>(p, b, T, d, first, 2, ring, bb, fl, tk, lane);
  }
}

// ---- the chains in packed f32 -----------------------------------------------------------------
// The halo structure above with an f32 lattice.  What a chain wave costs is its instruction count per step (an in-order
// wave issues an instruction every ~5.5 cycles at best, a dependent one every ~8.5 whatever its type --
// tools/diag/microbench/issue_latency.hip), and v_pk_fma_f32 advances two pairs per instruction: a lane holds the
// pairs (B0, L0), (B1, L1) as the packed registers B = (B0, B1), L = (L0, L1), and a step is
//   alpha: PL = (L1 of the lane below, L0);  B' = B*yb + PL;  L' = (L + wb*B + SK*PL) * E        1 DPP + 1 move + 4 packed
//   beta:  G = wb*B + SK*L;  TK = (G1, G0 of the lane above);  L' = (L + TK) * E;  B' = B*yb + L  1 DPP + 1 move + 5 packed
// for 112 owned pairs per wave (56 lanes + 8 halo lanes = 16 pairs, one of which goes stale per step, so that the edge
// lanes are exchanged every 16 steps): two waves per direction cover S <= 223, three the rest.  The checkpoint rows leave
// a chain wave as one LDS write of its true cells; the direction's frame wave, which has the time, finds the block
// floating-point exponents, converts and stores them.  The probability ring is f32 (label rows of 8 steps; the blank's row holds
// (probability, tilted probability) pairs, so that one packed operand carries both factors).
// Numerics: the cells carry f32 rounding through the whole utterance (~1e-6 relative in the partition sum over 1 000
// steps, measured); the loss is well inside its tolerance with that, and the gradient rows are normalised by their own
// row sum in the segment kernel, so that only the NON-uniform part of the drift reaches them.  The segment kernel's
// self-check compares every row sum with the chains' partition sum and runs at 1e-5 instead of 4e-6 with these chains
// (FastParams::ztol; their own rounding shows as 6e-7 median, 3e-6 at most over the sweeps): a row may lose 7e-6 of its
// posterior mass before it is redone, an absolute gradient error of that size on top of the chains' drift (worst
// gradient element over four randomised sweeps: 8.4e-6 off; include/e2e_ctc.h states 2e-5 for the option).  The common frame lags 16 steps as above; f32 has 126 bits of range for it, and a row that sinks further is
// caught by the self-check (the cells that matter were flushed).
constexpr int kHfHalo = 8;                    // halo lanes = 16 pairs: the waves exchange edge lanes every SECOND block
constexpr int kHfOwnLanes = 64 - kHfHalo;     // 56
constexpr int kHfOwn = 2 * kHfOwnLanes;       // 112 pairs a wave owns
constexpr int kHfLag = 2;                     // the frame follows the row's maximum two blocks late (one: the waves meet at every
                                              // block's end, 153 instead of 145 us per step at the headline shape)
// f32: the cells are kept 2^bias above that frame.  A row sinks 26-42 bits per block for uninformative emissions at V = 29..64
// and is not rescaled at all during its first three blocks, and cells 100 bits under the row's maximum still carry
// posterior mass at some t (measured: with the maximum at 2^-25 .. 2^-67, 15 % of the utterances at V = 64 lose 1e-4 of
// log Z).  Without tilt > 1 a row's maximum grows by at most 3x per step (alpha[j] <= 3 max(alpha) y), 2^25.4 over the
// 16 steps of lag: bias 100 cannot overflow.  With tilt > 1 (dense targets) rows can in principle grow faster; those keep
// 10 bits more head room (with 40 instead, a third of the dense utterances at V = 64 sank out of range), and an overflow
// ends as a non-finite row sum, which the segment kernel flags.
// (Extrapolating the sinking rate to decide the frame was tried: the differences amplify the row-to-row variation.)
__device__ __forceinline__ int hf_bias(float r_tilt) { return r_tilt <= 1.f ? 100 : 90; }

typedef float h_f2 __attribute__((ext_vector_type(2)));
typedef float h_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) h_f4 lds_f4;

// The two arithmetics of the halo chains.  Both hold two label pairs per lane as B = (B0, B1), L = (L0, L1).
//   ChainF32: packed instructions, 3 waves per direction at most (any S <= 255), 3 producers per direction, 16 waves.
//   ChainF64: the same structure in f64 -- the default for targets of 128..223 labels: every result as the single-wave
//             chains give it (same recurrence, same f64 cells; powers of two apart), 2 waves per direction, 2 producers
//             per direction, 12 waves.
struct ChainF32 {
  typedef float T; typedef h_f2 V2; typedef float R;      // cells, packed cells, ring elements
  static constexpr bool kF32 = true, kBigV = false;
  static constexpr int kRing = kRingBlks, kRingElem = 4;
  static constexpr int kMaxW = 3, kProducers = 3, kRowElems = kRow32, kElem = 4;
  static constexpr int kWaves = 2 * kMaxW + 2 + 2 * kProducers + 2;
  static constexpr bool kPairedLoop = true;
  __device__ static int top(float v) { return __float_as_int(v); }             // positive values order like these ints
  __device__ static int expo(int m) { return ((m >> 23) & 0xff) - 127; }
  __device__ static int bias(float r) { return hf_bias(r); }
  __device__ static float scale(float v, int e) { return ldexpf(v, e); }
  __device__ static float tilt2(float r) { return r * r; }
};
struct ChainF64 {
  typedef double T; typedef h_d2 V2; typedef double R;
  static constexpr bool kF32 = false, kBigV = false;
  static constexpr int kRing = kRingBlks, kRingElem = 8;
#ifndef E2E_F64_PRODUCERS             // Two producers per direction: 12 waves, i.e. three per SIMD and 168 registers for the chain
#define E2E_F64_PRODUCERS 2           // waves' paired interior loop (with three producers -- 14 waves, 128 registers -- that loop
#endif                                // spills in the beta wave: 155 against 131 us per step at the headline shape)
  static constexpr int kMaxW = 2, kProducers = E2E_F64_PRODUCERS, kRowElems = kRow, kElem = 8;      // (three waves per direction: 174
                                                                                       //  against 167 us at S in [200, 255] -- six chain waves on four SIMDs)
  static constexpr int kWaves = 2 * kMaxW + 2 + 2 * kProducers + 2;
  static constexpr bool kPairedLoop = true;
  __device__ static int top(double v) { return __double2hiint(v); }
  __device__ static int expo(int m) { return ((m >> 20) & 0x7ff) - 1023; }
  __device__ static int bias(float) { return 0; }                                // (f64 cells have the range for the lag)
  __device__ static double scale(double v, int e) { return ldexp(v, e); }
  __device__ static double tilt2(float r) { return (double)r * (double)r; }
};

//   ChainF64L: ChainF64 on four waves per direction -- targets of 256..447 labels (eight pairs per segment-kernel lane).  16
//             waves leave 128 registers per wave: the plain block loop (the paired one spills there).  Not a tuned path: it
//             exists so that long transcripts are not left to the exact kernel (~7 ms per 1000 frames).
struct ChainF64L : ChainF64 {
  static constexpr int kMaxW = 4, kProducers = 2;
  static constexpr int kWaves = 2 * kMaxW + 2 + 2 * kProducers + 2;
  static constexpr bool kPairedLoop = false;
};

//   ChainF64W: ChainF64 for alphabets of 97..224 columns (the compacted wide-alphabet targets of more than 95 word pieces, above
//             all): the probability ring is f32 -- what the producers compute anyway -- and four blocks deep, 87 KB at 224
//             columns (f64 and eight deep: 290 KB); a chain wave converts what it reads (four conversions per step) and forms
//             the tilted blank probability itself.  Same cells, same recurrence, same results as ChainF64.
#ifndef E2E_W_PRODUCERS
#define E2E_W_PRODUCERS 4
#endif
#ifndef E2E_W_RING                 // (tools/diag; round 5, whole call at B=256 T=1000 V=128 S<=100: 4 blocks 247 us, 6: 244, 8: 240 -- and
#define E2E_W_RING 4               //  beyond ~150 columns more than four do not fit the LDS: not worth an instance of its own)
#endif
struct ChainF64W : ChainF64 {
  typedef float R;
  static constexpr bool kBigV = true;
  static constexpr int kRing = E2E_W_RING, kRingElem = 4, kRowElems = kRow32;
  // the producers are what bounds these chains (28 columns per lane and block: ~700 instructions): four per direction, 16 waves,
  // 128 registers -- the plain block loop, as ChainF64L
  static constexpr int kProducers = E2E_W_PRODUCERS;
  static constexpr int kWaves = 2 * kMaxW + 2 + 2 * kProducers + 2;
  static constexpr bool kPairedLoop = E2E_W_PRODUCERS <= 2;
};

//   ChainF64LW: ChainF64L over the f32 ring -- alphabets of up to 448 columns with targets of up to 447 labels (word-piece
//             targets of 224..447 pieces).  The ring is two blocks deep (21.6 KB per block at 448 columns: 154 KB of LDS in
//             all), so a chain wave waits for its producers at every block; served, not tuned, like ChainF64L.
struct ChainF64LW : ChainF64L {
  typedef float R;
  static constexpr bool kBigV = true;
  static constexpr int kRing = 2, kRingElem = 4, kRowElems = kRow32;
};

struct HfLds {
  // byte offsets from the start of the workgroup's LDS
  int ring;        // [2][kRingBlks] blocks of blk_bytes: (V+1) label rows of kRowElems cells (row V: zeros) + 16 cells (yb, wb) x 8 steps
  int blk_bytes;
  int filled;      // [2][kRingBlks] ints; used: [dir][f] = kProducers + the last block producer f of the direction has finished
  int sortcnt;     // [130] ints (cellinfo_wave; 258 for more than 127 columns)
  int bnd;         // [2][kMaxW][kHaloSlots][kHfHalo] x 4 cells: wave w's edge lanes (B0, L0, B1, L1) after block n
  int zacc;        // [8] doubles
  int prog;        // [2][8] ints
  int exw;         // [2][kHaloSlots] ints
  int mxl;         // [2][kHaloSlots][kMaxW][64] ints
  int ckb;         // [2][2][kMaxW][64] x 4 cells: a checkpoint row's true cells (B0, L0, B1, L1 per lane), double-buffered
  int ckdone;      // [2] ints: checkpoint rows the direction's checkpoint wave has finished reading
  int total;
  __host__ __device__ HfLds(int V, int row_elems, int ring_elem, int elem, int maxw, int depth) {
    ring = 0;
    blk_bytes = ((V + 1) * row_elems + 16) * ring_elem;
    filled = ring + 2 * depth * blk_bytes;
    sortcnt = filled + 2 * kRingBlks * 4;
    bnd = (sortcnt + (V > 127 ? lstart_ints(V) : 130) * 4 + 15) & ~15;
    zacc = bnd + 2 * maxw * kHaloSlots * kHfHalo * 4 * elem;
    prog = zacc + 64;
    exw = prog + 2 * 8 * 4;
    mxl = exw + 2 * kHaloSlots * 4;
    ckb = mxl + 2 * kHaloSlots * maxw * 64 * 4;
    ckdone = ckb + 2 * 2 * maxw * 64 * 4 * elem;
    total = ckdone + 16;
  }
  template <typename X> __host__ __device__ static HfLds of(int V) { return HfLds(V, X::kRowElems, X::kRingElem, X::kElem, X::kMaxW, X::kRing); }
};

template <int DIR, int F2PPL, typename X>
__device__ __forceinline__ void hf_chain_wave(const FastParams& p, int b, int T, int S, unsigned char* smem, const HfLds hl,
                                              int lane, int w, int W) {
  typedef typename X::T CT;
  typedef typename X::V2 V2;
  constexpr int kLane4 = 4 * X::kElem;                            // bytes of a lane's four cells
  lds_u8* L0 = (lds_u8*)smem;
  const int V = p.V, blank = p.blank, L = 2 * S + 1;
  const int nblk = (T + kBlk - 1) / kBlk;
  const int ring_off = hl.ring + DIR * X::kRing * hl.blk_bytes;
  volatile int* myfilled = reinterpret_cast<int*>(smem + hl.filled) + DIR * kRingBlks;
  lds_u8* prog = L0 + hl.prog + DIR * 32;
  lds_u8* exw = L0 + hl.exw + DIR * (kHaloSlots * 4);
  lds_u8* mxl = L0 + hl.mxl + ((DIR * kHaloSlots * X::kMaxW + w) * 64 + lane) * 4;      // + slot * kMaxW * 256
  __builtin_amdgcn_s_setprio(3);
  unsigned long long prof_fill = 0, prof_nb = 0, prof_lag = 0, prof_t0 = __builtin_amdgcn_s_memtime();
  (void)prof_fill; (void)prof_nb; (void)prof_lag; (void)prof_t0;

  // this lane's two label pairs p0, p0 + 1: blank cells 2*p0, 2*p0 + 2, label cells 2*p0 + 1, 2*p0 + 3
  const int p0 = kHfOwn * w + 2 * (DIR == 0 ? lane - kHfHalo : lane);
  const bool owned = (DIR == 0 ? lane >= kHfHalo : lane < kHfOwnLanes) && p0 < 64 * F2PPL;
  const bool halo = DIR == 0 ? lane < kHfHalo : lane >= kHfOwnLanes;
  const bool has_up = DIR == 0 ? w > 0 : w < W - 1;
  const bool has_down = DIR == 0 ? w < W - 1 : w > 0;
  const int up = DIR == 0 ? w - 1 : w + 1;
  const int64_t* tg = p.targets + (int64_t)b * p.tgt_stride;
  const float r_tilt = fast_tilt(S, T);
  const CT rr2 = X::tilt2(r_tilt), inv_rr = (CT)1 / (CT)r_tilt;
  const CT skip_w = (CT)(r_tilt * r_tilt);        // (the segment kernel's weight of a skip: the tilt squared in f32)
  int lab[2]; CT skv[2]; bool badlab = false;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int pr = p0 + r;
    const bool in = pr >= 0 && pr < S;
    const int li = in ? (int)tg[pr] : -1;
    const int lpv = (pr >= 1 && pr - 1 < S) ? (int)tg[pr - 1] : -1;
    lab[r] = (in && li >= 0 && li < V) ? li : V;           // V: the always-zero row
    // alpha: the skip (pr-1) -> pr as pair pr sees it; beta: the same skip as pair pr-1 sees it (this pair prepares what
    // the label cell of the pair below takes from it)
    skv[r] = DIR == 0 ? ((in && pr >= 1 && li != blank && lpv != li) ? skip_w : (CT)0)   // ctc_loss.cpp:53-57
                      : ((in && pr >= 1 && lpv != blank && li != lpv) ? skip_w : (CT)0); // ctc_loss.cpp:91-96
    badlab |= owned && in && (li == blank || li < 0 || li >= V);
  }
  if (DIR == 0 && __any(badlab)) { if (lane == 0) atomicOr(&p.flags[b], 2); }
  const V2 SK = {skv[0], skv[1]};
  const bool cond = (T > 1 || L == 1);            // ctc_loss.cpp:39,76

  V2 Bc = {(CT)0, (CT)0}, Lc = {(CT)0, (CT)0};     // B~ (blank cells before their emission), L^ (label cells, tilted), times 2^bias
  const int bias = X::bias(r_tilt);
  const CT kOne = X::scale((CT)1, bias);
  CT yb_prev = (CT)0, wb_prev = (CT)0;
  int e_total = 0;
  int nck = 0;                                     // checkpoint rows handed to the checkpoint wave
  int lead = 0;                                    // blocks of probabilities known to be in the ring: [0, lead)
  auto need_blocks = [&](int k) {                  // (the producers run several blocks ahead: one look every few blocks)
    if (lead < k) {
      PROF_SPIN_BEGIN
      for (;;) {
        int a = peek(&myfilled[0]);
#pragma unroll
        for (int f = 1; f < X::kProducers; f++) a = min(a, peek(&myfilled[f]));
        lead = __builtin_amdgcn_readfirstlane(a);
        if (lead >= k) break;
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      PROF_SPIN_END(prof_fill)
    }
  };

  CT e0[kBlk], e1[kBlk], ybw[2 * kBlk];             // the block's probabilities of the two labels; (yb, wb) per step
  constexpr int kRowBytes = X::kRowElems * X::kRingElem;
  const int lab0_off = lab[0] * kRowBytes, lab1_off = lab[1] * kRowBytes, yw_off = (V + 1) * kRowBytes;
  auto load_half = [&](int n, auto half_tag) {     // steps 4H .. 4H+3 of block n
    constexpr int H = decltype(half_tag)::value;
    const int yo = ring_off + (n % X::kRing) * hl.blk_bytes;
    if constexpr (X::kBigV) {
      // f32 ring, f64 cells: four steps of a label per read; the tilted blank probability is formed here (exact, as the
      // producers of the f64 ring form it)
      const h_f4 a0 = *(lds_f4*)(L0 + yo + lab0_off + 16 * H), a1 = *(lds_f4*)(L0 + yo + lab1_off + 16 * H);
      const h_f4 y0 = *(lds_f4*)(L0 + yo + yw_off + 32 * H), y1 = *(lds_f4*)(L0 + yo + yw_off + 32 * H + 16);
#pragma unroll
      for (int k = 0; k < 4; k++) { e0[4 * H + k] = (CT)a0[k]; e1[4 * H + k] = (CT)a1[k]; }
      ybw[8 * H + 0] = (CT)y0[0]; ybw[8 * H + 2] = (CT)y0[2]; ybw[8 * H + 4] = (CT)y1[0]; ybw[8 * H + 6] = (CT)y1[2];
#pragma unroll
      for (int k = 0; k < 4; k++) ybw[8 * H + 2 * k + 1] = rr2 * ybw[8 * H + 2 * k];
    } else if constexpr (X::kF32) {
      const h_f4 a0 = *(lds_f4*)(L0 + yo + lab0_off + 16 * H), a1 = *(lds_f4*)(L0 + yo + lab1_off + 16 * H);
      const h_f4 y0 = *(lds_f4*)(L0 + yo + yw_off + 32 * H), y1 = *(lds_f4*)(L0 + yo + yw_off + 32 * H + 16);
#pragma unroll
      for (int k = 0; k < 4; k++) { e0[4 * H + k] = a0[k]; e1[4 * H + k] = a1[k]; ybw[8 * H + k] = y0[k]; ybw[8 * H + 4 + k] = y1[k]; }
    } else {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const h_d2 a0 = *(lds_d2*)(L0 + yo + lab0_off + 32 * H + 16 * q), a1 = *(lds_d2*)(L0 + yo + lab1_off + 32 * H + 16 * q);
        e0[4 * H + 2 * q] = a0.x; e0[4 * H + 2 * q + 1] = a0.y; e1[4 * H + 2 * q] = a1.x; e1[4 * H + 2 * q + 1] = a1.y;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const h_d2 y = *(lds_d2*)(L0 + yo + yw_off + 64 * H + 16 * q);
        ybw[8 * H + 2 * q] = y.x; ybw[8 * H + 2 * q + 1] = y.y;
      }
    }
  };
  auto store4 = [&](int off, CT x0, CT x1, CT x2, CT x3) {      // a lane's four cells
    if constexpr (X::kF32) { h_f4 v; v.x = x0; v.y = x1; v.z = x2; v.w = x3; *(lds_f4*)(L0 + off) = v; }
    else { h_d2 u, v; u.x = x0; u.y = x1; v.x = x2; v.y = x3; *(lds_d2*)(L0 + off) = u; *(lds_d2*)(L0 + off + 16) = v; }
  };

  // PAR / CK: the block's parity and whether it ends in a checkpoint row, where the caller knows them at compile time
  // (the interior loop below: an in-order wave pays ~5 cycles for every scalar test and index computation as well, and
  // there were a hundred of those per block); -1: decided here.
  int ckbuf = 0;                                   // which of the two checkpoint buffers the next row goes to
  auto run_block = [&](int n, auto steady_tag, auto par_tag, auto ck_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    constexpr int PAR = decltype(par_tag)::value, CK = decltype(ck_tag)::value;
    load_half(n, std::integral_constant<int, 1>{});
    const bool want_next = PAR >= 0 ? true : n + 1 < nblk;
    if (PAR >= 0 ? (PAR == 0 && has_up) : (n > 0 && (n & 1) == 0 && has_up)) {          // the halo lasts two blocks
      { PROF_SPIN_BEGIN HALO_WAIT(__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(prog + 4 * up)) >= n); PROF_SPIN_END(prof_nb) }
      const int off = hl.bnd + (((DIR * X::kMaxW + up) * kHaloSlots + ((n - 1) & (kHaloSlots - 1))) * kHfHalo + (lane & (kHfHalo - 1))) * kLane4;
      if constexpr (X::kF32) {
        const h_f4 v = *(lds_f4*)(L0 + off);
        if (halo) { Bc.x = v.x; Lc.x = v.y; Bc.y = v.z; Lc.y = v.w; }
      } else {
        const h_d2 u = *(lds_d2*)(L0 + off), v = *(lds_d2*)(L0 + off + 16);
        if (halo) { Bc.x = u.x; Lc.x = u.y; Bc.y = v.x; Lc.y = v.y; }
      }
    }
    const int tbase = block_time(DIR, n, 0, T);
    int xw = 0;
#pragma unroll
    for (int tt = 0; tt < kBlk; tt++) {
      const int t = DIR == 0 ? tbase + tt : tbase - tt;
      const CT yb = ybw[2 * tt], wb = ybw[2 * tt + 1];
      const V2 E = {e0[tt], e1[tt]};
      if (tt == 4) {
        xw = *(volatile lds_int*)(exw + 4 * (n & (kHaloSlots - 1)));
        if (want_next) need_blocks(n + 2);
        load_half(n + 1, std::integral_constant<int, 0>{});
      }
      if (STEADY || t < T) {
        const bool first = !STEADY && (DIR == 0 ? t == 0 : t == T - 1);
        const V2 YB = {yb_prev, yb_prev}, WB = {wb_prev, wb_prev};
        if (DIR == 0) {
          // alpha_t[j] = (alpha[j] + r*alpha[j-1] + r^2*skip*alpha[j-2]) * y_t[l_j], ctc_loss.cpp:47-60
          if (first) {
            if (p0 == 0) { Bc.x = cond ? kOne : (CT)0; Lc.x = kOne * rr2 * E.x; }   // ctc_loss.cpp:39-42
          } else {
            const V2 PL = {from_prev_lane(Lc.y), Lc.x};         // the label cell just below each pair's blank
            const V2 Bn = __builtin_elementwise_fma(Bc, YB, PL);
            V2 tl = __builtin_elementwise_fma(Bc, WB, Lc);
            tl = __builtin_elementwise_fma(SK, PL, tl);
            Lc = tl * E; Bc = Bn;
          }
        } else {
          // q_t[j] = (q[j] + r*q[j+1] + r^2*skipn*q[j+2]) * y_t[l_j]; q = beta * emission, ctc_loss.cpp:84-99
          if (first) {
            if (cond) { if (p0 == S) Bc.x = kOne; if (p0 + 1 == S) Bc.y = kOne; }  // ctc_loss.cpp:76
            if (p0 == S - 1) Lc.x = kOne * rr2 * E.x;                               // ctc_loss.cpp:78
            if (p0 + 1 == S - 1) Lc.y = kOne * rr2 * E.y;
          } else {
            const V2 G = __builtin_elementwise_fma(WB, Bc, SK * Lc);        // what the label cell of the pair below takes
            const V2 TK = {G.y, from_next_lane(G.x)};
            const V2 Lo = Lc;
            Lc = (Lo + TK) * E;
            Bc = __builtin_elementwise_fma(Bc, YB, Lo);
          }
        }
        yb_prev = yb; wb_prev = wb;
        if (tt == 7) {
          // the frame (see the section's header): leave the high word of the largest cell, remove what the frame wave decided
          const int m01 = max(X::top(Bc.x), X::top(Lc.x)), m23 = max(X::top(Bc.y), X::top(Lc.y));
          *(volatile lds_int*)(mxl + (n & (kHaloSlots - 1)) * (X::kMaxW * 256)) = max(m01, m23);
          xw = __builtin_amdgcn_readfirstlane(xw);
          if ((xw >> 12) != n) {
            PROF_SPIN_BEGIN
            int spins = 0;
            do {
              __builtin_amdgcn_s_sleep(1);
              xw = __builtin_amdgcn_readfirstlane(*(volatile lds_int*)(exw + 4 * (n & (kHaloSlots - 1))));
              if (++spins > (1 << 20)) { atomicOr(&p.flags[b], 128); xw = (n << 12) | 2048; }
            } while ((xw >> 12) != n);
            PROF_SPIN_END(prof_lag)
          }
          const int ex = (xw & 0xfff) - 2048;
          Bc.x = X::scale(Bc.x, -ex); Bc.y = X::scale(Bc.y, -ex); Lc.x = X::scale(Lc.x, -ex); Lc.y = X::scale(Lc.y, -ex);
          e_total += ex;
          const int kk = DIR == 0 ? (t + 1) : t;            // alpha row 16k-1 / beta row 16k -> slot k
          if (CK >= 0 ? CK == 1 : ((kk & (kSeg - 1)) == 0 && kk > 0 && kk < T)) {
            // the true cells (blank with its emission, label without the tilt); the frame wave converts and stores them
            // (two buffers: the checkpoint wave has 16 steps for each and is normally long done with the row before last)
            if (nck >= 2) HALO_WAIT(__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(L0 + hl.ckdone + 4 * DIR)) >= nck - 1);
            nck++;
            const V2 cb = Bc * yb_prev, cl = Lc * inv_rr;
            if (CK < 0) ckbuf = (kk / kSeg) & 1;
            store4(hl.ckb + (((DIR * 2 + ckbuf) * X::kMaxW + w) * 64 + lane) * kLane4, cb.x, cl.x, cb.y, cl.y);
            ckbuf ^= 1;
          }
        }
      }
    }
    if (PAR >= 0 ? (PAR == 1 && has_down) : ((n & 1) && has_down)) {
      const bool edge = DIR == 0 ? lane >= 64 - kHfHalo : lane < kHfHalo;
      if (edge)
        store4(hl.bnd + (((DIR * X::kMaxW + w) * kHaloSlots + (n & (kHaloSlots - 1))) * kHfHalo + (lane & (kHfHalo - 1))) * kLane4,
               Bc.x, Lc.x, Bc.y, Lc.y);
    }
    *(volatile lds_int*)(prog + 4 * w) = n + 1;
  };
  {
    typedef std::integral_constant<int, -1> Any;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    need_blocks(1);
    load_half(0, std::integral_constant<int, 0>{});
    const int steady_end = DIR == 0 ? T / kBlk : nblk;         // blocks [1, steady_end) are steady
    run_block(0, std::false_type{}, Any{}, Any{});
    int n = 1;
#ifndef E2E_F1_FAST_DIRS             // bit 0: alpha, bit 1: beta (tools/diag A/B)
#define E2E_F1_FAST_DIRS 3
#endif
#ifndef E2E_F1_PLAIN_LOOP
    // Interior blocks in pairs (even block, odd block) with everything that depends on the block's parity resolved at
    // compile time.  alpha: blocks 1 .. (T-1)/8 - 1, a checkpoint row (t = 16k-1) ends every odd block; beta: blocks
    // 1 .. M-1, M = (T-1)/8, a checkpoint row (t = 16k) ends the blocks of M's parity.
    const int fast_end = (T - 1) >> 3;
    if (X::kPairedLoop && fast_end - n >= 3 && ((E2E_F1_FAST_DIRS >> DIR) & 1)) {
      run_block(n, std::true_type{}, Any{}, Any{}); n++;      // (n = 2 now)
      if (DIR == 0 || (((T - 1) >> 3) & 1)) {
        for (; n + 1 < fast_end; n += 2) { run_block(n, std::true_type{}, I0{}, I0{}); run_block(n + 1, std::true_type{}, I1{}, I1{}); }
      } else {
        for (; n + 1 < fast_end; n += 2) { run_block(n, std::true_type{}, I0{}, I1{}); run_block(n + 1, std::true_type{}, I1{}, I0{}); }
      }
    }
#endif
    for (; n < steady_end; n++) run_block(n, std::true_type{}, Any{}, Any{});
    for (; n < nblk; n++) run_block(n, std::false_type{}, Any{}, Any{});
  }
#ifdef E2E_FAST_PROFILE
  if (lane == 0 && b < 256) { unsigned long long* g = g_prof3 + ((size_t)b * 16 + DIR * 8 + w) * 4;
    g[0] = __builtin_amdgcn_s_memtime() - prof_t0; g[1] = prof_fill; g[2] = prof_nb; g[3] = prof_lag; }
#endif
  // ---- log Z from this side ----
  if (DIR == 0) {
    double z = 0.0;
    if (owned) {
      if (p0 == S) z += (double)Bc.x * (double)yb_prev;          // ctc_loss.cpp:63-70, un-tilted relative to cell L-1
      if (p0 + 1 == S) z += (double)Bc.y * (double)yb_prev;
      if (p0 == S - 1) z += (double)Lc.x;                        // (= r * the label cell)
      if (p0 + 1 == S - 1) z += (double)Lc.y;
    }
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
    *(volatile lds_f64*)(L0 + hl.zacc + 8 * w) = z;
    *(volatile lds_int*)(prog + 4 * w) = nblk + 1;
    if (w == 0) {
      HALO_WAIT(__builtin_amdgcn_readfirstlane(lds_min8(prog)) >= nblk + 1);
      if (lane == 0) {
        double zs = 0.0;
        for (int k = 0; k < W; k++) zs += *(volatile lds_f64*)(L0 + hl.zacc + 8 * k);
        const double rr = (double)r_tilt;
        const double lz = log(zs) + (double)(e_total - bias) * 0.693147180559945309417 - (double)(L - 1) * log(rr);
        p.logz[2 * b] = lz;
        p.zt2[b] = log2(zs) + (double)(e_total - bias);
        p.losses[b] = (float)(-lz);
        if (!(zs > 0.0) || !(zs < __builtin_huge_val())) atomicOr(&p.flags[b], 4);     // infeasible or out of range
      }
    }
  } else if (w == 0 && lane == 0) {
    const double z = (cond ? (double)Bc.x * (double)yb_prev : 0.0) + (double)Lc.x;     // sum_j alpha_0[j]*beta_0[j]
    p.logz[2 * b + 1] = log(z) + (double)(e_total - bias) * 0.693147180559945309417 - (double)(L - 1) * log((double)r_tilt);
  }
}

// The checkpoint wave of a direction: the chain waves leave a checkpoint row's true cells in LDS; this wave finds the
// exponent of every group of F2PPL pairs (the segment kernel's lanes), scales, and stores cells and exponents.  (First
// given to the frame wave: that one is on the chains' critical path -- they wait for its word every block -- and the
// extra work showed up as 20-35 cycles per step of waiting.)
template <int DIR, int F2PPL, typename X>
__device__ __forceinline__ void hf_ckpt_wave(const FastParams& p, int b, int T, int S, lds_u8* L0, const HfLds hl, int lane, int W, int bias) {
  typedef typename X::T CT;
  constexpr int kLane4 = 4 * X::kElem;
  const int nblk = (T + kBlk - 1) / kBlk;
  const int nres = DIR == 0 ? T / kBlk : nblk;       // blocks whose step 7 is live (alpha's last block may be short)
  const int M = (T - 1) >> 3;
  lds_u8* prog = L0 + hl.prog + DIR * 32;
  float* ck = (DIR == 0 ? p.ckA : p.ckQ) + (size_t)b * p.NS * p.CELLS;
  const int idx = DIR == 0 ? lane - kHfHalo : lane;
  const bool own_lane = DIR == 0 ? lane >= kHfHalo : lane < kHfOwnLanes;
  int done = 0;
  for (int n = 0; n < nres; n++) {
    // the checkpoint row of this block, if it has one (alpha row 16k-1 / beta row 16k -> slot k)
    const int kk = DIR == 0 ? 8 * (n + 1) : 8 * (M - n);
    if (!((kk & (kSeg - 1)) == 0 && kk > 0 && kk < T)) continue;
    HALO_WAIT(__builtin_amdgcn_readfirstlane(lds_min8(prog)) >= n + 1);
    const int slot = kk / kSeg;
    CT cw[X::kMaxW][4];
#pragma unroll
    for (int w = 0; w < X::kMaxW; w++) {
      if (w < W) {
        const int off = hl.ckb + (((DIR * 2 + (slot & 1)) * X::kMaxW + w) * 64 + lane) * kLane4;
        if constexpr (X::kF32) { const h_f4 v = *(lds_f4*)(L0 + off); cw[w][0] = v.x; cw[w][1] = v.y; cw[w][2] = v.z; cw[w][3] = v.w; }
        else { const h_d2 u = *(lds_d2*)(L0 + off), v = *(lds_d2*)(L0 + off + 16); cw[w][0] = u.x; cw[w][1] = u.y; cw[w][2] = v.x; cw[w][3] = v.y; }
      }
    }
    *(volatile lds_int*)(L0 + hl.ckdone + 4 * DIR) = ++done;       // (LDS runs this wave's operations in order: the reads are done)
#pragma unroll
    for (int w = 0; w < X::kMaxW; w++) {
      if (w >= W) break;
      const CT* c = cw[w];
      const int p0 = kHfOwn * w + 2 * idx;
      const int ma = max(X::top(c[0]), X::top(c[1])), mb = max(X::top(c[2]), X::top(c[3]));
      int m0 = ma, m1 = mb;                                           // exponent source of pair p0 / p0 + 1
      if (F2PPL >= 2) { m0 = max(ma, mb); m1 = m0; }
      if (F2PPL >= 4) { m0 = max(m0, dpp_i<0xB1>(0, m0)); m1 = m0; }  // quad_perm [1,0,3,2]: the lane pair
      if (F2PPL >= 8) { m0 = max(m0, dpp_i<0x4E>(0, m0)); m1 = m0; }  // quad_perm [2,3,0,1]: the (aligned) four lanes
      const int own0 = X::expo(m0), own1 = X::expo(m1);
      const int st0 = m0 > 0 ? own0 - bias : -30000, st1 = m1 > 0 ? own1 - bias : -30000;     // relative to the frame
      if (own_lane && p0 < 64 * F2PPL && (p0 & ~(F2PPL - 1)) <= S) {      // (groups past the lattice are not read)
        h_f4 o;                                                       // (cells of an all-zero group stay zero whatever the exponent)
        o.x = m0 > 0 ? (float)X::scale(c[0], -own0) : 0.f; o.y = m0 > 0 ? (float)X::scale(c[1], -own0) : 0.f;
        o.z = m1 > 0 ? (float)X::scale(c[2], -own1) : 0.f; o.w = m1 > 0 ? (float)X::scale(c[3], -own1) : 0.f;
        *reinterpret_cast<h_f4*>(ck + (size_t)slot * p.CELLS + 2 * p0) = o;
        short* cke = p.ckE + (((size_t)b * p.NS + slot) * 2 + DIR) * 64;
        if (F2PPL == 1) { cke[p0] = (short)st0; cke[p0 + 1] = (short)st1; }
        else if ((p0 & (F2PPL - 1)) == 0) cke[p0 / F2PPL] = (short)st0;
      }
    }
  }
}

// Waves, in this order: 2 x kMaxW chain waves (alpha0, beta0, alpha1, beta1, ...: the first four on SIMDs 0,2,1,3), the two
// frame waves, 2 x kProducers probability-row waves (alternating alpha side / beta side), the two checkpoint waves (the
// second writes the lattice description first).
template <int PPL, typename X>
__global__ E2E_KERNEL_ALIGN __launch_bounds__(X::kWaves * 64) void ctc_fast_chain_hf_kernel(FastParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V;
  const HfLds hl = HfLds::of<X>(V);

  if (b == 0 && tid < 32) p.ctl[tid] = 0;
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  const bool bad = Tq < 1 || Tq > p.T || Sq < 0 || Sq > p.Smax;
  if (bad) {                       // the exact kernel poisons this utterance
    if (tid == 0) { p.flags[b] = 1; p.losses[b] = __builtin_nanf(""); }   // reason bit 0: bad lengths
    return;
  }
  const int T = (int)Tq, S = (int)Sq;
  constexpr int MAXW = (64 * PPL + kHfOwn - 1) / kHfOwn < X::kMaxW ? (64 * PPL + kHfOwn - 1) / kHfOwn : X::kMaxW;
  const int W = min(S / kHfOwn + 1, MAXW);                        // waves that hold a cell: pairs 0..S (pair S = the last blank)
  if (tid == 0) p.flags[b] = 0;
  for (int i = tid; i < p.MW; i += blockDim.x) p.segmask[(size_t)b * p.MW + i] = 0u;
  if (tid < 2 * kRingBlks) reinterpret_cast<int*>(smem + hl.filled)[tid] = (tid & (kRingBlks - 1)) < X::kProducers ? (tid & (kRingBlks - 1)) : 0;
  if (tid < 16) reinterpret_cast<int*>(smem + hl.prog)[tid] = (tid & 7) < W ? 0 : kHaloIdle;
  if (tid < 2 * kHaloSlots) reinterpret_cast<int*>(smem + hl.exw)[tid] = (tid & (kHaloSlots - 1)) < kHfLag ? ((tid & (kHaloSlots - 1)) << 12) | 2048 : -1;
  if (tid < 8) reinterpret_cast<double*>(smem + hl.zacc)[tid] = 0.0;
  if (tid < 2) reinterpret_cast<int*>(smem + hl.ckdone)[tid] = 0;
  for (int i = tid; i < 2 * X::kRing * kBlk; i += blockDim.x)      // the zero rows (label index V) of every block
    reinterpret_cast<typename X::R*>(smem + hl.ring + (i / kBlk) * hl.blk_bytes)[V * X::kRowElems + (i % kBlk)] = 0;
  __syncthreads();

  constexpr int kChains = 2 * X::kMaxW, kFrame = kChains, kProd = kChains + 2, kCkpt = kProd + 2 * X::kProducers;
  const int wave = __builtin_amdgcn_readfirstlane(wid);
  lds_u8* L0 = (lds_u8*)smem;
  const int bias = X::bias(fast_tilt(S, T));
  if (wave < kChains) {
    const int d = wave & 1, w = wave >> 1;
    if (w >= W) return;              // (holds no cell of this utterance; the segment kernel does not read past the lattice)
    if (d == 0) hf_chain_wave<0, PPL, X>(p, b, T, S, smem, hl, lane, w, W);
    else hf_chain_wave<1, PPL, X>(p, b, T, S, smem, hl, lane, w, W);
  } else if (wave == kFrame) halo_frame_wave<0, X::kF32, kHfLag, true>(p, b, T, L0, hl.prog, hl.exw, hl.mxl, X::kMaxW, lane, W, bias);
  else if (wave == kFrame + 1) halo_frame_wave<1, X::kF32, kHfLag, true>(p, b, T, L0, hl.prog, hl.exw, hl.mxl, X::kMaxW, lane, W, bias);
  else if (wave == kCkpt) hf_ckpt_wave<0, PPL, X>(p, b, T, S, L0, hl, lane, W, bias);
  else if (wave == kCkpt + 1) {
    cellinfo_wave<PPL>(p, b, T, S, reinterpret_cast<int*>(smem + hl.sortcnt), lane);
    hf_ckpt_wave<1, PPL, X>(p, b, T, S, L0, hl, lane, W, bias);
  } else {
    const int d = (wave - kProd) & 1;                    // alternating: alpha rows, beta rows
    const int first = (wave - kProd) >> 1;               // the producers of a direction take every kProducers-th block
    lds_u8* prog = L0 + hl.prog + d * 32;
    const double rr2 = (double)X::tilt2(fast_tilt(S, T));        // (the chain waves' own expression)
    unsigned char* ring = smem + hl.ring + d * X::kRing * hl.blk_bytes;
    volatile int* fl = reinterpret_cast<int*>(smem + hl.filled) + d * kRingBlks;
    constexpr int MODE = X::kF32 ? 2 : 1;
    if constexpr (X::kBigV) {
      if constexpr (X::kMaxW > 2) {                          // ChainF64LW
        if (V <= 288) prep_wave_big<36, X::kRing, 1>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, lane, prog, rr2);
        else prep_wave_big<56, X::kRing, 1>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, lane, prog, rr2);
      } else if (V <= 128) prep_wave_big<16, X::kRing>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, lane, prog, rr2);
      else if (V <= 176) prep_wave_big<22, X::kRing>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, lane, prog, rr2);
      else prep_wave_big<28, X::kRing>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, lane, prog, rr2);
    } else
    if (V <= 16) prep_wave<2, MODE>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, nullptr, lane, prog, rr2);
    else if (V <= 32) prep_wave<4, MODE>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, nullptr, lane, prog, rr2);
    else if (V <= 48) prep_wave<6, MODE>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, nullptr, lane, prog, rr2);
    else if (V <= 64) prep_wave<8, MODE>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, nullptr, lane, prog, rr2);
    else prep_wave<12, MODE>(p, b, T, d, first, X::kProducers, ring, hl.blk_bytes, fl, nullptr, lane, prog, rr2);
  }
}

// The probabilities of every live frame for ChainF64W (see prep_wave_big): a wave takes kProbRows consecutive frames at once
// (their loads are all in flight together: one frame per wave left the kernel latency-bound at 3.3 TB/s), up to four columns per
// lane; fused log-softmax for raw logits (ctc_loss.cpp reads log-probabilities: CTCLoss applies log_softmax first).
#ifndef E2E_PROB_ROWS              // (tools/diag; round 5, whole call at V = 200 / 448: 4 rows 313 / 959 us, 8 rows 328 / 980, 16 rows 401 / 1 053)
#define E2E_PROB_ROWS 4
#endif
constexpr int kProbRows = E2E_PROB_ROWS;
template <int NK>                  // columns per lane: 4 (<= kMaxBigV) or 7 (<= kMaxHugeV)
__global__ __launch_bounds__(256) void ctc_fast_prob_kernel(FastParams p) {
  const int lane = threadIdx.x & 63, V = p.V;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kProbRows, nrows = (int64_t)p.B * p.T;
  if (row0 >= nrows) return;
  const float ninf = -__builtin_huge_valf();
  float x[kProbRows][NK];
  bool live[kProbRows];
#pragma unroll
  for (int r = 0; r < kProbRows; r++) {
    const int64_t row = row0 + r < nrows ? row0 + r : nrows - 1;
    int b, t; split_frame(row, p.T, b, t);
    const int64_t Tq = p.x_len[b];
    live[r] = row0 + r < nrows && Tq >= 1 && Tq <= p.T && t < Tq;
    const int64_t xr = (int64_t)b * p.sB + (int64_t)t * p.sT;
#pragma unroll
    for (int k = 0; k < NK; k++) { const int v = lane + 64 * k; x[r][k] = v < V ? load_elem(p.x, xr + (int64_t)v * p.sV, p.xdt) : ninf; }
  }
#pragma unroll
  for (int r = 0; r < kProbRows; r++) {
    if (!live[r]) continue;                                  // (wave-uniform)
    float y[NK];
    if (p.logprobs) {
#pragma unroll
      for (int k = 0; k < NK; k++) y[k] = (x[r][k] > ninf && x[r][k] < -78.f) ? kTinyProb : exp_le0(x[r][k]);
    } else {
      float m = x[r][0];
#pragma unroll
      for (int k = 1; k < NK; k++) m = fmaxf(m, x[r][k]);
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float ssum = 0.f;
#pragma unroll
      for (int k = 0; k < NK; k++) {
        // x - m with its rounding error (two-sum): at |x - m| up to 80 half an ulp of the difference is 3.8e-6 -- 4e-6 relative in
        // the probability, and alignments that compete for a frame differ in dozens of such factors (7e-6 on single gradient
        // elements at logits of scale 8..12, tools/diag/ext_case_row.py).  This kernel waits for HBM: the six operations are free.
        const float a = x[r][k], d = a - m, bb = d - a;
        const float err = (a - (d - bb)) + (-m - bb);
        const float e0 = exp_le0(d);
        y[k] = a > ninf ? fmaf(e0, err, e0) : 0.f;
        ssum += y[k];
      }
      ssum = wave_sum(ssum);
      float inv = __builtin_amdgcn_rcpf(ssum);
      inv = fmaf(fmaf(-ssum, inv, 1.0f), inv, inv);        // one Newton step: ~0.5 ulp
#pragma unroll
      for (int k = 0; k < NK; k++) y[k] = (x[r][k] > ninf && x[r][k] - m < -78.f) ? kTinyProb : y[k] * inv;
    }
    float* yrow = p.ytab + (size_t)(row0 + r) * V;
#pragma unroll
    for (int k = 0; k < NK; k++) { const int v = lane + 64 * k; if (v < V) yrow[v] = y[k]; }
  }
}

// ============================================================================================
// F2: one wave per (utterance, 16-step segment)
// ============================================================================================
#ifndef E2E_F2_ABL                  // tools/diag: timing builds with parts of the segment kernel switched off (results meaningless)
#define E2E_F2_ABL 0
#endif
// the segment kernel's parameters with the gradient's element width as a compile-time fact (an instance per width: three
// copies of the gradient passes behind a run-time test cost the f32 kernel 80 bytes of scratch and 5 us)
// (BIG: alphabets of 97..224 columns -- the probability tile is staged in two rounds, a gradient lane takes up to four labels)
template <bool O16, bool BIG = false> struct SegParams : FastParams { static constexpr bool kOut16 = O16, kBigV = BIG; };
constexpr int kHalf = 8;           // rows of alpha*beta buffered in LDS before they are summed and written out
// Between the segment wave's LDS phases (scatter -> scan -> per-label reads -> next half's scatter).  The LDS executes one
// wave's operations in order, so a read issued after a write of the same wave sees it without a wait; only the compiler has
// to keep the order.  (E2E_F2_DRAIN: the full drain this used to be, for A/B.)
#if defined(E2E_F2_DRAIN)
#define F2_LDS_ORDER asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(E2E_F2_NOFENCE)
#define F2_LDS_ORDER
#else
#define F2_LDS_ORDER asm volatile("" ::: "memory");
#endif
#ifndef E2E_F2_HALF                 // four pairs per lane: keep eight of the segment's alpha rows and compute the other eight twice
#define E2E_F2_HALF 0              // (three waves per SIMD, 92 bytes of scratch, 8 more alpha steps: the kernel takes 61.8 - 63.0 us
#endif                             //  against 61.3 with all sixteen rows kept: not on)
#ifndef E2E_SMIN                    // (both overridable for tools/diag experiments)
#define E2E_SMIN 0x1p-120f         // smallest row sum sum_j alpha*beta the gradient rows are trusted with
#endif
#ifndef E2E_ZTOL
#define E2E_ZTOL 4e-6f
#endif
constexpr float kZTol = E2E_ZTOL;   // |log2| tolerance of the rows' self-check (2.8e-6 relative; rounding alone stays below 1e-6)
#ifndef E2E_ZTOL_F32
#define E2E_ZTOL_F32 1e-5f
#endif
constexpr float kZTolF32 = E2E_ZTOL_F32;   // the same with f32 chains (ctc_fast_chain_hf_kernel), whose own rounding reaches 3e-6
#ifndef E2E_KYS
#define E2E_KYS (kSeg + 4)
#endif
constexpr int kYs = E2E_KYS;      // row stride (floats) of the transposed probability tile: 80 B spreads the
                                   // 16-byte gathers of different labels over the LDS bank row

template <int PPL>
struct F2Lds {
  // label cells in label order; every row is preceded by 4 pad floats whose last one stays 0: the prefix sum "before
  // slot 0", so that a label's posterior is pre[hi-1] - pre[lo-1] without a case distinction
  static constexpr int PROW = 64 * PPL + 4;
  float* Ps;        // [kHalf][PROW], Ps[k*PROW - 1] == 0
  float* ys;        // [V+1][kYs]   probabilities of the segment, TRANSPOSED (label-major, 16 steps + pad), row V = 0
  int* starts;      // [130] first sorted slot of every label (V+1 entries used)
  __device__ F2Lds(unsigned char* smem, int V) {
    Ps = reinterpret_cast<float*>(smem) + 4;
    ys = Ps + kHalf * PROW;
    starts = reinterpret_cast<int*>(ys + kYs * (V + 1));
  }
  __host__ __device__ static size_t bytes(int V) { return sizeof(float) * (4 + kHalf * PROW + kYs * (V + 1)) + sizeof(int) * (V > 127 ? lstart_ints(V) : 130); }
};

// The gradient rows are written one lane per (row, label): with V <= 32 columns the 64 lanes cover two rows per pass (four
// with V <= 16), beyond 64 columns a lane takes two labels.  What a lane needs about its label is the same for all 16
// rows of the segment and is looked up once.
template <int NSET>
struct GradLanesT {
  static constexpr int kSets = NSET;
  int hi[NSET], lo[NSET];     // float index into Ps of the label's last sorted slot / of the slot before its first (+ the lane's row)
  int y[NSET];          // float index of the label's row in the transposed probability tile (+ the lane's row)
  float isblank[NSET];  // 1 for the blank column (it takes the pre-summed blank cells), else 0
  int rpp;              // rows per pass: 1, 2 or 4 (wave-uniform)
  int rsel;             // this lane's row inside a pass
  int goff;             // rsel * V + v: the lane's offset inside a pass's rows of the gradient; -1: the lane writes nothing
  template <int PPL>
  __device__ void init(const F2Lds<PPL>& lds, int V, int blank, int lane) {
    rpp = V <= 16 ? 4 : V <= 32 ? 2 : 1;
    const int vp = 64 / rpp;
    rsel = rpp == 1 ? 0 : lane / vp;
    const int v0 = rpp == 1 ? lane : lane & (vp - 1);
    goff = v0 < V ? rsel * V + v0 : -1;
#pragma unroll
    for (int s = 0; s < NSET; s++) {
      const int v = min(v0 + 64 * s, V - 1);
      hi[s] = lds.starts[v + 1] - 1 + rsel * F2Lds<PPL>::PROW; lo[s] = lds.starts[v] - 1 + rsel * F2Lds<PPL>::PROW;
      y[s] = v * kYs + rsel; isblank[s] = v == blank ? 1.f : 0.f;
    }
  }
};

// rows [h*8, h*8+8) of the segment: per-label sums, normaliser, gradient rows.  FULL: all 8 rows are live.
typedef GradLanesT<2> GradLanes;        // up to 128 columns (kMaxSmallV); four sets: kMaxBigV, seven: kMaxHugeV

template <int PPL, bool FULL, typename P, typename GL>
__device__ __forceinline__ void finish_rows(const P& p, int b, int t0, int n, int h, const F2Lds<PPL>& lds,
                                            const GL& gl, const float (&pb)[kHalf], int lane, float& smin, float& smax,
                                            int u_lo, int u_hi, float zfrac) {
  constexpr int PROW = F2Lds<PPL>::PROW;
  const int V = p.V;
  const int rows = FULL ? kHalf : min(kHalf, n - h * kHalf);      // live rows of this half (>= 1)
  F2_LDS_ORDER
  // prefix sums over the label-sorted cells, totals, s_t = sum_j alpha_t[j]*beta_t[j]
  float btot8[kHalf];                     // blank cells: the lanes' partial sums never went through the LDS
  wave_sum8(pb, btot8, lane);
  float st8[kHalf];                       // (wave-uniform)
  if (E2E_F2_ABL & 1) {
#pragma unroll
    for (int k = 0; k < kHalf; k++) st8[k] = btot8[k] + 1.f;
  } else {
    float c[kHalf][PPL], inc[kHalf];
#pragma unroll
    for (int k = 0; k < kHalf; k++) {
#pragma unroll
      for (int r = 0; r < PPL; r++) c[k][r] = lds.Ps[k * PROW + PPL * lane + r];
    }
#pragma unroll
    for (int r = 1; r < PPL; r++) {
#pragma unroll
      for (int k = 0; k < kHalf; k++) c[k][r] += c[k][r - 1];
    }
#pragma unroll
    for (int k = 0; k < kHalf; k++) inc[k] = c[k][PPL - 1];
    wave_scan8(inc);
#pragma unroll
    for (int k = 0; k < kHalf; k++) {
      const float excl = inc[k] - c[k][PPL - 1];
      if constexpr (PPL % 2 == 0) {
        // (two slots per packed add: the lane's slots sit in aligned register pairs as the 16-byte read left them)
        typedef float pk2 __attribute__((ext_vector_type(2)));
        const pk2 e2 = {excl, excl};
#pragma unroll
        for (int r = 0; r < PPL; r += 2) {
          pk2 v = {c[k][r], c[k][r + 1]};
          v += e2;
          lds.Ps[k * PROW + PPL * lane + r] = v.x; lds.Ps[k * PROW + PPL * lane + r + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int r = 0; r < PPL; r++) lds.Ps[k * PROW + PPL * lane + r] = c[k][r] + excl;
      }
      st8[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc[k]), 63)) + btot8[k];
    }
  }
  // Self-check of a row: sum_j alpha_t[j] beta_t[j] is the same number at every t -- the tilted partition sum the
  // chains ended with (zt2).  st is that sum in this segment's unit, u the exponent of the unit (minus zt2's integer
  // part): log2(st) + u must equal zt2's fraction.  Cells that mattered but were flushed -- too few bits in the f32
  // checkpoints, a recomputed row sinking below the lane's unit -- only ever LOWER the sum, so the deviation bounds
  // the posterior mass the row lost.  (Rounding alone: < 1e-6, measured over the fuzz sweeps.)  Lane k checks row k.
  if (!(E2E_F2_ABL & 4)) {
    float my_st = 1.f;
#pragma unroll
    for (int k = 0; k < kHalf; k++) my_st = lane == k ? st8[k] : my_st;
    const int u = lane == kHalf - 1 ? u_hi : u_lo;
    const float dev = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(my_st)) - zfrac + (float)(__builtin_amdgcn_frexp_expf(my_st) + u);
    const bool live = lane < rows;
#ifdef E2E_FAST_PROFILE
    { const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
      if (wg < 16384 && live) atomicMax(reinterpret_cast<int*>(&g_zdev[wg]), __float_as_int(fminf(fabsf(dev), 1e30f))); }
#endif
    // (smin / smax: see the range check at the end of the kernel)
    const bool bad = live && !(fabsf(dev) <= p.ztol && my_st >= E2E_SMIN);
    if (__any(bad)) smin = 0.f;
    // borderline: within the tolerance, but beyond what rounding alone leaves (< 1e-6).  Not a reason to flag the utterance;
    // if OTHER rows flag it for range, the f64 redo takes this segment along (smin == 1 marks it: see the kernel's end)
    else if (__any(live && !(fabsf(dev) <= 0.25f * p.ztol))) smin = fminf(smin, 1.f);
    if (__any(live && !(my_st < __builtin_huge_valf()))) smax = __builtin_huge_valf();
  }
  F2_LDS_ORDER
  F2_STAMP(5)
  // gradient rows: y - posterior (d loss/d logits in fused mode; exp(lp) - posterior otherwise).  Lane v writes column
  // v of every row: its label's slots, its row of the probability tile and whether it is the blank are the same for
  // all rows (GradLanes); per row that leaves three LDS reads at constant offsets, three VALU operations and a store.
  const size_t g0 = ((size_t)b * p.T + t0 + h * kHalf) * V;
  // (the gradient has the logits' dtype: f32, or -- P::kOut16 -- one of the two 16-bit types, chosen per store without a branch)
  const bool out_bf16 = p.xdt == E2E_BF16;
  auto put = [&](size_t idx, float g) {
    if constexpr (!P::kOut16) reinterpret_cast<float*>(p.grads)[g0 + idx] = g;
    else {
      const unsigned short hb = __builtin_bit_cast(unsigned short, (bf16_t)g), hh = __builtin_bit_cast(unsigned short, (f16_t)g);
      reinterpret_cast<unsigned short*>(p.grads)[g0 + idx] = out_bf16 ? hb : hh;
    }
  };
  if (!(E2E_F2_ABL & 2)) {
    // RPP rows per pass: the lane's row is k + rsel; the rows' normalisers and blank sums are wave-uniform, the lane picks its row's
    auto pass = [&](auto rpp_tag) {
      constexpr int RPP = decltype(rpp_tag)::value;
      const float* pre_hi = lds.Ps + gl.hi[0];
      const float* pre_lo = lds.Ps + gl.lo[0];
      const float* yrow = lds.ys + gl.y[0] + h * kHalf;
#pragma unroll
      for (int k = 0; k < kHalf; k += RPP) {
        if (FULL || k < rows) {
          float st = st8[k], bt = btot8[k];
#pragma unroll
          for (int j = 1; j < RPP; j++) { st = gl.rsel == j ? st8[k + j] : st; bt = gl.rsel == j ? btot8[k + j] : bt; }
          const float pv = (pre_hi[k * PROW] - pre_lo[k * PROW]) + gl.isblank[0] * bt;
          const float g = (yrow[k] - pv * __builtin_amdgcn_rcpf(st)) * p.gscale;
          if (gl.goff >= 0 && (FULL || k + gl.rsel < rows)) put((size_t)k * V + gl.goff, g);
        }
      }
    };
    if (gl.rpp == 2) pass(std::integral_constant<int, 2>{});
    else if (gl.rpp == 4) pass(std::integral_constant<int, 4>{});
    else {
      const int nsets = (V + 63) >> 6;
#pragma unroll
      for (int s = 0; s < GL::kSets; s++) {
        if (s < nsets) {
          const int v = lane + 64 * s;
          const float* pre_hi = lds.Ps + gl.hi[s];
          const float* pre_lo = lds.Ps + gl.lo[s];
          const float* yrow = lds.ys + gl.y[s] + h * kHalf;
#pragma unroll
          for (int k = 0; k < kHalf; k++) {
            if (FULL || k < rows) {
              const float pv = (pre_hi[k * PROW] - pre_lo[k * PROW]) + gl.isblank[s] * btot8[k];
              const float g = (yrow[k] - pv * __builtin_amdgcn_rcpf(st8[k])) * p.gscale;
              if (v < V && (!(E2E_F2_ABL & 64) || g == 1234.5f)) put((size_t)k * V + v, g);
            }
          }
        }
      }
    }
  }
  F2_LDS_ORDER   // Ps is rewritten by the next half
}

// what a segment needs from F1's workspace besides the probabilities; requested before the tile is staged so that
// the two round trips overlap
template <int PPL>
struct SegIn {
  float a[2 * PPL];                 // alpha checkpoint row (this lane's cells), segment > 0
  int ownA;                         // its per-lane exponent
  float q[2 * PPL];                 // beta-with-emission checkpoint row at the segment's end (not the last segment)
  int ownB;
  int eA7, eA15, eB0, eB8;          // rescale exponents inside the segment
  int EA0, EB16;                    // what the chains had removed in total: alpha before step t0, beta down to step t0+16
  float zfrac; int zint;            // the chains' log2 of the tilted partition sum (zt2), split: zint + zfrac
  __device__ void load(const FastParams& p, int b, int seg, int S, int lane) {
    const int t0 = seg * kSeg;
    const bool in_lattice = lane * 2 * PPL <= 2 * S;      // (the chains store no cells past the lattice's 2S+1)
    const int* cA = p.cumA + (size_t)b * p.NB + (t0 >> 3);
    const int* cB = p.cumB + (size_t)b * p.NB + (t0 >> 3);
    const int* tA = p.trkA + (size_t)b * p.NB + (t0 >> 3);
    const int* tB = p.trkB + (size_t)b * p.NB + (t0 >> 3);
    const int a0 = cA[0], b2 = cB[2];
    const int ta0 = tA[0], ta1 = tA[1], ta2 = tA[2], tb0 = tB[0], tb1 = tB[1], tb2 = tB[2];
    eA7 = ta1 - ta0; eA15 = ta2 - ta1; eB0 = tb0 - tb1; eB8 = tb1 - tb2;
    EA0 = a0; EB16 = b2;
    { const double z = p.zt2[b]; const double zi = floor(z); zfrac = (float)(z - zi);
      zint = (int)fmax(fmin(zi, 1e9), -1e9); if (!(z == z)) zfrac = z; }      // (infeasible: -inf; never passes the check)
    ownA = 0;
#pragma unroll
    for (int k = 0; k < 2 * PPL; k++) a[k] = 0.f;
    if (seg > 0) {
      ownA = -30000;
      if (in_lattice) {
        const float* src = p.ckA + ((size_t)b * p.NS + seg) * p.CELLS + lane * 2 * PPL;
#pragma unroll
        for (int k = 0; k < 2 * PPL; k++) a[k] = src[k];
        ownA = p.ckE[(((size_t)b * p.NS + seg) * 2 + 0) * 64 + lane];
      }
    }
    ownB = 0;
#pragma unroll
    for (int k = 0; k < 2 * PPL; k++) q[k] = 0.f;
    if (seg + 1 < p.NS) {            // (row seg+1 exists; whether it is meaningful depends on the utterance's length)
      ownB = -30000;
      if (in_lattice) {
        const float* src = p.ckQ + ((size_t)b * p.NS + seg + 1) * p.CELLS + lane * 2 * PPL;
#pragma unroll
        for (int k = 0; k < 2 * PPL; k++) q[k] = src[k];
        ownB = p.ckE[(((size_t)b * p.NS + seg + 1) * 2 + 1) * 64 + lane];
      }
    }
  }
};

// A label cell's slot in a row of the label-sorted products, kept as the LDS byte address of the slot in row 0: the compiler
// recomputed `base + (slot << 2)` for every store (64 per segment) rather than keep a second register per slot.
__device__ __forceinline__ int ps_slot_address(float* Ps, int slot) {           // LDS byte address of Ps[slot]
  typedef __attribute__((address_space(3))) float lds_f32;
  return (int)(unsigned)(size_t)((lds_f32*)Ps + slot);
}
__device__ __forceinline__ void ps_put(int slot_address, int row_floats, float v) {
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  typedef __attribute__((address_space(3))) float lds_f32;
  *(lds_f32*)((lds_byte*)(size_t)(unsigned)slot_address + 4 * row_floats) = v;    // (the row is the instruction's immediate)
}

// FULL: an interior segment (16 live steps, neither t = 0 nor t = T-1 inside): no guards in the loops.
template <int PPL, bool FULL, typename P, typename GL>
__device__ __forceinline__ void segment_body(const P& p, int b, int seg, int T, int S, int n,
                                             const LaneCells<PPL>& lc, const int (&rank)[PPL],
                                             const SegIn<PPL>& in,
                                             const F2Lds<PPL>& lds, const GL& gl, int lane, float& smin, float& smax) {
  constexpr int NC = 2 * PPL;
  constexpr int kSlope = 3 * NC;    // exponent drop allowed per lane (see the alpha load below)
  constexpr int PROW = F2Lds<PPL>::PROW;
  const int blank = p.blank, L = 2 * S + 1, t0 = seg * kSeg;
  const float* ys = lds.ys;
  const float rr = lc.r;
  const bool cond = (T > 1 || L == 1);
  // the four rescale exponents that fall inside this segment (alpha at t%8 == 7, beta at t%8 == 0)
  const int eA7 = in.eA7, eA15 = in.eA15, eB0 = in.eB0, eB8 = in.eB8;

  // ---- alpha rows of the segment, kept in registers ----
  // (Keeping only 8 rows and recomputing the other 8 was tried: 168 VGPRs, three waves per SIMD -- and slower.  With
  // two waves per SIMD the kernel already saturates the VALU issue slots; the 8 extra steps cost more than the
  // third wave hides.)
  float A[kSeg][NC];
  float a[NC];
  int eA = 0;                       // this lane's exponent for its alpha cells during the segment
  if (seg == 0) {
#pragma unroll
    for (int k = 0; k < NC; k++) a[k] = 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < NC; k++) a[k] = in.a[k];
    const int own = in.ownA;
    // Mass flows from lane n-1 to lane n and can cross 2 cells per step, i.e. 32/NC lanes within the segment;
    // every lane crossed multiplies the stored value by the hand-over factor 2^(eA[n-1]-eA[n]).  Limiting the
    // exponent drop to 3 bits per cell (kSlope per lane) bounds the worst growth over a segment by 2^96.  A lane
    // whose own cells lie further below its left neighbour than that takes the neighbour's unit minus kSlope and
    // keeps its cells as small numbers (still exact down to 2^-126 of that unit).
    // (eA_n = max_{m <= n} (own_m - kSlope*(n - m)): a prefix maximum of own_m + kSlope*m.)  Lanes whose cells are all
    // zero -- above the band's upper edge early in an utterance -- take the unit of the last non-zero lane below them
    // instead of sliding down kSlope bits per lane: the mass that reaches them within the segment fits that unit, and
    // the beta rows, which live in the reciprocal units, are not pushed 24 bits per lane towards underflow.
    const bool nz = own > -30000;
    const int pmax = wave_scan_max(nz ? own + kSlope * lane : -0x20000000);
    const int mstar = wave_scan_max(nz ? lane : -1);
    eA = mstar >= 0 ? pmax - kSlope * mstar : own;
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
  F2_STAMP(1)
  // this lane's label rows (and the blank row) in the transposed tile: one 16-byte read fetches 4 time steps
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4* ylab[PPL];
#pragma unroll
  for (int r = 0; r < PPL; r++) ylab[r] = reinterpret_cast<const f4*>(ys + lc.lab[r] * kYs);
  const f4* yblank = reinterpret_cast<const f4*>(ys + blank * kYs);
  f4 e4[PPL], b4;

#pragma unroll
  for (int tt = 0; tt < kSeg; tt++) {
    if ((tt & 3) == 0) {
      b4 = yblank[tt >> 2];
#pragma unroll
      for (int r = 0; r < PPL; r++) e4[r] = ylab[r][tt >> 2];
    }
    if (FULL || tt < n) {
      const float yb = b4[tt & 3];
      if (!FULL && t0 + tt == 0) {
#pragma unroll
        for (int k = 0; k < NC; k++) a[k] = 0.f;
        if (lane == 0) { a[0] = cond ? yb : 0.f; a[1] = rr * e4[0][tt & 3]; }
      } else {
        float pl = from_prev_lane(a[NC - 1]) * fA;
#pragma unroll
        for (int r = 0; r < PPL; r++) {
          const float ob = a[2 * r], ol = a[2 * r + 1];
          a[2 * r] = (ob + rr * pl) * yb;
          a[2 * r + 1] = (ol + rr * ob + lc.skp[r] * pl) * e4[r][tt & 3];
          pl = ol;
        }
      }
      if ((tt & 7) == 7) {             // t0 is a multiple of 16: t & 7 == tt & 7
        const int e = tt == 7 ? eA7 : eA15;
        if (e != 0) {
#pragma unroll
          for (int k = 0; k < NC; k++) a[k] = ldexpf(a[k], -e);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) A[tt][k] = a[k];
  }

  F2_STAMP(2)
  // ---- beta backwards through the segment, posteriors, per-label accumulation ----
  // beta rows live in the reciprocal units of the alpha lanes (lane n: 2^(emax - eA_n)), so that alpha*beta is
  // in one common unit across the wave; the hand-over factor from lane n+1 is then 2^(eA_n - eA_{n+1}) <= 2^kSlope
  float q[NC];
  const bool last_seg = (t0 + n == T);
  float end_unit = 1.f;
  int unit_exp = 0;                 // alpha*beta products of this segment are in units of 2^unit_exp (chains' units)
  if (FULL || !last_seg) {
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = in.q[k];
    const int ownB = in.ownB;
    // The common unit of alpha*beta is taken where the two rows actually meet: the alpha row at the segment's LAST
    // step (it is in registers) against the beta checkpoint of the same time.  (eA alone describes alpha at the
    // segment's first step; with sharply peaked rows that move two cells per step the mass sits up to four lanes
    // further on by the end, and a unit derived from the start row would be off by the full dynamic range --
    // sum alpha*beta then left f32 and the utterance was handed to the exact kernel for nothing.)
    float a_end = 0.f;
#pragma unroll
    for (int k = 0; k < NC; k++) a_end = fmaxf(a_end, A[kSeg - 1][k]);          // (this branch: n == 16)
    // (a lane whose alpha cells have sunk to the denormal range carries no precision: it must not define the unit)
    const int e_end = a_end >= 0x1p-120f ? (int)((__float_as_uint(a_end) >> 23) & 0xffu) - 127 : -200;
    const int emax = wave_max(eA + ownB + e_end);
    unit_exp = emax;
    const int want = eA + ownB - emax;
    const int sh = min(max(want, -200), 120);
    // (a lane whose beta cells would need more than 2^120 in this unit while it still holds beta mass: out of range)
    float q_any = 0.f;
#pragma unroll
    for (int k = 0; k < NC; k++) q_any = fmaxf(q_any, q[k]);
    if (__any(want > 120 && q_any > 0.f)) smax = __builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = ldexpf(q[k], sh);
  } else {
#pragma unroll
    for (int k = 0; k < NC; k++) q[k] = 0.f;
    const int eA_ref = __shfl(eA, (L - 1) / NC, 64);          // lane holding cell L-1
    end_unit = ldexpf(1.f, max(min(eA - eA_ref, 126), -126));
    unit_exp = eA_ref;
  }
  F2_STAMP(3)
#pragma unroll
  for (int h = kSeg / kHalf - 1; h >= 0; h--) {
    if (!FULL && h * kHalf >= n) continue;
    float pb[kHalf];                 // this lane's blank-cell part of sum alpha*beta, per row of the half
#pragma unroll
    for (int k = 0; k < kHalf; k++) pb[k] = 0.f;
#pragma unroll
    for (int k = kHalf - 1; k >= 0; k--) {
      const int tt = h * kHalf + k;
      if ((tt & 3) == 3) {
        b4 = yblank[tt >> 2];
#pragma unroll
        for (int r = 0; r < PPL; r++) e4[r] = ylab[r][tt >> 2];
      }
      if (FULL || tt < n) {
        const int t = t0 + tt;
        const float yb = b4[tt & 3];
        float bs[NC];                 // beta_t[j] (no emission at t)
        if (!FULL && t == T - 1) {
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
        // alpha*beta of this lane's cells: label cells go to their label-sorted slot, blank cells are pre-summed
        float pblank = 0.f;
#pragma unroll
        for (int r = 0; r < PPL; r++) {
          pblank += A[tt][2 * r] * bs[2 * r];
          ps_put(rank[r], k * PROW, A[tt][2 * r + 1] * bs[2 * r + 1]);
        }
        pb[k] = pblank;
        // q_t = beta_t * y_t
#pragma unroll
        for (int r = 0; r < PPL; r++) {
          q[2 * r] = bs[2 * r] * yb;
          q[2 * r + 1] = bs[2 * r + 1] * e4[r][tt & 3];
        }
        if ((tt & 7) == 0) {
          const int e = tt == 0 ? eB0 : eB8;
          if (e != 0) {
#pragma unroll
            for (int kk = 0; kk < NC; kk++) q[kk] = ldexpf(q[kk], -e);
          }
        }
      }
    }
    F2_STAMP(4)
    // exponent of the rows' unit in the chains' terms: what alpha had removed by the row (rescales at tt = 7, 15) plus
    // what beta had removed down to it (at tt = 8, 0 -- after the row's product), minus the integer part of zt2
    const int ea_lo = in.EA0 + (h ? eA7 : 0), eb = in.EB16 + (h ? 0 : eB8);
    const int u_lo = unit_exp + ea_lo + eb - in.zint, u_hi = u_lo + (h ? eA15 : eA7);
    finish_rows<PPL, FULL>(p, b, t0, n, h, lds, gl, pb, lane, smin, smax, u_lo, u_hi, in.zfrac);
    F2_STAMP(6)
  }
}


// The segment body for four pairs per lane with the recurrences in PACKED f32 (v_pk_fma_f32 / v_pk_mul_f32).  A lane's
// pairs 0..3 are held as two groups, A = pairs (0, 2) and B = pairs (1, 3), each as packed registers of blank and label
// cells: BA = (B0, B2), LA = (L0, L2), BB = (B1, B3), LB = (L1, L3).  With that split the neighbour a pair takes from is,
// for one of the groups, simply the other group's register (alpha: pairs 1, 3 take from pairs 0, 2; beta: pairs 0, 2 take
// from pairs 1, 3), and for the other group one DPP move and one plain move assemble it.  The emission factors stay
// scalar multiplies (a lane's labels sit in different rows of the probability tile).  Same arithmetic as segment_body,
// cell for cell; 14 + 21 vector instructions per row instead of 22 + 30.
template <bool FULL, typename P, typename GL>
__device__ __forceinline__ void segment_body_pk(const P& p, int b, int seg, int T, int S, int n,
                                                const LaneCells<4>& lc, const int (&rank)[4], const SegIn<4>& in,
                                                const F2Lds<4>& lds, const GL& gl, int lane, float& smin, float& smax) {
  constexpr int PPL = 4, NC = 8;
  constexpr int kSlope = 3 * NC;
  constexpr int PROW = F2Lds<PPL>::PROW;
  const int blank = p.blank, L = 2 * S + 1, t0 = seg * kSeg;
  const float* ys = lds.ys;
  const float rr = lc.r;
  const bool cond = (T > 1 || L == 1);
  const int eA7 = in.eA7, eA15 = in.eA15, eB0 = in.eB0, eB8 = in.eB8;
  const h_f2 RR = {rr, rr};
  const h_f2 SKPA = {lc.skp[0], lc.skp[2]}, SKPB = {lc.skp[1], lc.skp[3]};
  const h_f2 SKNA = {lc.skn[0], lc.skn[2]}, SKNB = {lc.skn[1], lc.skn[3]};
  // a power-of-two rescale of the lane's eight cells: one packed multiply per register while the factor is a normal f32
  // (branch-free, so that a half's rows stay one basic block: 2^-e as two factors, exact for every |e| <= 252 -- beyond
  //  that an f32 cell is 0 or infinite either way)
  auto rescale = [](h_f2& x0, h_f2& x1, h_f2& x2, h_f2& x3, int e) {
    e = max(min(e, 252), -252);
    const int e1 = e / 2, e2 = e - e1;
    const float f1 = __int_as_float((127 - e1) << 23), f2 = __int_as_float((127 - e2) << 23);
    const h_f2 F1 = {f1, f1}, F2 = {f2, f2};
    x0 = (x0 * F1) * F2; x1 = (x1 * F1) * F2; x2 = (x2 * F1) * F2; x3 = (x3 * F1) * F2;
  };

  // ---- alpha rows of the segment, kept in registers ----
  // E2E_F2_HALF: only the eight rows of the half that beta is walking are kept -- the sweep below keeps rows 8..15, and rows
  // 0..7 are computed a second time before beta reaches them: 64 registers less (three waves per SIMD) for 8 more alpha steps
  constexpr int kKeep = E2E_F2_HALF ? kHalf : kSeg;
  h_f2 ABA[kKeep], ALA[kKeep], ABB[kKeep], ALB[kKeep];
  h_f2 BA = {0.f, 0.f}, LA = {0.f, 0.f}, BB = {0.f, 0.f}, LB = {0.f, 0.f};
  int eA = 0, shA = 0;
  if (seg != 0) {
    const int own = in.ownA;
    const bool nz = own > -30000;                       // (see segment_body for the slope-limited lane units)
    const int pmax = wave_scan_max(nz ? own + kSlope * lane : -0x20000000);
    const int mstar = wave_scan_max(nz ? lane : -1);
    eA = mstar >= 0 ? pmax - kSlope * mstar : own;
    shA = max(own - eA, -200);
  }
  auto alpha_start = [&]() {
    if (seg != 0) {
      BA.x = ldexpf(in.a[0], shA); LA.x = ldexpf(in.a[1], shA); BB.x = ldexpf(in.a[2], shA); LB.x = ldexpf(in.a[3], shA);
      BA.y = ldexpf(in.a[4], shA); LA.y = ldexpf(in.a[5], shA); BB.y = ldexpf(in.a[6], shA); LB.y = ldexpf(in.a[7], shA);
    } else { BA = h_f2{0.f, 0.f}; LA = BA; BB = BA; LB = BA; }
  };
  alpha_start();
  float fA, fB;
  {
    const int ep = __shfl_up(eA, 1, 64), en = __shfl_down(eA, 1, 64);
    fA = lane > 0 ? ldexpf(1.f, max(min(ep - eA, 126), -126)) : 0.f;
    fB = lane < 63 ? ldexpf(1.f, max(min(eA - en, 126), -126)) : 0.f;
  }
  F2_STAMP(1)
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4* ylab[PPL];
#pragma unroll
  for (int r = 0; r < PPL; r++) ylab[r] = reinterpret_cast<const f4*>(ys + lc.lab[r] * kYs);
  const f4* yblank = reinterpret_cast<const f4*>(ys + blank * kYs);
  f4 e4[PPL], b4;

  auto alpha_rows = [&](auto first_tag, auto last_tag, auto keep_tag) {
    constexpr int T0 = decltype(first_tag)::value, T1 = decltype(last_tag)::value, KEEP0 = decltype(keep_tag)::value;
#pragma unroll
    for (int tt = T0; tt < T1; tt++) {
      if ((tt & 3) == 0) {
        b4 = yblank[tt >> 2];
#pragma unroll
        for (int r = 0; r < PPL; r++) e4[r] = ylab[r][tt >> 2];
      }
      if ((FULL || tt < n) && !(E2E_F2_ABL & 16)) {
        const float yb = b4[tt & 3];
        if (!FULL && t0 + tt == 0) {
          BA = h_f2{0.f, 0.f}; LA = BA; BB = BA; LB = BA;
          if (lane == 0) { BA.x = cond ? yb : 0.f; LA.x = rr * e4[0][tt & 3]; }
        } else {
          h_f2 PLA;                                        // the label cell below pairs 0 and 2
          PLA.x = from_prev_lane(LB.y) * fA; PLA.y = LB.x;
          const h_f2 YB = {yb, yb};
          const h_f2 nBA = (BA + RR * PLA) * YB;
          const h_f2 uA = LA + RR * BA + SKPA * PLA;
          const h_f2 nBB = (BB + RR * LA) * YB;             // (pairs 1, 3 take from pairs 0, 2)
          const h_f2 uB = LB + RR * BB + SKPB * LA;
          BA = nBA; BB = nBB;
          LA.x = uA.x * e4[0][tt & 3]; LA.y = uA.y * e4[2][tt & 3];
          LB.x = uB.x * e4[1][tt & 3]; LB.y = uB.y * e4[3][tt & 3];
        }
        if ((tt & 7) == 7) rescale(BA, LA, BB, LB, tt == 7 ? eA7 : eA15);
      }
      if (tt >= KEEP0) { ABA[tt - KEEP0] = BA; ALA[tt - KEEP0] = LA; ABB[tt - KEEP0] = BB; ALB[tt - KEEP0] = LB; }
    }
  };
  typedef std::integral_constant<int, 0> I0_; typedef std::integral_constant<int, kHalf> I8_; typedef std::integral_constant<int, kSeg> I16_;
  if (E2E_F2_HALF) alpha_rows(I0_{}, I16_{}, I8_{}); else alpha_rows(I0_{}, I16_{}, I0_{});

  F2_STAMP(2)
  // ---- beta backwards through the segment, posteriors, per-label accumulation (see segment_body) ----
  h_f2 qBA = {0.f, 0.f}, qLA = qBA, qBB = qBA, qLB = qBA;
  const bool last_seg = (t0 + n == T);
  float end_unit = 1.f;
  int unit_exp = 0;
  if (FULL || !last_seg) {
    const int ownB = in.ownB;
    float a_end = fmaxf(fmaxf(fmaxf(ABA[kKeep - 1].x, ABA[kKeep - 1].y), fmaxf(ALA[kKeep - 1].x, ALA[kKeep - 1].y)),
                        fmaxf(fmaxf(ABB[kKeep - 1].x, ABB[kKeep - 1].y), fmaxf(ALB[kKeep - 1].x, ALB[kKeep - 1].y)));
    const int e_end = a_end >= 0x1p-120f ? (int)((__float_as_uint(a_end) >> 23) & 0xffu) - 127 : -200;
    const int emax = wave_max(eA + ownB + e_end);
    unit_exp = emax;
    const int want = eA + ownB - emax;
    const int sh = min(max(want, -200), 120);
    float q_any = 0.f;
#pragma unroll
    for (int k = 0; k < NC; k++) q_any = fmaxf(q_any, in.q[k]);
    if (__any(want > 120 && q_any > 0.f)) smax = __builtin_huge_valf();
    qBA.x = ldexpf(in.q[0], sh); qLA.x = ldexpf(in.q[1], sh); qBB.x = ldexpf(in.q[2], sh); qLB.x = ldexpf(in.q[3], sh);
    qBA.y = ldexpf(in.q[4], sh); qLA.y = ldexpf(in.q[5], sh); qBB.y = ldexpf(in.q[6], sh); qLB.y = ldexpf(in.q[7], sh);
  } else {
    const int eA_ref = __shfl(eA, (L - 1) / NC, 64);          // lane holding cell L-1
    end_unit = ldexpf(1.f, max(min(eA - eA_ref, 126), -126));
    unit_exp = eA_ref;
  }
  F2_STAMP(3)
#pragma unroll
  for (int h = kSeg / kHalf - 1; h >= 0; h--) {
    if (!FULL && h * kHalf >= n) continue;
    if (E2E_F2_HALF && h == 0) { alpha_start(); alpha_rows(I0_{}, I8_{}, I0_{}); }      // rows 0..7 again, kept this time
    float pb[kHalf];
#pragma unroll
    for (int k = 0; k < kHalf; k++) pb[k] = 0.f;
#pragma unroll
    for (int k = kHalf - 1; k >= 0; k--) {
      const int tt = h * kHalf + k;
      if ((tt & 3) == 3) {
        b4 = yblank[tt >> 2];
#pragma unroll
        for (int r = 0; r < PPL; r++) e4[r] = ylab[r][tt >> 2];
      }
      if (FULL || tt < n) {
        const int t = t0 + tt;
        const float yb = b4[tt & 3];
        h_f2 bsBA, bsLA, bsBB, bsLB;                     // beta_t[j] (no emission at t)
        if (E2E_F2_ABL & 8) { bsBA = qBA; bsLA = qLA; bsBB = qBB; bsLB = qLB; }
        else if (!FULL && t == T - 1) {
          const int i0 = PPL * lane;
          bsBA.x = (2 * i0 == L - 1 && cond) ? end_unit : 0.f;           bsLA.x = (2 * i0 + 1 == L - 2) ? rr * end_unit : 0.f;
          bsBB.x = (2 * (i0 + 1) == L - 1 && cond) ? end_unit : 0.f;     bsLB.x = (2 * (i0 + 1) + 1 == L - 2) ? rr * end_unit : 0.f;
          bsBA.y = (2 * (i0 + 2) == L - 1 && cond) ? end_unit : 0.f;     bsLA.y = (2 * (i0 + 2) + 1 == L - 2) ? rr * end_unit : 0.f;
          bsBB.y = (2 * (i0 + 3) == L - 1 && cond) ? end_unit : 0.f;     bsLB.y = (2 * (i0 + 3) + 1 == L - 2) ? rr * end_unit : 0.f;
        } else {
          h_f2 NB, NL;                                   // the pair above pairs 1 and 3: pair 2, the next lane's pair 0
          NB.x = qBA.y; NB.y = from_next_lane(qBA.x) * fB;
          NL.x = qLA.y; NL.y = from_next_lane(qLA.x) * fB;
          bsLA = qLA + RR * qBB + SKNA * qLB;             // (pairs 0, 2 take from pairs 1, 3)
          bsBA = qBA + RR * qLA;
          bsLB = qLB + RR * NB + SKNB * NL;
          bsBB = qBB + RR * qLB;
        }
        // alpha*beta of this lane's cells: label cells go to their label-sorted slot, blank cells are pre-summed
        constexpr int kA = E2E_F2_HALF ? kHalf - 1 : kSeg - 1;       // (index mask of the kept rows)
        const h_f2 pbl = ABA[tt & kA] * bsBA + ABB[tt & kA] * bsBB;
        pb[k] = pbl.x + pbl.y;
        const h_f2 PA = ALA[tt & kA] * bsLA, PB = ALB[tt & kA] * bsLB;
        ps_put(rank[0], k * PROW, PA.x); ps_put(rank[1], k * PROW, PB.x);
        ps_put(rank[2], k * PROW, PA.y); ps_put(rank[3], k * PROW, PB.y);
        // q_t = beta_t * y_t
        const h_f2 YB = {yb, yb};
        qBA = bsBA * YB; qBB = bsBB * YB;
        qLA.x = bsLA.x * e4[0][tt & 3]; qLA.y = bsLA.y * e4[2][tt & 3];
        qLB.x = bsLB.x * e4[1][tt & 3]; qLB.y = bsLB.y * e4[3][tt & 3];
        if ((tt & 7) == 0) rescale(qBA, qLA, qBB, qLB, tt == 0 ? eB0 : eB8);
      }
    }
    F2_STAMP(4)
    const int ea_lo = in.EA0 + (h ? eA7 : 0), eb = in.EB16 + (h ? 0 : eB8);
    const int u_lo = unit_exp + ea_lo + eb - in.zint, u_hi = u_lo + (h ? eA15 : eA7);
    finish_rows<PPL, FULL>(p, b, t0, n, h, lds, gl, pb, lane, smin, smax, u_lo, u_hi, in.zfrac);
    F2_STAMP(6)
  }
}

#ifndef E2E_F2_LDSPAD
#define E2E_F2_LDSPAD 0
#endif
#ifndef E2E_F2_WPB                  // independent segment waves per workgroup (tools/diag)
#define E2E_F2_WPB 1
#endif
template <int PPL, typename P>
__device__ __forceinline__ void segment_wave(const P& p, unsigned char* smem) {
  const int b = blockIdx.y, seg = blockIdx.x * E2E_F2_WPB + (E2E_F2_WPB > 1 ? (int)(threadIdx.x >> 6) : 0), lane = threadIdx.x & 63;
  if (E2E_F2_WPB > 1 && seg >= p.NS) return;
  const int V = p.V, Tmax = p.T, t0 = seg * kSeg;
  if (E2E_F2_WPB > 1) smem += (threadIdx.x >> 6) * ((F2Lds<PPL>::bytes(V) + E2E_F2_LDSPAD + 15) & ~(size_t)15);
  const F2Lds<PPL> lds(smem, V);
  typedef float f4 __attribute__((ext_vector_type(4)));
  F2_STAMP(-1)

  // Everything the wave needs from global memory is requested up front, BEFORE the utterance's lengths are known
  // (no address depends on them; what a dead or flagged segment reads stays inside the workspace and is dropped):
  // one memory round trip instead of three in a kernel whose waves live for only ~10 us.
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  // (a) the utterance's lattice description, left in the workspace by F1 (cellinfo_wave)
  unsigned w[PPL];
  {
    const unsigned* ci = p.cinfo + (size_t)b * (p.CELLS / 2) + PPL * lane;
#pragma unroll
    for (int r = 0; r < PPL; r++) w[r] = ci[r];
  }
  constexpr bool BIG = P::kBigV;
  const int* ls = p.lstart + (size_t)b * p.LS;
  int s0 = 0, s1 = 0, s2 = 0;
  if (!BIG) { s0 = ls[lane]; s1 = ls[64 + lane]; s2 = ls[128 + (lane & 1)]; }      // (BIG: up to 514 entries, copied below)
  // (b) the segment's probabilities.  Small alphabets: F1 left them as the tile is laid out here -- [label][16 steps] -- so a
  //     lane's 16-byte load IS four steps of a label (<= 6 loads per lane for V <= 96).  BIG: the table is row-major [frame][V];
  //     a lane takes COLUMNS -- 64 c + lane, the 16 steps of each in 16 four-byte loads (a wave's load is 256 consecutive bytes
  //     of a row) -- so that what it holds afterwards is four 16-byte runs of the transposed tile.  Four chunks of 64 columns
  //     now, the rest after these have been staged.  (Until round 5 a lane loaded 16 consecutive bytes of a row and scattered
  //     them into the tile float by float: the writes of a wave went to two of the 32 banks -- the tile's rows are 80 bytes apart
  //     -- and with five such waves on a CU staging alone was 14.9k of a segment's 34k cycles at 200 columns.)
  constexpr int kTileLoads = BIG ? 1 : (4 * kMaxSmallV + 63) / 64;
  constexpr int kColChunks = BIG ? 4 : 1;
  f4 tile[kTileLoads];
  float col[kColChunks][kSeg];
  const int rlast = min(kSeg, Tmax - t0) - 1;            // (rows past the table's end are not touched: they are dead steps, zeroed below)
  const float* csrc = p.ytab + ((size_t)b * Tmax + t0) * V;
  auto load_cols = [&](int c0) {
#pragma unroll
    for (int c = 0; c < kColChunks; c++) {
      if (64 * (c0 + c) < V) {                                              // (uniform)
        const int vo = min(64 * (c0 + c) + lane, V - 1);
        const float* row = csrc;                                            // (a running scalar row address and one lane offset)
#pragma unroll
        for (int tt = 0; tt < kSeg; tt++) { col[c][tt] = row[vo]; row += tt < rlast ? V : 0; }
      }
    }
  };
  if (BIG) load_cols(0);
  else {
    const f4* src = reinterpret_cast<const f4*>(p.ytab + ((size_t)b * p.NS + seg) * V * kSeg);
#pragma unroll
    for (int j = 0; j < kTileLoads; j++)
      if (64 * j < 4 * V) tile[j] = src[min(64 * j + lane, 4 * V - 1)];     // (uniform test; the last load's surplus lanes re-read the end)
  }
  // (c) alpha checkpoint and the rescale exponents
  SegIn<PPL> in;
  in.load(p, b, seg, (int)Sq, lane);       // (Sq was requested first and has had the other requests' time to arrive)

  if (Tq < 1 || Tq > Tmax || Sq < 0 || Sq > p.Smax) return;    // flagged by F1, the exact kernel poisons it
  const int T = (int)Tq, S = (int)Sq;
  {
    // frames past the utterance's end: exp(lp) in log-prob mode (quirk Q1), zero for fused logits
    const size_t g0 = (size_t)b * Tmax * V;
    const int64_t xo = (int64_t)b * p.sB;
    const int tend = min(t0 + kSeg, Tmax);
    for (int t = max(t0, T); t < tend; t++)
      for (int v = lane; v < V; v += 64)
        store_elem(p.grads, g0 + (size_t)t * V + v,
                   p.logprobs ? expf(load_elem(p.x, xo + (int64_t)t * p.sT + (int64_t)v * p.sV, p.xdt)) * p.gscale : 0.f, p.xdt);
  }
  if (t0 >= T) return;
  const int n = min(t0 + kSeg, T) - t0;

  LaneCells<PPL> lc;
  int rank[PPL];
  lc.unpack(w, S, T, rank);
#pragma unroll
  for (int r = 0; r < PPL; r++) rank[r] = ps_slot_address(lds.Ps, rank[r]);      // kept as LDS byte addresses: see ps_put
  if (BIG) { for (int i = lane; i < lstart_ints(V); i += 64) lds.starts[i] = ls[i]; }
  else { lds.starts[lane] = s0; lds.starts[64 + lane] = s1; if (lane < 2) lds.starts[128 + lane] = s2; }
  if (lane < kYs) lds.ys[V * kYs + lane] = 0.f;                           // the zero row V
  if (!BIG) {
    // one 16-byte LDS write per load: label (lane >> 2) + 16 j, steps 4 (lane & 3) .. + 3
    float* dst = lds.ys + (lane >> 2) * kYs + 4 * (lane & 3);
    const int q4 = 4 * (lane & 3);
#pragma unroll
    for (int j = 0; j < kTileLoads; j++) {
      if (64 * j < 4 * V) {
        f4 v = tile[j];
        if (n < kSeg) {                                                    // (uniform) dead steps of a short last segment: zeros
#pragma unroll
          for (int c = 0; c < 4; c++) v[c] = q4 + c < n ? v[c] : 0.f;
        }
        if (64 * j + lane < 4 * V) *reinterpret_cast<f4*>(dst + 16 * j * kYs) = v;
      }
    }
  } else if (!(E2E_F2_ABL & 32)) {
    // the lane's columns into the transposed tile: four 16-byte writes per column
    auto put_cols = [&](int c0) {
#pragma unroll
      for (int c = 0; c < kColChunks; c++) {
        const int v = 64 * (c0 + c) + lane;
        if (64 * (c0 + c) < V && v < V) {
          f4* dst = reinterpret_cast<f4*>(lds.ys + v * kYs);
#pragma unroll
          for (int k = 0; k < kSeg / 4; k++) dst[k] = f4{col[c][4 * k], col[c][4 * k + 1], col[c][4 * k + 2], col[c][4 * k + 3]};
        }
      }
    };
    put_cols(0);
    for (int c0 = kColChunks; 64 * c0 < V; c0 += kColChunks) { load_cols(c0); put_cols(c0); }      // (beyond 256 columns: a second round)
    if (n < kSeg)                                                          // dead steps of a short last segment: zeros
      for (int v = lane; v < V; v += 64)
        for (int tt = n; tt < kSeg; tt++) lds.ys[v * kYs + tt] = 0.f;
  }
  if (lane < kHalf) lds.Ps[lane * F2Lds<PPL>::PROW - 1] = 0.f;           // the rows' zero guards
  F2_LDS_ORDER   // staged rows visible to this (single) wave
  GradLanesT<BIG ? (PPL == 8 ? (kMaxHugeV + 63) / 64 : (kMaxBigV + 63) / 64) : 2> gl;
  gl.init(lds, V, p.blank, lane);
  F2_STAMP(0)
  float smin = __builtin_huge_valf(), smax = 0.f;
  const bool full = __builtin_amdgcn_readfirstlane((seg > 0 && n == kSeg && t0 + n < T) ? 1 : 0) != 0;
#ifndef E2E_F2_SCALAR                // (the scalar body for every width: tools/diag A/B)
  if constexpr (PPL == 4) {
    if (full) segment_body_pk<true>(p, b, seg, T, S, n, lc, rank, in, lds, gl, lane, smin, smax);
    else segment_body_pk<false>(p, b, seg, T, S, n, lc, rank, in, lds, gl, lane, smin, smax);
  } else
#endif
  if (full) segment_body<PPL, true>(p, b, seg, T, S, n, lc, rank, in, lds, gl, lane, smin, smax);
  else segment_body<PPL, false>(p, b, seg, T, S, n, lc, rank, in, lds, gl, lane, smin, smax);
#ifdef E2E_FAST_PROFILE
  if (lane == 0) s_prof_acc[7] = ((unsigned long long)__float_as_uint(smin) << 32) | __float_as_uint(smax);
#endif
  F2_FLUSH
  // range check: everything that carries posterior mass was representable (see the header comment)
  const bool finite_ok = smax < __builtin_huge_valf();
  // (smin: a row sum that can still be inverted and whose terms of relative weight 2^-6 are normal f32.  What decides
  // whether the rows kept everything that mattered is the self-check in finish_rows, which also zeroes smin: with a
  // margin alone -- 2^-90 -- two utterances in three with emissions that contradict their targets were sent to the f64
  // redo for nothing, and with 2^-110 and no self-check a randomised sweep let gradients through that were off by 2e-3.
  // Rows between F1's rescales legitimately sit 2^-40 .. 2^-80 below the unit: alpha is rescaled at t%8 == 7 and beta
  // at t%8 == 0, so every row in between carries nine steps of decay.)
  // (the segment's bit in the mask: what the f64 redo of a range-flagged utterance takes -- the segments that failed and the
  //  borderline ones; the rows of every other segment passed their self-check with room to spare and stay as written)
  if (!(smin > 0.f) || !finite_ok) {
    if (lane == 0) { atomicOr(&p.flags[b], finite_ok ? 8 : 16); atomicOr(&p.segmask[(size_t)b * p.MW + (seg >> 5)], 1u << (seg & 31)); }
  } else if (smin < __builtin_huge_valf()) {
    if (lane == 0) atomicOr(&p.segmask[(size_t)b * p.MW + (seg >> 5)], 1u << (seg & 31));
  }
#ifdef E2E_F2_MASK_ALL              // (tools/diag: every segment of a range-flagged utterance is redone)
  if (lane == 0) atomicOr(&p.segmask[(size_t)b * p.MW + (seg >> 5)], 1u << (seg & 31));
#endif
  if (seg == 0 && lane == 0) {
    const double za = p.logz[2 * b], zb = p.logz[2 * b + 1];
    if (!(fabs(za - zb) <= 1e-6 * fabs(za) + 1e-4)) atomicOr(&p.flags[b], 32);
  }
}

// (Folding the flagged-utterance scan and the loss reduction into this kernel's tail -- every wave takes a ticket, the
// last one scans the flags -- was built and measured: the ticket needs the wave's flag atomics ordered before it, i.e.
// the wave has to wait for its outstanding gradient stores instead of retiring under them, 65 -> 97 us for the kernel;
// with a release fence at agent scope, which is an L2 write-back on this multi-XCD part, 411 us.  An empty launch costs
// ~4.5 us here whatever it does, so the scan stays in the fallback launch, which also writes the optional reduction.)
#ifndef E2E_F2_MINW                 // (both overridable for tools/diag occupancy experiments)
#define E2E_F2_MINW 2
#endif
#define E2E_F2_MINW4 (E2E_F2_HALF ? 3 : E2E_F2_MINW)     // four pairs per lane: three waves per SIMD with half the alpha rows kept
#ifndef E2E_F2_MINW2                // two pairs per lane: 134 registers by themselves = three waves per SIMD; held to 128 (five spilled, 24 bytes
#define E2E_F2_MINW2 4              // of scratch) four fit -- round 6: B=256, T=1000, S<=100 111.3 -> 102.8 us per call, the compact lattice of
#endif                              // configs[4] (B=512, T=256, 65 columns) 78.5 -> 75.6
// (eight pairs per lane: 16 alpha rows of 16 cells are 256 registers by themselves -- one wave per SIMD, no spills)
template <int PPL, bool O16, bool BIG = false>
__global__ E2E_KERNEL_ALIGN __launch_bounds__(64 * E2E_F2_WPB, PPL == 8 ? 1 : PPL == 4 ? E2E_F2_MINW4 : PPL == 2 ? E2E_F2_MINW2 : E2E_F2_MINW) void ctc_fast_segment_kernel(FastParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  SegParams<O16, BIG> q;
  static_cast<FastParams&>(q) = p;
  segment_wave<PPL>(q, smem);
}

// the segment kernel behind a chain kernel: the instance of the gradient's element width
template <int PPL, bool BIG = false>
int launch_segments(const FastParams& p, size_t lds, hipStream_t stream) {
  const dim3 grid((p.NS + E2E_F2_WPB - 1) / E2E_F2_WPB, p.B), block(64 * E2E_F2_WPB);
  if (E2E_F2_WPB > 1) lds = ((lds + 15) & ~(size_t)15) * E2E_F2_WPB;
  if constexpr (BIG) {
    if (dtype_is_16bit(p.xdt)) hipLaunchKernelGGL((ctc_fast_segment_kernel<PPL, true, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((ctc_fast_segment_kernel<PPL, false, true>), grid, block, lds, stream, p);
  } else
  if (dtype_is_16bit(p.xdt)) hipLaunchKernelGGL((ctc_fast_segment_kernel<PPL, true>), grid, block, lds, stream, p);
  else hipLaunchKernelGGL((ctc_fast_segment_kernel<PPL, false>), grid, block, lds, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_segment_kernel launch");
  return E2E_OK;
}

// (Two waves per segment -- a 128-thread workgroup, two label pairs per lane, the waves' lattice halves overlapping by 18
// pairs so that nothing crosses between them inside the 16 steps, rows of a half split between the waves for the scan and
// the gradient, 154 registers and three waves per SIMD -- was built, parity-green on the whole loss suite, and measured:
// 145 against 133 us per step.  The waves split the lattice but not the bookkeeping: a wave of the pair executes ~80 % of
// the one-wave kernel's instructions (its scalar work does not shrink at all), and this kernel's time is its instruction
// count -- a SIMD issues one per ~4.3 cycles with two such waves and one per ~3.5 with three.)
// (Persistent segment waves -- G waves per utterance, each walking every G-th segment with the next segment's inputs
// requested one segment ahead by LDS-DMA (global_load_lds, no destination registers) and a counted vmcnt wait that lets the
// gradient stores drain under the next segment -- were built, parity-green, and measured: 137 against 131 us per step.  The
// one-segment waves already overlap each other's load latency and store tails; a register prefetch spills (tried twice).)
// targets of 256..447 labels: the halo chains on four waves per direction, the segment kernel with eight pairs per lane
// ... over the f32 ring and the probability table of the wide-row form (ChainF64LW) where the alphabet asks for it -- or where the
// f64 ring of ChainF64L on its sixteen waves does not fit the LDS (73..96 columns: 170..192 KB; until round 5 such a call failed
// with "hipFuncSetAttribute: invalid argument")
bool long_wide_rows(int V) { return V > kMaxSmallV || HfLds::of<ChainF64L>(V).total > 160 * 1024; }
int launch_fast_long(const FastParams& p, hipStream_t stream) {
  if (long_wide_rows(p.V)) {       // the wide-row form (see ChainF64LW)
    hipLaunchKernelGGL(ctc_fast_prob_kernel<(kMaxHugeV + 63) / 64>, dim3((unsigned)(((int64_t)p.B * p.T + 4 * kProbRows - 1) / (4 * kProbRows))), dim3(256), 0, stream, p);
    E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_prob_kernel launch");
    const HfLds hl = HfLds::of<ChainF64LW>(p.V);
    E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_hf_kernel<8, ChainF64LW>), hl.total), "hipFuncSetAttribute");
    hipLaunchKernelGGL((ctc_fast_chain_hf_kernel<8, ChainF64LW>), dim3(p.B), dim3(ChainF64LW::kWaves * 64), hl.total, stream, p);
    E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_hf_kernel launch");
    return launch_segments<8, true>(p, F2Lds<8>::bytes(p.V), stream);
  }
  const HfLds hl = HfLds::of<ChainF64L>(p.V);
  E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_hf_kernel<8, ChainF64L>), hl.total), "hipFuncSetAttribute");
  hipLaunchKernelGGL((ctc_fast_chain_hf_kernel<8, ChainF64L>), dim3(p.B), dim3(ChainF64L::kWaves * 64), hl.total, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_hf_kernel launch");
  return launch_segments<8>(p, F2Lds<8>::bytes(p.V), stream);
}

constexpr bool kLeanDefault = true;    // (targets of 128..223 labels: 132.4 against 134.2 us per call at the headline shape)

template <int PPL>
int launch_fast_ppl(const FastParams& p, hipStream_t stream) {
  if (p.V > kMaxSmallV) {
    // 97..224 columns: the halo chains over the f32 ring, the segment kernel's wide-row form (fast_supported: PPL == 4 here)
    if constexpr (PPL == 4) {
      hipLaunchKernelGGL(ctc_fast_prob_kernel<(kMaxBigV + 63) / 64>, dim3((unsigned)(((int64_t)p.B * p.T + 4 * kProbRows - 1) / (4 * kProbRows))), dim3(256), 0, stream, p);
      E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_prob_kernel launch");
      const HfLds hl = HfLds::of<ChainF64W>(p.V);
      E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_hf_kernel<4, ChainF64W>), hl.total), "hipFuncSetAttribute");
      hipLaunchKernelGGL((ctc_fast_chain_hf_kernel<4, ChainF64W>), dim3(p.B), dim3(ChainF64W::kWaves * 64), hl.total, stream, p);
      E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_hf_kernel launch");
      return launch_segments<4, true>(p, F2Lds<4>::bytes(p.V) + E2E_F2_LDSPAD, stream);
    } else { set_error("fast CTC path: %d columns need four pairs per lane", p.V); return E2E_ERR_UNSUPPORTED; }
  }
  const size_t lds1 = F1Lds::bytes(p.V);
  E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_kernel<PPL>), (int)lds1), "hipFuncSetAttribute");
  const size_t lds2 = F2Lds<PPL>::bytes(p.V) + E2E_F2_LDSPAD;
  // f32 chains: where the caller allows them (e2e_ctc_loss_opts.chains) and they are faster, i.e. at the widest rows
  // (145 against 165 us per step at S <= 200, but 115 / 93 against 120 / 94 us at S <= 127 / 63, with looser gradients: not
  // worth it there); E2E_F1_F32=1 forces them everywhere (tests)
  static const bool force_f32_chains = getenv("E2E_F1_F32") != nullptr;
  if (force_f32_chains || (p.chains == E2E_CHAINS_F32 && PPL == 4)) {
    const HfLds hl = HfLds::of<ChainF32>(p.V);
    E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_hf_kernel<PPL, ChainF32>), hl.total), "hipFuncSetAttribute");
    hipLaunchKernelGGL((ctc_fast_chain_hf_kernel<PPL, ChainF32>), dim3(p.B), dim3(ChainF32::kWaves * 64), hl.total, stream, p);
    E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_hf_kernel launch");
    FastParams q = p; q.ztol = kZTolF32;                          // (trkA / trkB: written by the frame waves)
    return launch_segments<PPL>(q, lds2, stream);
  }
  static const bool force_single = getenv("E2E_F1_SINGLE") != nullptr, force_halo = getenv("E2E_F1_HALO") != nullptr;
  // the lean halo chains of ctc_loss_fast_h1.hip.  E2E_F1_LEAN=0 / 1: never / wherever they fit (A/B, tests)
  static const char* lean_env = getenv("E2E_F1_LEAN");
  const bool no_h1 = lean_env && lean_env[0] == '0', force_h1 = lean_env && lean_env[0] == '1';
  // (targets of 64..127 labels, round 6: the lean chains win while every utterance has a CU to itself -- B=256, T=1000, S<=100: 108.3
  //  against 116.9 us per call, S<=127: 115.4 against 117.6 -- and lose beyond, where two workgroups of the single-wave kernel share
  //  a CU and one of the lean kernel fills it: B=512, T=256, S<=64 73.8 against 59.3 us; tools/diag/lean_ab.sh)
  const bool lean_pays = PPL == 4 || (PPL == 2 && p.B <= 256);
  if (!no_h1 && !force_single && (kLeanDefault ? lean_pays || force_h1 : force_h1) && h1_supported(p.V, p.Smax, PPL)) {
    const int rc = launch_fast_h1_chain(p, PPL, stream);
    if (rc != E2E_OK) return rc;
    return launch_segments<PPL>(p, lds2, stream);
  }
  // f64 halo chains: two waves per direction hold 224 label pairs.  Where they win: the widest rows (158 against 166 us per
  // step at S <= 200; at S <= 127 the single wave carries two pairs per lane itself and wins, 120 against 131 us).
  // E2E_F1_SINGLE=1: the single-wave chains everywhere, E2E_F1_HALO=1: the halo chains wherever they fit (A/B, tests)
  if (!force_single && (PPL == 4 || (force_halo && PPL >= 1)) && p.Smax + 1 <= ChainF64::kMaxW * kHfOwn) {
    const HfLds hl = HfLds::of<ChainF64>(p.V);
    E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_hf_kernel<PPL, ChainF64>), hl.total), "hipFuncSetAttribute");
    hipLaunchKernelGGL((ctc_fast_chain_hf_kernel<PPL, ChainF64>), dim3(p.B), dim3(ChainF64::kWaves * 64), hl.total, stream, p);
    E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_hf_kernel launch");
    return launch_segments<PPL>(p, lds2, stream);
  }
  // (the ring at half its depth where the full one keeps a second workgroup off a CU that the batch has work for)
  static const char* rb_env = getenv("E2E_F1_RING");               // (A/B: 4 / 8 forces the depth)
  const bool shallow = rb_env ? rb_env[0] == '4' : (p.B > 256 && lds1 > 80 * 1024 && F1Lds::bytes(p.V, 4) <= 80 * 1024);
  if (shallow) {
    const size_t lds4 = F1Lds::bytes(p.V, 4);
    E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_kernel<PPL, 4>), (int)lds4), "hipFuncSetAttribute");
    hipLaunchKernelGGL((ctc_fast_chain_kernel<PPL, 4>), dim3(p.B), dim3(512), lds4, stream, p);
  } else hipLaunchKernelGGL(ctc_fast_chain_kernel<PPL>, dim3(p.B), dim3(512), lds1, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_kernel launch");
  FastParams q = p; q.trkA = p.cumA; q.trkB = p.cumB;          // (its frame follows the maximum itself)
  return launch_segments<PPL>(q, lds2, stream);
}

int ppl_for(int Smax) {
  // pairs per lane: 64*PPL pairs plus the final blank must fit 128*PPL cells
  if (Smax <= 63) return 1;
  if (Smax <= 127) return 2;
  if (Smax <= 255) return 4;
  if (Smax + 1 <= ChainF64L::kMaxW * kHfOwn) return 8;      // 447: what four halo chain waves per direction hold
  return 0;
}

// ... of a call: alphabets beyond kMaxSmallV run on ChainF64W and the segment kernel's four-pairs form only
int ppl_of(int V, int Smax) {
  if (V > kMaxSmallV) {
    if (V <= kMaxBigV && Smax + 1 <= ChainF64W::kMaxW * kHfOwn) return 4;
    return (V <= kMaxHugeV && Smax + 1 <= ChainF64LW::kMaxW * kHfOwn) ? 8 : 0;      // (more columns or more labels: eight pairs per lane)
  }
  return ppl_for(Smax);
}

struct FastLayout {
  size_t ytab, ckA, ckQ, ckE, cumA, cumB, trkA, trkB, logz, zt2, flags, segmask, cinfo, lstart, ctl, ckXA, ckXQ, extz, total;
  int NS, NB, CELLS, MW, LS;
};

FastLayout fast_layout(int B, int T, int V, int Smax) {
  FastLayout l;
  const int ppl = ppl_of(V, Smax);
  l.CELLS = 128 * (ppl > 0 ? ppl : 1);
  l.LS = max(130, lstart_ints(V));
  l.NS = (T + kSeg - 1) / kSeg;
  l.NB = (T + kBlk - 1) / kBlk + 4;       // (cumA / cumB are read up to two blocks past an utterance's last)
  size_t o = 0;
  l.ytab = o; o += align_up((size_t)B * l.NS * kSeg * V * sizeof(float), 256);
  l.ckA = o; o += align_up((size_t)B * l.NS * l.CELLS * sizeof(float), 256);
  l.ckQ = o; o += align_up((size_t)B * l.NS * l.CELLS * sizeof(float), 256);
  l.ckE = o; o += align_up((size_t)B * l.NS * 2 * 64 * sizeof(short), 256);
  l.cumA = o; o += align_up((size_t)B * l.NB * sizeof(int), 256);
  l.cumB = o; o += align_up((size_t)B * l.NB * sizeof(int), 256);
  l.trkA = o; o += align_up((size_t)B * l.NB * sizeof(int), 256);
  l.trkB = o; o += align_up((size_t)B * l.NB * sizeof(int), 256);
  l.zt2 = o; o += align_up((size_t)B * sizeof(double), 256);
  l.logz = o; o += align_up((size_t)B * 2 * sizeof(double), 256);
  l.flags = o; o += align_up((size_t)B * sizeof(int), 256);
  l.MW = (l.NS + 31) / 32;
  l.segmask = o; o += align_up((size_t)B * l.MW * sizeof(unsigned), 256);
  l.cinfo = o; o += align_up((size_t)B * (l.CELLS / 2) * sizeof(unsigned), 256);
  l.lstart = o; o += align_up((size_t)B * l.LS * sizeof(int), 256);
  l.ctl = o; o += 256;
  // the extended-range redo of the flagged-utterance launch (ctc_ext.h): one exponent per checkpoint cell, both directions
  l.ckXA = o; o += align_up((size_t)B * l.NS * l.CELLS * sizeof(int), 256);
  l.ckXQ = o; o += align_up((size_t)B * l.NS * l.CELLS * sizeof(int), 256);
  l.extz = o; o += align_up((size_t)B * 10 * sizeof(double), 256);     // [B][2] Z, then [B][2 sides][4] the sides' cells of Z (ext_chains)
  l.total = o;
  return l;
}

}  // namespace

bool fast_supported(int T, int V, int Smax, int dtype) {
  // (T: the halo chains tag their frame words with the block index in 19 bits)
  // (16-bit logits: read and written in their dtype around the same f32 lattice)
  return (dtype == E2E_F32 || dtype_is_16bit(dtype)) && V >= 2 && ppl_of(V, Smax) != 0 && T < (1 << 22);
}

size_t fast_workspace_bytes(int B, int T, int V, int Smax) {
  return fast_layout(B, T, V, Smax).total;
}

int launch_exact_flagged(const LossArgs& a, int* flags, int mode, const FastRetry* retry);

int launch_fast(const LossArgs& a, bool fallback_to_exact) {
  const FastLayout l = fast_layout(a.B, a.T, a.V, a.Smax);
  size_t need = l.total;
  if (fallback_to_exact) need += exact_fallback_workspace_bytes(a.B, a.T, a.V, a.Smax);
  if (!a.ws || a.ws_bytes < need) { set_error("workspace too small: %zu < %zu", a.ws_bytes, need); return E2E_ERR_WORKSPACE; }
  if (a.B == 0) return E2E_OK;
  char* ws = reinterpret_cast<char*>(a.ws);
  FastParams p;
  p.x = a.x; p.xdt = a.dtype; p.sB = a.sB; p.sT = a.sT; p.sV = a.sV; p.logprobs = a.logprobs;
  p.targets = a.targets; p.tgt_stride = a.tgt_stride; p.x_len = a.x_len; p.t_len = a.t_len;
  p.B = a.B; p.T = a.T; p.V = a.V; p.Smax = a.Smax; p.blank = a.blank;
  p.losses = reinterpret_cast<float*>(a.losses); p.grads = a.grads;
  p.ytab = reinterpret_cast<float*>(ws + l.ytab);
  p.ckA = reinterpret_cast<float*>(ws + l.ckA); p.ckQ = reinterpret_cast<float*>(ws + l.ckQ);
  p.ckE = reinterpret_cast<short*>(ws + l.ckE);
  p.cumA = reinterpret_cast<int*>(ws + l.cumA); p.cumB = reinterpret_cast<int*>(ws + l.cumB);
  p.trkA = reinterpret_cast<int*>(ws + l.trkA); p.trkB = reinterpret_cast<int*>(ws + l.trkB);
  p.zt2 = reinterpret_cast<double*>(ws + l.zt2);
  p.logz = reinterpret_cast<double*>(ws + l.logz); p.flags = reinterpret_cast<int*>(ws + l.flags);
  p.segmask = reinterpret_cast<unsigned*>(ws + l.segmask); p.MW = l.MW;
  p.cinfo = reinterpret_cast<unsigned*>(ws + l.cinfo); p.lstart = reinterpret_cast<int*>(ws + l.lstart); p.LS = l.LS;
  p.ctl = reinterpret_cast<int*>(ws + l.ctl);
  p.gscale = (float)a.grad_scale;
  p.ztol = kZTol;
  p.chains = a.chains;
  p.NS = l.NS; p.NB = l.NB; p.CELLS = l.CELLS;
  int rc;
  switch (ppl_of(a.V, a.Smax)) {
    case 1: rc = launch_fast_ppl<1>(p, a.stream); break;
    case 2: rc = launch_fast_ppl<2>(p, a.stream); break;
    case 4: rc = launch_fast_ppl<4>(p, a.stream); break;
    case 8: rc = launch_fast_long(p, a.stream); break;
    default: set_error("fast CTC path: Smax=%d too long", a.Smax); return E2E_ERR_UNSUPPORTED;
  }
  if (rc != E2E_OK) return rc;
  LossArgs e = a;
  e.ws = ws + l.total; e.ws_bytes = a.ws_bytes - l.total;
  // mode 1: redo flagged utterances exactly; mode 2: no fallback requested -> poison them
  FastRetry rt;
  rt.ytab = p.ytab; rt.ytab_segments = (a.V <= kMaxSmallV && !(ppl_of(a.V, a.Smax) == 8 && long_wide_rows(a.V))) ? 1 : 0; rt.ckA = p.ckA; rt.ckQ = p.ckQ; rt.ckE = p.ckE; rt.cumA = p.cumA; rt.cumB = p.cumB;
  rt.NS = p.NS; rt.NB = p.NB; rt.CELLS = p.CELLS; rt.PPL = ppl_of(a.V, a.Smax); rt.logz = p.logz; rt.ctl = p.ctl;
  rt.segmask = p.segmask; rt.MW = p.MW;
  rt.ckXA = reinterpret_cast<int*>(ws + l.ckXA); rt.ckXQ = reinterpret_cast<int*>(ws + l.ckXQ); rt.extz = reinterpret_cast<double*>(ws + l.extz);
  return launch_exact_flagged(e, p.flags, fallback_to_exact ? 1 : 2, &rt);
}

}  // namespace e2e

#ifdef E2E_FAST_PROFILE
extern "C" int e2e_debug_fast_zdev(float* host, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  if (reset) { void* ptr; if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(e2e::fastk::g_zdev)) != hipSuccess) return E2E_ERR_HIP;
    return hipMemset(ptr, 0, sizeof(float) * 16384) == hipSuccess ? 0 : E2E_ERR_HIP; }
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(e2e::fastk::g_zdev), sizeof(float) * 16384) == hipSuccess ? 0 : E2E_ERR_HIP;
}
extern "C" int e2e_debug_fast_profile2(unsigned long long* host, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  if (reset) { void* ptr; if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(e2e::fastk::g_prof2)) != hipSuccess) return E2E_ERR_HIP;
    return hipMemset(ptr, 0, sizeof(unsigned long long) * 16384 * 8) == hipSuccess ? 0 : E2E_ERR_HIP; }
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(e2e::fastk::g_prof2), sizeof(unsigned long long) * 16384 * 8) == hipSuccess ? 0 : E2E_ERR_HIP;
}
extern "C" int e2e_debug_fast_profile3(unsigned long long* host, int reset) {
  if (reset) { void* ptr; if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(e2e::fastk::g_prof3)) != hipSuccess) return E2E_ERR_HIP;
    return hipMemset(ptr, 0, sizeof(unsigned long long) * 256 * 16 * 4) == hipSuccess ? 0 : E2E_ERR_HIP; }
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(e2e::fastk::g_prof3), sizeof(unsigned long long) * 256 * 16 * 4) == hipSuccess ? 0 : E2E_ERR_HIP;
}
extern "C" int e2e_debug_fast_profile(unsigned long long* host, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(e2e::fastk::g_prof), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : E2E_ERR_HIP;
}
#endif

// Diagnostics (not part of include/e2e_ctc.h): copy the fast path's per-utterance flag words and both log Z
// values out of a workspace that the last e2e_ctc_loss_fwd_bwd(ALGO_FAST/AUTO) call used.  Synchronises.
// Diagnostics: how many flagged utterances of the last AUTO call the f64 redo of the segments could NOT settle (they were
// recomputed by the exact kernel: ~7 ms for a batch instead of ~1).  Synchronises.
extern "C" int e2e_debug_fast_redo_failures(const void* workspace, int B, int T, int V, int Smax, int* count_host) {
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const char* ws = reinterpret_cast<const char*>((base + 255) & ~(uintptr_t)255);
  const e2e::FastLayout l = e2e::fast_layout(B, T, V, Smax);
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  int ctl[4];
  if (hipMemcpy(ctl, ws + l.ctl, sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  *count_host = ctl[1];
  return E2E_OK;
}
// Diagnostics: the flagged-utterance launch's phases as workgroup 0 saw them, in microseconds since its start: end of its f64
// redos of single segments, of the wait for the other workgroups, of its extended-range chains, of its extended-range segments,
// and the end of the launch's last workgroup.  Zeros if nothing was flagged.  Synchronises.
extern "C" int e2e_debug_flagged_phases(const void* workspace, int B, int T, int V, int Smax, double* us_host) {
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const char* ws = reinterpret_cast<const char*>((base + 255) & ~(uintptr_t)255);
  const e2e::FastLayout l = e2e::fast_layout(B, T, V, Smax);
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  unsigned long long st[8];
  if (hipMemcpy(st, ws + l.ctl + 64, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  for (int k = 1; k <= 4; k++) us_host[k - 1] = st[0] && st[k] ? (double)(long long)(st[k] - st[0]) * 0.01 : 0.0;
  us_host[4] = st[0] && st[6] ? (double)(long long)(st[6] - st[0]) * 0.01 : 0.0;
  us_host[5] = st[0] && st[5] ? (double)(long long)(st[5] - st[0]) * 0.01 : 0.0;
  return E2E_OK;
}
// Diagnostics: bounded waits of the flagged-utterance launch that ran out (ctl[4]: a workgroup gave up waiting for the others, or a
// segment for its utterance's chains -- what they left undone was recomputed by the exact kernel) and redos of single segments
// that failed (ctl[5]).  Both 0 on an idle GPU.  Synchronises.
extern "C" int e2e_debug_flagged_counters(const void* workspace, int B, int T, int V, int Smax, int* timeouts_host, int* failed_redos_host) {
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const char* ws = reinterpret_cast<const char*>((base + 255) & ~(uintptr_t)255);
  const e2e::FastLayout l = e2e::fast_layout(B, T, V, Smax);
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  int ctl[8];
  if (hipMemcpy(ctl, ws + l.ctl, sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  *timeouts_host = ctl[4]; *failed_redos_host = ctl[5];
  return E2E_OK;
}
extern "C" int e2e_debug_fast_state(const void* workspace, int B, int T, int V, int Smax, int* flags_host, double* logz_host) {
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const char* ws = reinterpret_cast<const char*>((base + 255) & ~(uintptr_t)255);
  const e2e::FastLayout l = e2e::fast_layout(B, T, V, Smax);
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  if (hipMemcpy(flags_host, ws + l.flags, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  if (hipMemcpy(logz_host, ws + l.logz, sizeof(double) * 2 * (size_t)B, hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  return E2E_OK;
}

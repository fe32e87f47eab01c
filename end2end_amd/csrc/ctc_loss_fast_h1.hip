// The lean halo chain kernel of the fast CTC path (see ctc_loss_fast.hip for the path's description and fast_common.h
// for what it shares with the other chain kernels).
#include "fast_common.h"

namespace e2e {
namespace fastk {
namespace {

// ============================================================================================
// F1, lean halo form ("hx"): NP label pairs per lane, f64 cells
// ============================================================================================
// A chain wave's time is the number of instructions it issues (an in-order wave issues one every ~5.4 cycles, a dependent
// one every ~8.5, whatever its kind: tools/diag/microbench/issue_latency.hip), and the two-pairs-per-lane halo kernel of
// ctc_loss_fast.hip spends 27 per step of which 13 are the lattice.  This kernel is that structure -- several waves per chain,
// halo lanes refilled from the neighbour's edge lanes, one common frame decided two blocks late by a frame wave, checkpoint
// rows handed to a checkpoint wave through LDS, producers filling the f64 probability ring -- with the bookkeeping cut down:
//   * the interior blocks run eight at a time, one per ring slot, so that every LDS address of a block is a register that
//     never changes plus a constant, and block parity / checkpoint phase are compile-time facts;
//   * the beta chain holds its cells SHIFTED by one -- lane slot j keeps (blank j, label j-1) -- because
//         label'(j-1) = (label(j-1) + wb blank(j) + skip(j) label(j)) y[l(j-1)],      blank'(j) = blank(j) yb + label(j)
//     takes ONE value from the slot above, label(j): the step is the mirror image of alpha's, 4 f64 operations per pair and
//     one value (2 DPP moves) per lane in either direction (the (blank j, label j) pairing costs beta a fifth operation);
//   * a wave without an upstream neighbour reads a wave slot that holds no cells (progress "idle", zero records), and lanes
//     with nothing to publish store to a dump row: no case distinctions, no exec masks around the stores;
//   * the neighbour's progress word and edge record, and the checkpoint wave's progress, are asked for two steps / four
//     steps before they are needed and only looked at then;
//   * a half block's probabilities are requested so that what is used first is asked for LAST: one wait covers the half.
// NP = 2: two waves per direction, 12 waves (the default for targets of 128..223 labels).  NP = 1: four waves per direction,
// 16 waves -- built because halving a wave's lattice work looked like the lever; it is not (E2E_F1_NP=1 selects it: A/B).
#ifndef E2E_HX_ABL                  // tools/diag: timing builds with parts of the chain waves' work switched off (results meaningless)
#define E2E_HX_ABL 0               //  1: probabilities read once, 2: no waiting for the producers, 4: no halo exchange, 8: no frame, 16: no checkpoints
#endif
#ifndef E2E_HX_SLIM                 // 1: eight waves (NP = 2) -- one producer per direction, frame and checkpoint roles of a direction on ONE
#define E2E_HX_SLIM 0              //    wave, the lattice description written by the alpha producer when its rows are done: a third of every
#endif                             //    SIMD's registers stays free for waves of another kernel (the segment kernel's polling form)
constexpr int kHxHalo = 8;                    // halo lanes (NP pairs each: the edge lanes are exchanged every NP-th block)
constexpr int kHxOwnLanes = 64 - kHxHalo;     // 56
constexpr int kHxProducers = E2E_HX_SLIM ? 1 : 2;
template <int NP> struct Hx {
  static constexpr int kOwn = NP * kHxOwnLanes;                     // pairs a wave owns
  static constexpr int kMaxW = 4 / NP;                              // chain waves per direction: 224 pairs, S <= 223
  static constexpr int kWaves = E2E_HX_SLIM ? 2 * kMaxW + 2 * kHxProducers + 2 : 2 * kMaxW + 2 + 2 * kHxProducers + 2;
};

struct HxLds {
  static constexpr int kCkPad = 4;                                   // cells in front of a checkpoint buffer (beta's first slot writes cell -1)
  static constexpr int kCkCells = 2 * 4 * kHxOwnLanes + 2 * kCkPad;  // cells of one checkpoint row buffer
  int ring;        // [2][kRingBlks] blocks of blk_bytes: (V+1) label rows of kRow doubles (row V: zeros) + 16 doubles (yb, wb) x 8 steps
  int blk_bytes;
  int filled;      // [2][kRingBlks] ints; used: [dir][f] = the next block producer f of the direction has not finished yet
  int sortcnt;     // [130] ints (cellinfo_wave)
  int bnd;         // [2][bw][kHaloSlots][kHxHalo] x 4 doubles: wave w's edge lanes after block n (wave slots that hold no
                   //  cells stay zero: what the first / last wave of a direction reads as its neighbour's); bw = 8 wave slots
                   //  with one pair per lane, 4 with two (waves 0, 1; slot 2: beta's "wave above the last", 3: alpha's "below the first")
  int bw;
  int dump;        // 12 KB nobody reads: where lanes that have nothing to publish store
  int zacc;        // [8] doubles
  int prog;        // [2][8] ints
  int exw;         // [2][kHaloSlots] ints
  int mxl;         // [2][kHaloSlots][4][64] ints
  int ckb;         // [2][2][kCkCells] doubles: a checkpoint row's true cells in lattice order, double-buffered
  int ckdone;      // [2] ints
  int total;
  __host__ __device__ HxLds(int V, int np = 2) {
    bw = np == 1 ? 8 : 4;
    ring = 0;
    blk_bytes = ((V + 1) * kRow + 16) * 8;
    filled = ring + 2 * kRingBlks * blk_bytes;
    sortcnt = filled + 2 * kRingBlks * 4;
    bnd = (sortcnt + 130 * 4 + 15) & ~15;
    dump = bnd + 2 * bw * kHaloSlots * kHxHalo * 32;
    zacc = dump + 12288;
    prog = zacc + 64;
    exw = prog + 2 * 8 * 4;
    mxl = exw + 2 * kHaloSlots * 4;
    ckb = mxl + 2 * kHaloSlots * 4 * 64 * 4;
    ckdone = ckb + 2 * 2 * kCkCells * 8;
    total = ckdone + 16;
  }
};

#ifdef E2E_FAST_PROFILE
#define HX_TL(k) { if (w == 0 && lane == 0 && b < 256) { unsigned long long* g = g_tl + ((size_t)b * 2 + DIR) * 12 + 2 * (k); \
    g[0] = __builtin_amdgcn_s_memtime(); g[1] = wall_clock64(); } }
#else
#define HX_TL(k)
#endif
template <int DIR, int NP>
__device__ __forceinline__ void hx_chain_wave(const FastParams& p, int b, int T, int S, unsigned char* smem, const HxLds hl,
                                              int lane, int w, int W) {
  lds_u8* L0 = (lds_u8*)smem;
  HX_TL(1)
  const int V = p.V, blank = p.blank, L = 2 * S + 1;
  const int nblk = (T + kBlk - 1) / kBlk;
  const int M = (T - 1) >> 3;
  const int ring_off = hl.ring + DIR * kRingBlks * hl.blk_bytes;
  volatile int* myfilled = reinterpret_cast<int*>(smem + hl.filled) + DIR * kRingBlks;
  __builtin_amdgcn_s_setprio(3);
  unsigned long long prof_fill = 0, prof_nb = 0, prof_lag = 0, prof_t0 = __builtin_amdgcn_s_memtime();
  (void)prof_fill; (void)prof_nb; (void)prof_lag; (void)prof_t0;

  // slot r of the lane: alpha holds (blank gi, label gi), beta (blank gi, label gi - 1), gi = g0 + r
  const int g0 = Hx<NP>::kOwn * w + NP * (DIR == 0 ? lane - kHxHalo : lane);
  const bool halo = DIR == 0 ? lane < kHxHalo : lane >= kHxOwnLanes;
  const bool edge = DIR == 0 ? lane >= 64 - kHxHalo : lane < kHxHalo;
  // the upstream neighbour's progress word and edge records.  A wave without one (alpha's first, beta's last) reads a wave slot
  // that holds no cells -- progress kHaloIdle, records zero --, which is what its halo lanes hold anyway
  const int up = DIR == 0 ? (w > 0 ? w - 1 : 7) : w + 1;
  const int64_t* tg = p.targets + (int64_t)b * p.tgt_stride;
  const float r_tilt = fast_tilt(S, T);
  const double rr2 = (double)r_tilt * (double)r_tilt, inv_rr = 1.0 / (double)r_tilt;
  const double skip_w = (double)(r_tilt * r_tilt);         // (the segment kernel's weight of a skip: the tilt squared in f32)
  int lab[NP]; double sk[NP];
  bool badlab = false;
#pragma unroll
  for (int r = 0; r < NP; r++) {
    const int gi = g0 + r, li = DIR == 0 ? gi : gi - 1;
    const bool lin = li >= 0 && li < S;
    const int lv = lin ? (int)tg[li] : -1;
    lab[r] = (lin && lv >= 0 && lv < V) ? lv : V;          // V: the always-zero row
    if (DIR == 0) {
      const int lpv = (li >= 1 && li - 1 < S) ? (int)tg[li - 1] : -1;
      sk[r] = (lin && li >= 1 && lv != blank && lpv != lv) ? skip_w : 0.0;              // ctc_loss.cpp:53-57
      badlab |= !halo && lin && (lv == blank || lv < 0 || lv >= V);
    } else {
      const int tj = (gi >= 1 && gi < S) ? (int)tg[gi] : -1;                            // the skip label gi-1 -> label gi
      sk[r] = (gi >= 1 && gi < S && lv != blank && tj != lv) ? skip_w : 0.0;            // ctc_loss.cpp:91-96
    }
  }
  if (DIR == 0 && __any(badlab)) { if (lane == 0) atomicOr(&p.flags[b], 2); }
  const bool cond = (T > 1 || L == 1);            // ctc_loss.cpp:39,76

  // LDS addresses that do not change: everything a block touches is one of these plus a constant
  const int a_sync = hl.prog;                                                      // (uniform) the words of both directions
  const int o_prog = DIR * 32, o_exw = hl.exw - hl.prog + DIR * (kHaloSlots * 4), o_ckdone = hl.ckdone - hl.prog + 4 * DIR;
  const int a_mxl = hl.mxl + ((DIR * kHaloSlots * 4 + w) * 64 + lane) * 4;         // + slot * 1024
  const int up_rec = DIR == 0 ? (w > 0 ? w - 1 : hl.bw - 1) : w + 1;          // (its edge records: the last wave slot is alpha's empty one)
  const int a_bnd_up = hl.bnd + ((DIR * hl.bw + up_rec) * kHaloSlots * kHxHalo + (lane & (kHxHalo - 1))) * 32;     // + slot * 256
  // own edge record: the edge lanes' cells; the other lanes write to a dump row of their own (no exec mask around the store)
  const int a_bnd_my = edge ? hl.bnd + ((DIR * hl.bw + w) * kHaloSlots * kHxHalo + (lane & (kHxHalo - 1))) * 32
                            : hl.dump + lane * 32;                                  // (+ slot * 256 < 4 KB)
  // checkpoint cells of slot 0 (label cell first for beta: cells 2 gi - 1, 2 gi; alpha: cells 2 gi, 2 gi + 1); halo lanes: dump
  const int a_ck = halo ? hl.dump + 4096 + lane * 32
                        : hl.ckb + (DIR * 2 * HxLds::kCkCells + HxLds::kCkPad) * 8 + (DIR == 0 ? 16 * g0 : 16 * g0 - 8);
  int a_lab[NP];
#pragma unroll
  for (int r = 0; r < NP; r++) a_lab[r] = ring_off + lab[r] * (kRow * 8);         // + slot * blk_bytes
  const int a_yw = ring_off + (V + 1) * (kRow * 8);                                // (uniform)

  double Bc[NP], Lc[NP];                          // B~ (blank cells before their emission), L^ (label cells, tilted)
#pragma unroll
  for (int r = 0; r < NP; r++) { Bc[r] = 0.0; Lc[r] = 0.0; }
  double yb_prev = 0.0, wb_prev = 0.0;
  int e_total = 0;
  int nck = 0, ckbuf = 0;
  int lead = 0;
  auto need_blocks = [&](int k) {
    if (!(E2E_HX_ABL & 2) && lead < k) {
      PROF_SPIN_BEGIN
      for (;;) {
        int a = peek(&myfilled[0]);
#pragma unroll
        for (int f = 1; f < kHxProducers; f++) a = min(a, peek(&myfilled[f]));
        lead = __builtin_amdgcn_readfirstlane(a);
        if (lead >= k) break;
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      PROF_SPIN_END(prof_fill)
    }
  };

  double e[NP][kBlk], ybw[2 * kBlk];
  // steps 4H .. 4H+3 of the block in ring slot `slot`.  The reads are issued so that what is used FIRST (step 4H) is asked for
  // LAST: the wait in front of that first use then covers the whole half, instead of one wait per register pair
  bool abl_loaded[2] = {false, false};
  auto load_half = [&](int slot_bytes, auto half_tag) {
    constexpr int H = decltype(half_tag)::value;
    if (E2E_HX_ABL & 1) { if (abl_loaded[H]) return; abl_loaded[H] = true; }
    lds_u8* py = L0 + (a_yw + slot_bytes);
#pragma unroll
    for (int q = 3; q >= 1; q--) {
      const h_d2 y = *(volatile lds_d2*)(py + 64 * H + 16 * q);
      ybw[8 * H + 2 * q] = y.x; ybw[8 * H + 2 * q + 1] = y.y;
    }
#pragma unroll
    for (int r = 0; r < NP; r++) {
      const h_d2 a = *(volatile lds_d2*)(L0 + (a_lab[r] + slot_bytes) + 32 * H + 16); e[r][4 * H + 2] = a.x; e[r][4 * H + 3] = a.y;
    }
#pragma unroll
    for (int r = 0; r < NP; r++) {
      const h_d2 a = *(volatile lds_d2*)(L0 + (a_lab[r] + slot_bytes) + 32 * H); e[r][4 * H] = a.x; e[r][4 * H + 1] = a.y;
    }
    { const h_d2 y = *(volatile lds_d2*)(py + 64 * H); ybw[8 * H] = y.x; ybw[8 * H + 1] = y.y; }
  };
  auto read_edge = [&](int slot, h_d2 (&hv)[NP]) {
#pragma unroll
    for (int r = 0; r < NP; r++) hv[r] = *(volatile lds_d2*)(L0 + a_bnd_up + slot * 256 + 16 * r);
  };

  int hprog = 0;                                   // the upstream neighbour's progress as of step 6 of the block before a refill
  h_d2 hv[NP];                                     // ... and its edge record of that block, asked for at the same time
#pragma unroll
  for (int r = 0; r < NP; r++) hv[r] = h_d2{0.0, 0.0};
  int ckd = 0;                                     // checkpoint rows the checkpoint wave had read, as of step 4
  // One block.  STEADY: all 8 rows are live, none is the chain's first row, and the next block exists.  SLOT: n & 7 where the
  // caller knows it at compile time (the interior loop, unrolled over the ring's eight slots), -1: computed.  CK: whether the
  // block ends in a checkpoint row (1 / 0; -1: decided here).  The halo holds NP * 8 pairs: it is refilled before the blocks
  // with n % NP == 0 and the edge lanes are published after the blocks with n % NP == NP - 1.
  auto run_block = [&](int n, auto steady_tag, auto slot_tag, auto ck_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    constexpr int SLOT = decltype(slot_tag)::value, CK = decltype(ck_tag)::value;
    const int slot = SLOT >= 0 ? SLOT : (n & (kRingBlks - 1));
    const int nslot = SLOT >= 0 ? ((SLOT + 1) & (kRingBlks - 1)) : ((n + 1) & (kRingBlks - 1));
    const int pslot = SLOT >= 0 ? ((SLOT + 7) & (kRingBlks - 1)) : ((n + 7) & (kRingBlks - 1));
    const bool refill = SLOT >= 0 ? (SLOT % NP) == 0 : (n % NP) == 0;
    const bool publish = SLOT >= 0 ? (SLOT % NP) == NP - 1 : (n % NP) == NP - 1;
    load_half(slot * hl.blk_bytes, std::integral_constant<int, 1>{});
    if (!(E2E_HX_ABL & 4) && refill && (STEADY || n > 0)) {
      if (__builtin_amdgcn_readfirstlane(hprog) < n) {
        PROF_SPIN_BEGIN HALO_WAIT(__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(L0 + a_sync + o_prog + 4 * up)) >= n); PROF_SPIN_END(prof_nb)
        read_edge(pslot, hv);
      }
      if (halo) {
#pragma unroll
        for (int r = 0; r < NP; r++) { Bc[r] = hv[r].x; Lc[r] = hv[r].y; }
      }
    }
    const bool want_next = STEADY ? true : n + 1 < nblk;
    const int tbase = block_time(DIR, n, 0, T);
    int xw = 0;
#pragma unroll
    for (int tt = 0; tt < kBlk; tt++) {
      const int t = DIR == 0 ? tbase + tt : tbase - tt;
      const double yb = ybw[2 * tt], wb = ybw[2 * tt + 1];
      if (tt == 4) {
        if (!(E2E_HX_ABL & 8)) xw = *(volatile lds_int*)(L0 + a_sync + o_exw + 4 * slot);
        if (CK != 0) ckd = *(volatile lds_int*)(L0 + a_sync + o_ckdone);
        if (want_next) need_blocks(n + 2);
        load_half(nslot * hl.blk_bytes, std::integral_constant<int, 0>{});
      }
      if (!(E2E_HX_ABL & 4) && tt == 6 && publish) {                    // (speculative: valid if the neighbour has finished block n by now)
        hprog = *(volatile lds_int*)(L0 + a_sync + o_prog + 4 * up);
        read_edge(slot, hv);
      }
      if (STEADY || t < T) {
        const bool first = !STEADY && (DIR == 0 ? t == 0 : t == T - 1);
        if (first) {
#pragma unroll
          for (int r = 0; r < NP; r++) {
            if (DIR == 0) {
              if (g0 + r == 0) { Bc[r] = cond ? 1.0 : 0.0; Lc[r] = rr2 * e[r][tt]; }        // ctc_loss.cpp:39-42
            } else {
              if (g0 + r == S) { if (cond) Bc[r] = 1.0; Lc[r] = rr2 * e[r][tt]; }           // ctc_loss.cpp:76,78 (label S-1 sits in this slot)
            }
          }
        } else if (DIR == 0) {
          // alpha_t[j] = (alpha[j] + r*alpha[j-1] + r^2*skip*alpha[j-2]) * y_t[l_j], ctc_loss.cpp:47-60
          double nb = from_prev_lane(Lc[NP - 1]);           // the label cell below the slot's blank
#pragma unroll
          for (int r = 0; r < NP; r++) {
            const double ol = Lc[r];
            const double Bn = __builtin_fma(Bc[r], yb_prev, nb);
            double tl = __builtin_fma(Bc[r], wb_prev, ol);
            tl = __builtin_fma(sk[r], nb, tl);
            Lc[r] = tl * e[r][tt]; Bc[r] = Bn; nb = ol;
          }
        } else {
          // q_t[j] = (q[j] + r*q[j+1] + r^2*skipn*q[j+2]) * y_t[l_j]; q = beta * emission, ctc_loss.cpp:84-99
          double nb = from_next_lane(Lc[0]);                // label(gi) of the slot above the lane's last
#pragma unroll
          for (int r = NP - 1; r >= 0; r--) {
            const double ol = Lc[r];
            const double Bn = __builtin_fma(Bc[r], yb_prev, nb);
            double tl = __builtin_fma(Bc[r], wb_prev, ol);
            tl = __builtin_fma(sk[r], nb, tl);
            Lc[r] = tl * e[r][tt]; Bc[r] = Bn; nb = ol;
          }
        }
        yb_prev = yb; wb_prev = wb;
        if (tt == 7) {
          int top = max(__double2hiint(Bc[0]), __double2hiint(Lc[0]));
#pragma unroll
          for (int r = 1; r < NP; r++) top = max(top, max(__double2hiint(Bc[r]), __double2hiint(Lc[r])));
          if (!(E2E_HX_ABL & 8)) *(volatile lds_int*)(L0 + a_mxl + slot * 1024) = top;
          xw = (E2E_HX_ABL & 8) ? ((n << 12) | (2048 + 30)) : __builtin_amdgcn_readfirstlane(xw);
          if ((xw >> 12) != n) {
            PROF_SPIN_BEGIN
            int spins = 0;
            do {
              __builtin_amdgcn_s_sleep(1);
              xw = __builtin_amdgcn_readfirstlane(*(volatile lds_int*)(L0 + a_sync + o_exw + 4 * slot));
              if (++spins > (1 << 20)) { atomicOr(&p.flags[b], 128); xw = (n << 12) | 2048; }
            } while ((xw >> 12) != n);
            PROF_SPIN_END(prof_lag)
          }
          const int ex = (xw & 0xfff) - 2048;
#pragma unroll
          for (int r = 0; r < NP; r++) { Bc[r] = ldexp(Bc[r], -ex); Lc[r] = ldexp(Lc[r], -ex); }
          e_total += ex;
          const int kk = DIR == 0 ? (t + 1) : t;            // alpha row 16k-1 / beta row 16k -> slot k
          if (!(E2E_HX_ABL & 16) && (CK >= 0 ? CK == 1 : ((kk & (kSeg - 1)) == 0 && kk > 0 && kk < T))) {
            // (two buffers: the checkpoint wave has 16 steps for each and is normally long done with the row before last)
            if (nck >= 2 && __builtin_amdgcn_readfirstlane(ckd) < nck - 1)
              HALO_WAIT(__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(L0 + a_sync + o_ckdone)) >= nck - 1);
            if (nck == 0) ckbuf = (kk / kSeg) & 1;
            nck++;
            // the true cells (blank with its emission, label without the tilt), in lattice order
            lds_u8* dst = L0 + (a_ck + ckbuf * (HxLds::kCkCells * 8));
#pragma unroll
            for (int r = 0; r < NP; r++) {
              const double cb = Bc[r] * yb_prev, cl = Lc[r] * inv_rr;
              if (DIR == 0) { *(volatile lds_f64*)(dst + 16 * r) = cb; *(volatile lds_f64*)(dst + 16 * r + 8) = cl; }
              else { *(volatile lds_f64*)(dst + 16 * r) = cl; *(volatile lds_f64*)(dst + 16 * r + 8) = cb; }
            }
            ckbuf ^= 1;
          }
        }
      }
    }
    if (!(E2E_HX_ABL & 4) && publish) {
#pragma unroll
      for (int r = 0; r < NP; r++) { h_d2 v; v.x = Bc[r]; v.y = Lc[r]; *(volatile lds_d2*)(L0 + a_bnd_my + slot * 256 + 16 * r) = v; }
    }
    *(volatile lds_int*)(L0 + a_sync + o_prog + 4 * w) = n + 1;
  };
  {
    typedef std::integral_constant<int, -1> Any;
    need_blocks(1);
    HX_TL(2)
    load_half(0, std::integral_constant<int, 0>{});
    const int steady_end = DIR == 0 ? T / kBlk : nblk;         // blocks [1, steady_end - 1) are steady (live, with a successor)
    run_block(0, std::false_type{}, Any{}, Any{});
    int n = 1;
#ifndef E2E_HX_PLAIN_LOOP
    // Interior blocks eight at a time, one per ring slot.  alpha: a checkpoint row (t = 16k - 1) ends every odd block; beta:
    // a checkpoint row (t = 16k) ends the blocks of M's parity, M = (T-1)/8.  Blocks 1 .. M-1 qualify (live, not first, a
    // successor, checkpoint rows inside (0, T)).
    if (M - 1 >= 16) {
      for (; n < 8; n++) run_block(n, std::true_type{}, Any{}, Any{});
      auto eight = [&](auto par_tag) {
        constexpr int P = decltype(par_tag)::value;             // parity of the blocks that end in a checkpoint row
#define HX_BLK(K) run_block(n + K, std::true_type{}, std::integral_constant<int, K>{}, std::integral_constant<int, ((K & 1) == P) ? 1 : 0>{});
        for (; n + 8 <= M; n += 8) { HX_BLK(0) HX_BLK(1) HX_BLK(2) HX_BLK(3) HX_BLK(4) HX_BLK(5) HX_BLK(6) HX_BLK(7) }
#undef HX_BLK
      };
      if (DIR == 0 || (M & 1)) eight(std::integral_constant<int, 1>{});
      else eight(std::integral_constant<int, 0>{});
    }
#endif
    for (; n < steady_end - 1; n++) run_block(n, std::true_type{}, Any{}, Any{});
    for (; n < nblk; n++) run_block(n, std::false_type{}, Any{}, Any{});
  }
  HX_TL(3)
#ifdef E2E_FAST_PROFILE
  if (lane == 0 && b < 256) { unsigned long long* g = g_prof3 + ((size_t)b * 16 + DIR * 8 + w) * 4;
    g[0] = __builtin_amdgcn_s_memtime() - prof_t0; g[1] = prof_fill; g[2] = prof_nb; g[3] = prof_lag; }
#endif
  lds_u8* prog = L0 + hl.prog + DIR * 32;
  // ---- log Z from this side ----
  if (DIR == 0) {
    double z = 0.0;
    if (!halo) {
#pragma unroll
      for (int r = 0; r < NP; r++) {
        if (g0 + r == S) z += Bc[r] * yb_prev;                   // ctc_loss.cpp:63-70, un-tilted relative to cell L-1
        if (g0 + r == S - 1) z += Lc[r];                         // (= r * the label cell)
      }
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
        const double lz = log(zs) + (double)e_total * 0.693147180559945309417 - (double)(L - 1) * log(rr);
        p.logz[2 * b] = lz;
        p.zt2[b] = log2(zs) + (double)e_total;
        p.losses[b] = (float)(-lz);
        if (!(zs > 0.0) || !(zs < __builtin_huge_val())) atomicOr(&p.flags[b], 4);     // infeasible or out of range
      }
    }
  } else if (w == 0) {
    // sum_j alpha_0[j]*beta_0[j]: blank 0 is slot 0 of lane 0, label 0 the slot after it
    const double l0 = NP == 1 ? __shfl(Lc[0], 1, 64) : Lc[NP - 1];
    if (lane == 0) {
      const double z = (cond ? Bc[0] * yb_prev : 0.0) + l0;
      p.logz[2 * b + 1] = log(z) + (double)e_total * 0.693147180559945309417 - (double)(L - 1) * log((double)r_tilt);
    }
  }
  HX_TL(4)
}

// The probability rows of one chain, leaner than prep_wave (fast_common.h), which it follows: a block is one pass, each
// group of 8 lanes takes one of the block's 8 time steps, a lane holds the columns v = l8 + 8k, k < NV = ceil(V/8).  What a
// producer costs is again its instruction count -- at ~235 per block prep_wave weighs as much as a chain wave, and the chains
// were waiting for it (tools/diag/h1_wave_profile.py: ring wait) --, so: the direction is a template parameter, the row's
// address is one clamp and one multiply-add from the block number, only the last column groups of a bracket are tested,
// steady blocks (all rows live) have no row tests, the blank's (probability, tilted probability) pair is one masked store
// after the columns, and only the alpha side tracks the smallest log-probability.
template <int NV, int DIR>
__device__ __forceinline__ void hx_prep_wave(const FastParams& p, int b, int T, int first, unsigned char* smem, const HxLds hl,
                                             int lane, double rr2) {
  constexpr int stride = kHxProducers;
  lds_u8* L0 = (lds_u8*)smem;
  const int V = p.V;
  const int nblk = (T + kBlk - 1) / kBlk, M = (T - 1) >> 3;
  const int tt = lane >> 3, l8 = lane & 7;
  const float ninf = -__builtin_huge_valf();
  // NV is the alphabet's bracket (2: V <= 16, 4: <= 32, 6: <= 48, 8: <= 64, 12: <= 96): the first kFull column groups are
  // live in every lane, the others are tested
  constexpr int kFull = NV == 2 ? 0 : NV == 12 ? 8 : NV - 2;
  bool live[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) live[k] = k < kFull || l8 + 8 * k < V;
  const bool contig = p.sV == 1;
  const int64_t xl = (int64_t)b * p.sB + (int64_t)l8 * p.sV;        // this lane's first column (element offset)
  const int64_t cstep = 8 * p.sV;
  float* yl = p.ytab + ((size_t)b * p.NS * V + l8) * kSeg;          // [segment][label][16 steps]
  const int t_first = DIR == 0 ? tt : 8 * M + 7 - tt;                // the lane's row in block 0; block n: t_first +- 8 n
  lds_u8* prog = L0 + hl.prog + DIR * 32;
  const int a_fill = hl.filled + (DIR * kRingBlks + first) * 4;
  const int a_row = hl.ring + DIR * kRingBlks * hl.blk_bytes + (l8 * kRow + tt) * 8;       // + slot * blk_bytes + k * 8 * kRow * 8
  const int a_yw = hl.ring + DIR * kRingBlks * hl.blk_bytes + ((V + 1) * kRow + 2 * tt) * 8;
  const int kb = p.blank >> 3;
  const bool blank_lane = l8 == (p.blank & 7);

  auto row_of = [&](int n) { return DIR == 0 ? t_first + 8 * n : t_first - 8 * n; };
  // unconditional loads from a clamped row (what a dead row or column reads is replaced when it is used), two of this wave's
  // blocks ahead, three register sets rotating through a loop unrolled three times: see prep_wave
  // (F32IN: f32 logits -- the loop below exists twice, so that no test of the dtype sits between the loads)
  auto load_block = [&](auto f32_tag, int n, float (&out)[NV]) {
    constexpr bool F32IN = decltype(f32_tag)::value;
    const int t = row_of(n);
    const int tc = min(max(t, 0), T - 1);
    const int64_t xr = xl + (int64_t)tc * p.sT;
    int64_t off[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) off[k] = (k < kFull || live[k]) ? (contig ? (int64_t)(8 * k) : k * cstep) : 0;
    if (F32IN) {
      const float* src = reinterpret_cast<const float*>(p.x) + xr;
#pragma unroll
      for (int k = 0; k < NV; k++) out[k] = src[off[k]];
    } else {
      const unsigned short* src = reinterpret_cast<const unsigned short*>(p.x) + xr;
      unsigned short h[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) h[k] = src[off[k]];
      const bool bf = p.xdt == E2E_BF16;
#pragma unroll
      for (int k = 0; k < NV; k++)
        out[k] = bf ? __uint_as_float((unsigned)h[k] << 16) : (float)__builtin_bit_cast(f16_t, h[k]);
    }
  };
  int consumed = 0;
  float lpmin = 0.f;
  auto process = [&](int n, const float (&xraw)[NV]) {
    constexpr bool STEADY = false;                                 // (a copy of the body without row tests: 60 instances, not worth it)
    const int t = row_of(n);
    const bool row_live = t >= 0 && t < T;
    float xv[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) xv[k] = (row_live && (k < kFull || live[k])) ? xraw[k] : ninf;
    if (n >= kRingBlks) {
      // (the readers' progress is looked at again only when the last look does not cover this block)
      while (consumed < n - kRingBlks + 1) {
        consumed = __builtin_amdgcn_readfirstlane(lds_min8(prog));
        if (consumed < n - kRingBlks + 1) __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    }
    float y[NV];
    if (p.logprobs) {
#pragma unroll
      for (int k = 0; k < NV; k++) {
        y[k] = exp_le0(xv[k]);
        if (DIR == 0) lpmin = fminf(lpmin, xv[k] > ninf ? xv[k] : 0.f);      // (-inf: an impossible symbol, exact; so are dead rows)
      }
    } else {
      float m = xv[0];
#pragma unroll
      for (int k = 1; k < NV; k++) m = fmaxf(m, xv[k]);
      m = row8_max(m);
      float ssum = 0.f;
#ifndef E2E_H1_FAST_EXP          // exp(x - m) as 2^(x log2(e) - m log2(e)): ONE rounding of the exponent (the fused multiply-add's; the
#define E2E_H1_FAST_EXP 1        // rounding of m log2(e) is common to the row and cancels in the normalisation) and v_exp_f32 -- where exp_le0
#endif                           // rounds x - m and then takes eight instructions for an exponential exact to an ulp of THAT.  Same error
      const float mM = -m * 1.44269504088896340736f;               // bound (half an ulp of an exponent of up to 115), six instructions less per element
#pragma unroll
      for (int k = 0; k < NV; k++) {
        const float d = xv[k] - m;
#ifdef E2E_H1_TWOSUM            // (off: 0.8 us of the headline call -- 127.6 against 126.8 us -- for 4e-6 relative in probabilities below e^-64)
        const float bb = d - xv[k], err = (xv[k] - (d - bb)) + (-m - bb);    // the subtraction's rounding error (see ctc_fast_prob_kernel)
        const float e0 = exp_le0(d);
        y[k] = xv[k] > ninf ? fmaf(e0, err, e0) : 0.f; ssum += y[k];
#elif E2E_H1_FAST_EXP
        y[k] = __builtin_amdgcn_exp2f(fmaf(xv[k], 1.44269504088896340736f, mM)); ssum += y[k];      // (2^-inf = 0: dead columns and rows)
#else
        y[k] = exp_le0(d); ssum += y[k];
#endif
        if (DIR == 0) lpmin = fminf(lpmin, xv[k] > ninf ? d : 0.f);          // (>= the log-probability)
      }
      ssum = row8_sum(ssum);
      float inv = __builtin_amdgcn_rcpf(ssum);
      inv = fmaf(fmaf(-ssum, inv, 1.0f), inv, inv);        // one Newton step: ~0.5 ulp
#pragma unroll
      for (int k = 0; k < NV; k++) y[k] *= inv;
    }
    if (!STEADY) {
#pragma unroll
      for (int k = 0; k < NV; k++) y[k] = row_live ? y[k] : 0.f;
    }
    const int sb = (n & (kRingBlks - 1)) * hl.blk_bytes;
    lds_u8* dst = L0 + (a_row + sb);
#pragma unroll
    for (int k = 0; k < NV; k++) {
      if (k < kFull || live[k]) *(volatile lds_f64*)(dst + k * (8 * kRow * 8)) = (double)y[k];             // transposed: [label][step]
    }
    {
      float ybl = y[0];
#pragma unroll
      for (int k = 1; k < NV; k++) ybl = kb == k ? y[k] : ybl;
      if (blank_lane) { h_d2 yw; yw.x = (double)ybl; yw.y = rr2 * (double)ybl; *(volatile lds_d2*)(L0 + (a_yw + sb)) = yw; }
    }
    if (DIR == 0 && row_live) {
      float* yrow = yl + (size_t)(t >> 4) * V * kSeg + (t & (kSeg - 1));
#pragma unroll
      for (int k = 0; k < NV; k++) {
        if (k < kFull || live[k]) yrow[8 * k * kSeg] = y[k];
      }
    }
    // every lane stores the same word ("my blocks up to n are there"): no divergence, one LDS write
    *(volatile lds_int*)(L0 + a_fill) = n + stride;
  };
  auto run = [&](auto f32_tag) {
    float xa[NV], xb[NV], xc[NV];
    load_block(f32_tag, first, xa);
    load_block(f32_tag, first + stride, xb);
    for (int n = first; n < nblk; n += 3 * stride) {       // this wave fills every `stride`-th block
      load_block(f32_tag, n + 2 * stride, xc); process(n, xa);
      load_block(f32_tag, n + 3 * stride, xa); if (n + stride < nblk) process(n + stride, xb);
      load_block(f32_tag, n + 4 * stride, xb); if (n + 2 * stride < nblk) process(n + 2 * stride, xc);
    }
  };
  if (p.xdt == E2E_F32) run(std::true_type{}); else run(std::false_type{});
  // (see prep_wave: emissions near the end of f32 -> the utterance is recomputed entirely by the exact kernel)
  { const bool t1 = __any(lpmin < -69.f), t2 = __any(lpmin < -78.f);      // (both votes by the whole wave, outside the lane test)
    if (DIR == 0 && t1 && lane == 0) atomicOr(&p.flags[b], t2 ? 64 | 256 : 64); }       // e^-69 = 2^-100
}

// The checkpoint wave of a direction: reads a checkpoint row's true cells in lattice order (pair i: cells 2i, 2i+1), finds the
// exponent of every group of F2PPL pairs (the segment kernel's lanes), scales, and stores cells and exponents.
template <int DIR, int F2PPL>
__device__ __forceinline__ void hx_ckpt_wave(const FastParams& p, int b, int T, int S, lds_u8* L0, const HxLds hl, int lane, int npairs) {
  static_assert(F2PPL == 1 || F2PPL == 2 || F2PPL == 4, "groups of up to four lanes (one DPP quad)");
  const int nblk = (T + kBlk - 1) / kBlk;
  const int nres = DIR == 0 ? T / kBlk : nblk;
  const int M = (T - 1) >> 3;
  lds_u8* prog = L0 + hl.prog + DIR * 32;
  float* ck = (DIR == 0 ? p.ckA : p.ckQ) + (size_t)b * p.NS * p.CELLS;
  int done = 0;
  for (int n = 0; n < nres; n++) {
    const int kk = DIR == 0 ? 8 * (n + 1) : 8 * (M - n);
    if (!((kk & (kSeg - 1)) == 0 && kk > 0 && kk < T)) continue;
    HALO_WAIT(__builtin_amdgcn_readfirstlane(lds_min8(prog)) >= n + 1);
    const int slot = kk / kSeg;
    const int base = hl.ckb + ((DIR * 2 + (slot & 1)) * HxLds::kCkCells + HxLds::kCkPad) * 8;
    h_d2 c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = 64 * q + lane;
      c[q] = h_d2{0.0, 0.0};
      if (64 * q < npairs) c[q] = *(volatile lds_d2*)(L0 + base + 16 * min(i, npairs - 1));
      // cells past the lattice are zero (beta: no lane holds the label cell of the last pair, its buffer cell is never written)
      if (i >= S) c[q].y = 0.0;
      if (i > S) c[q].x = 0.0;
    }
    *(volatile lds_int*)(L0 + hl.ckdone + 4 * DIR) = ++done;       // (LDS runs this wave's operations in order: the reads are done)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (64 * q >= npairs) break;
      const int i = 64 * q + lane;
      int m = max(__double2hiint(c[q].x), __double2hiint(c[q].y));
      if (F2PPL >= 2) m = max(m, dpp_i<0xB1>(0, m));               // quad_perm [1,0,3,2]: the lane pair
      if (F2PPL >= 4) m = max(m, dpp_i<0x4E>(0, m));               // quad_perm [2,3,0,1]: the (aligned) four lanes
      const int own = ((m >> 20) & 0x7ff) - 1023;
      const int st = m > 0 ? own : -30000;
      if (i < npairs && (i & ~(F2PPL - 1)) <= S) {                 // (groups past the lattice are not read)
        float2 o;
        o.x = m > 0 ? (float)ldexp(c[q].x, -own) : 0.f; o.y = m > 0 ? (float)ldexp(c[q].y, -own) : 0.f;
        *reinterpret_cast<float2*>(ck + (size_t)slot * p.CELLS + 2 * i) = o;
        short* cke = p.ckE + (((size_t)b * p.NS + slot) * 2 + DIR) * 64;
        if ((i & (F2PPL - 1)) == 0) cke[i / F2PPL] = (short)st;
      }
    }
  }
}

// E2E_HX_SLIM: the frame wave and the checkpoint wave of a direction as ONE wave.  Both follow the chain waves block by block
// (the same progress words); the frame's word is wanted two blocks later, the checkpoint row's buffer 16 steps later, so per
// block the frame comes first (at the chains' priority) and a checkpoint row, every other block, behind it.
template <int DIR, int F2PPL>
__device__ __forceinline__ void hx_frame_ckpt_wave(const FastParams& p, int b, int T, int S, lds_u8* L0, const HxLds hl, int lane, int W, int npairs) {
  static_assert(F2PPL == 1 || F2PPL == 2 || F2PPL == 4, "groups of up to four lanes (one DPP quad)");
  constexpr int LAG = kHaloLag, maxw = 4, kExMax = 1000;
  const int nblk = (T + kBlk - 1) / kBlk;
  const int nres = DIR == 0 ? T / kBlk : nblk;
  const int M = (T - 1) >> 3;
  int* cum = (DIR == 0 ? p.cumA : p.cumB) + (size_t)b * p.NB;
  int* trk = (DIR == 0 ? p.trkA : p.trkB) + (size_t)b * p.NB;
  lds_u8* prog = L0 + hl.prog + DIR * 32;
  lds_u8* exw = L0 + hl.exw + DIR * (kHaloSlots * 4);
  lds_u8* mxl = L0 + hl.mxl + (DIR * kHaloSlots * maxw * 64 + lane) * 4;
  float* ck = (DIR == 0 ? p.ckA : p.ckQ) + (size_t)b * p.NS * p.CELLS;
  if (lane == 0) {
    if (DIR == 0) { cum[0] = 0; trk[0] = 0; }
    else { cum[M + 1] = 0; cum[M + 2] = 0; trk[M + 1] = 0; trk[M + 2] = 0; }
    for (int n = 0; n < LAG && n < nres; n++) cum[DIR == 0 ? n + 1 : M - n] = 0;
  }
  int through = 0, ex1 = 0, ex2 = 0, absolute = 0, done = 0;
  for (int n = 0; n < nres; n++) {
    __builtin_amdgcn_s_setprio(3);
    HALO_WAIT(__builtin_amdgcn_readfirstlane(lds_min8(prog)) >= n + 1);
    // ---- the frame (halo_frame_wave<DIR, false, kHaloLag, true>) ----
    int m = 0;
    for (int w = 0; w < W; w++) m = max(m, *(volatile lds_int*)(mxl + ((n & (kHaloSlots - 1)) * maxw + w) * 256));
    m = wave_max(m);
    if (m > 0) {
      const int e = ((m >> 20) & 0x7ff) - 1023;
      absolute = e + (through - ex1 - ex2);
    }
    if (lane == 0) trk[DIR == 0 ? n + 1 : M - n] = absolute;
    if (n + LAG < nres) {
      const int ex = m > 0 ? max(min(absolute - through, kExMax), -kExMax) : 0;
      through += ex; ex2 = ex1; ex1 = ex;
      const int nn = n + LAG;
      *(volatile lds_int*)(exw + 4 * (nn & (kHaloSlots - 1))) = (nn << 12) | (ex + 2048);
      if (lane == 0) cum[DIR == 0 ? nn + 1 : M - nn] = through;
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- the checkpoint row that block n ended in, if any (hx_ckpt_wave<DIR, F2PPL>) ----
    const int kk = DIR == 0 ? 8 * (n + 1) : 8 * (M - n);
    if (!((kk & (kSeg - 1)) == 0 && kk > 0 && kk < T)) continue;
    const int slot = kk / kSeg;
    const int base = hl.ckb + ((DIR * 2 + (slot & 1)) * HxLds::kCkCells + HxLds::kCkPad) * 8;
    h_d2 c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = 64 * q + lane;
      c[q] = h_d2{0.0, 0.0};
      if (64 * q < npairs) c[q] = *(volatile lds_d2*)(L0 + base + 16 * min(i, npairs - 1));
      if (i >= S) c[q].y = 0.0;
      if (i > S) c[q].x = 0.0;
    }
    *(volatile lds_int*)(L0 + hl.ckdone + 4 * DIR) = ++done;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (64 * q >= npairs) break;
      const int i = 64 * q + lane;
      int mm = max(__double2hiint(c[q].x), __double2hiint(c[q].y));
      if (F2PPL >= 2) mm = max(mm, dpp_i<0xB1>(0, mm));
      if (F2PPL >= 4) mm = max(mm, dpp_i<0x4E>(0, mm));
      const int own = ((mm >> 20) & 0x7ff) - 1023;
      const int st = mm > 0 ? own : -30000;
      if (i < npairs && (i & ~(F2PPL - 1)) <= S) {
        float2 o;
        o.x = mm > 0 ? (float)ldexp(c[q].x, -own) : 0.f; o.y = mm > 0 ? (float)ldexp(c[q].y, -own) : 0.f;
        *reinterpret_cast<float2*>(ck + (size_t)slot * p.CELLS + 2 * i) = o;
        short* cke = p.ckE + (((size_t)b * p.NS + slot) * 2 + DIR) * 64;
        if ((i & (F2PPL - 1)) == 0) cke[i / F2PPL] = (short)st;
      }
    }
  }
}

// Waves: 2 * kMaxW chain waves (alpha0, beta0, alpha1, beta1, ...: waves of a workgroup land on the SIMDs in the order 0,2,1,3),
// the two frame waves, four probability-row waves (alternating alpha side / beta side), the two checkpoint waves (the second
// writes the lattice description first).
template <int PPL, int NP>
__global__ E2E_KERNEL_ALIGN __launch_bounds__(Hx<NP>::kWaves * 64) void ctc_fast_chain_hx_kernel(FastParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V;
  const HxLds hl(V, NP);
#ifdef E2E_FAST_PROFILE
  if (tid < 128 && lane == 0 && b < 256) { unsigned long long* g = g_tl + ((size_t)b * 2 + (tid >> 6)) * 12; g[0] = __builtin_amdgcn_s_memtime(); g[1] = wall_clock64(); }
#endif

  if (b == 0 && tid < 32) p.ctl[tid] = 0;
  const int64_t Tq = p.x_len[b], Sq = p.t_len[b];
  const bool bad = Tq < 1 || Tq > p.T || Sq < 0 || Sq > p.Smax;
  if (bad) {                       // the exact kernel poisons this utterance
    if (tid == 0) { p.flags[b] = 1; p.losses[b] = __builtin_nanf(""); }   // reason bit 0: bad lengths
    return;
  }
  const int T = (int)Tq, S = (int)Sq;
  const int W = min(S / Hx<NP>::kOwn + 1, Hx<NP>::kMaxW);         // waves that hold a cell: pairs 0..S (pair S = the last blank)
  if (tid == 0) p.flags[b] = 0;
  for (int i = tid; i < p.MW; i += blockDim.x) p.segmask[(size_t)b * p.MW + i] = 0u;
  if (tid < 2 * kRingBlks) reinterpret_cast<int*>(smem + hl.filled)[tid] = (tid & (kRingBlks - 1)) < kHxProducers ? (tid & (kRingBlks - 1)) : 0;
  if (tid < 16) reinterpret_cast<int*>(smem + hl.prog)[tid] = (tid & 7) < W ? 0 : kHaloIdle;
  if (tid < 2 * kHaloSlots) reinterpret_cast<int*>(smem + hl.exw)[tid] = (tid & (kHaloSlots - 1)) < kHaloLag ? ((tid & (kHaloSlots - 1)) << 12) | 2048 : -1;
  if (tid < 8) reinterpret_cast<double*>(smem + hl.zacc)[tid] = 0.0;
  if (tid < 2) reinterpret_cast<int*>(smem + hl.ckdone)[tid] = 0;
  for (int i = tid; i < (hl.dump - hl.bnd) / 16; i += blockDim.x) reinterpret_cast<double2*>(smem + hl.bnd)[i] = double2{0.0, 0.0};
  for (int i = tid; i < 2 * kRingBlks * kBlk; i += blockDim.x)      // the zero rows (label index V) of every block
    reinterpret_cast<double*>(smem + hl.ring + (i / kBlk) * hl.blk_bytes)[V * kRow + (i % kBlk)] = 0;
  __syncthreads();

  constexpr int kChains = 2 * Hx<NP>::kMaxW, kFrame = E2E_HX_SLIM ? kChains + 2 * kHxProducers : kChains,
                kProd = E2E_HX_SLIM ? kChains : kChains + 2, kCkpt = kProd + 2 * kHxProducers;
  const int wave = __builtin_amdgcn_readfirstlane(wid);
  lds_u8* L0 = (lds_u8*)smem;
  if (wave < kChains) {
    const int d = wave & 1, w = wave >> 1;
    if (w >= W) return;
    if (d == 0) hx_chain_wave<0, NP>(p, b, T, S, smem, hl, lane, w, W);
    else hx_chain_wave<1, NP>(p, b, T, S, smem, hl, lane, w, W);
  } else if (E2E_HX_SLIM && wave >= kFrame) {
    if (wave == kFrame) hx_frame_ckpt_wave<0, PPL>(p, b, T, S, L0, hl, lane, W, Hx<NP>::kOwn * W);
    else hx_frame_ckpt_wave<1, PPL>(p, b, T, S, L0, hl, lane, W, Hx<NP>::kOwn * W);
  } else if ((E2E_HX_ABL & 8) && (wave == kFrame || wave == kFrame + 1)) { return;
  } else if ((E2E_HX_ABL & 16) && (wave == kCkpt || wave == kCkpt + 1)) { return;
  } else if ((E2E_HX_ABL & 2) && wave >= kProd && wave < kCkpt) { return;
  } else if (wave == kFrame) halo_frame_wave<0, false, kHaloLag, true>(p, b, T, L0, hl.prog, hl.exw, hl.mxl, 4, lane, W, 0);
  else if (wave == kFrame + 1) halo_frame_wave<1, false, kHaloLag, true>(p, b, T, L0, hl.prog, hl.exw, hl.mxl, 4, lane, W, 0);
  else if (wave == kCkpt) hx_ckpt_wave<0, PPL>(p, b, T, S, L0, hl, lane, Hx<NP>::kOwn * W);
  else if (wave == kCkpt + 1) {
    cellinfo_wave<PPL>(p, b, T, S, reinterpret_cast<int*>(smem + hl.sortcnt), lane);
    hx_ckpt_wave<1, PPL>(p, b, T, S, L0, hl, lane, Hx<NP>::kOwn * W);
  } else {
    const int d = (wave - kProd) & 1;                    // alternating: alpha rows, beta rows
    const int first = (wave - kProd) >> 1;               // the producers of a direction take every kHxProducers-th block
    const double rr2 = (double)fast_tilt(S, T) * (double)fast_tilt(S, T);        // (the chain waves' own expression)
#define HX_PREP(NVV) { if (d == 0) hx_prep_wave<NVV, 0>(p, b, T, first, smem, hl, lane, rr2); else hx_prep_wave<NVV, 1>(p, b, T, first, smem, hl, lane, rr2); }
    if (V <= 16) HX_PREP(2) else if (V <= 32) HX_PREP(4) else if (V <= 48) HX_PREP(6) else if (V <= 64) HX_PREP(8) else HX_PREP(12)
#undef HX_PREP
    // (eight waves: the lattice description -- only the segment kernel reads it -- by the alpha producer, whose rows are all in
    //  the ring a ring's depth of blocks before the chains end)
    if (E2E_HX_SLIM && d == 0 && first == 0) cellinfo_wave<PPL>(p, b, T, S, reinterpret_cast<int*>(smem + hl.sortcnt), lane);
  }
}

template <int PPL, int NP>
int launch_hx(const FastParams& p, hipStream_t stream) {
  const HxLds hl(p.V, NP);
  E2E_HIP_CHECK(allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_fast_chain_hx_kernel<PPL, NP>), hl.total), "hipFuncSetAttribute");
  hipLaunchKernelGGL((ctc_fast_chain_hx_kernel<PPL, NP>), dim3(p.B), dim3(Hx<NP>::kWaves * 64), hl.total, stream, p);
  E2E_HIP_CHECK(hipGetLastError(), "ctc_fast_chain_hx_kernel launch");
  return E2E_OK;
}

}  // namespace

bool h1_supported(int V, int Smax, int ppl) {
  return (ppl == 2 || ppl == 4) && Smax + 1 <= 4 * kHxOwnLanes && HxLds(V).total <= 160 * 1024;
}

int launch_fast_h1_chain(const FastParams& p, int ppl, hipStream_t stream) {
  static const char* np_env = getenv("E2E_F1_NP");                   // (A/B: 1 = one pair per lane on four waves per direction)
  const bool one = np_env && np_env[0] == '1';
  if (ppl == 4) return one ? launch_hx<4, 1>(p, stream) : launch_hx<4, 2>(p, stream);
  if (ppl == 2) return launch_hx<2, 2>(p, stream);
  set_error("lean halo chains: %d pairs per segment-kernel lane", ppl);
  return E2E_ERR_UNSUPPORTED;
}

}  // namespace fastk
}  // namespace e2e

#ifdef E2E_FAST_PROFILE
extern "C" int e2e_debug_fast_timeline_h1(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(e2e::fastk::g_tl), sizeof(unsigned long long) * 256 * 2 * 12) == hipSuccess ? 0 : E2E_ERR_HIP;
}
extern "C" int e2e_debug_fast_profile3_h1(unsigned long long* host, int reset) {
  if (reset) { void* ptr; if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(e2e::fastk::g_prof3)) != hipSuccess) return E2E_ERR_HIP;
    return hipMemset(ptr, 0, sizeof(unsigned long long) * 256 * 16 * 4) == hipSuccess ? 0 : E2E_ERR_HIP; }
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(e2e::fastk::g_prof3), sizeof(unsigned long long) * 256 * 16 * 4) == hipSuccess ? 0 : E2E_ERR_HIP;
}
#endif

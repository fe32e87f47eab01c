// Extended-range redo of utterances the fast CTC path cannot settle (included by ctc_loss_exact.hip, inside its anonymous
// namespace, after ExactParams): emissions that contradict the targets sharply -- mislabelled utterances in front of a
// trained model, logits of scale 8 against unrelated targets.  There the lattice rows span thousands of bits: alpha's mass
// and beta's sit on different cells, a row scaled by ONE power of two (the fast path's chains, the exact kernel's scaled
// form) flushes cells whose posterior matters, and what was left was the log-domain walk -- ~10 ms per utterance on one
// workgroup, 30-220 ms per call.
//
// Here every lattice cell carries its own exponent: value = m * 2^e, m an f64, e an int.  A step aligns the (up to three)
// terms of a cell to the largest exponent among them -- nothing is ever flushed that f64 addition would not drop anyway --
// and the mantissas are renormalised every fourth step (a probability is >= 2^-100 or exactly zero here: smaller ones
// carry flag bit 64 and go to the exact kernel; so a mantissa drifts by < 2^-401 between renormalisations).  No tilt, no
// common frame, no log: ~22 instructions per label pair and step instead of 4, but no exp() / log() per cell either.
// Same recurrences as the reference (src/losses/ctc_loss.cpp:33-117) in the probability domain.
//
// Structure = the fast path's own, run again for the flagged utterances only, inside the flagged launch:
//   * chains: ONE workgroup per utterance, waves (alpha w, beta w), w < 4, each owning 56 lanes x NP label pairs and
//     carrying 8 halo lanes of its upstream neighbour's pairs (mass moves at most one pair per step, so the halo lasts
//     8 NP steps; then the edge lanes are exchanged through LDS behind a workgroup barrier).  Probabilities come from the
//     table the fast path's producers left (ytab).  Every 16 steps a wave stores its row as a checkpoint: f32 mantissas
//     over the fast path's own checkpoint arrays, one int exponent per cell beside them.  ~0.07 (NP = 1) / 0.13 (NP = 2)
//     us per frame instead of 1.6 (scaled form) or 10 (log domain).
//   * segments: one workgroup per (utterance, 16-step segment), a wave per 32 label pairs (64 held: 16 of halo on either
//     side, so the 16 steps need no exchange at all); alpha rows of the segment in registers, beta walks back, the
//     posteriors alpha beta / Z are ordinary doubles in [0, 1] (THEIR underflow is exact) and are summed per label in LDS.
//   Every workgroup of the launch takes segments; utterances are handed from the chains to the segments through flag bits.
#pragma once
#include <type_traits>

#ifndef E2E_EXT_ABL                 // tools/diag: timing builds with parts of the chain waves' work switched off (results meaningless)
#define E2E_EXT_ABL 0               //  1: probabilities loaded once, 2: no lattice arithmetic, 4: no halo exchange, 8: no checkpoint stores
#endif
constexpr int kXZero = -(1 << 30);            // exponent of a zero cell (below anything a row of 2^22 frames can reach)
constexpr int kXHalo = 8, kXOwnLanes = 64 - kXHalo;
constexpr int kExtDone = 2048;                // flag bit: the extended-range chains of the utterance are done (checkpoints, Z, loss)
constexpr int kExtBad = 4096;                 // flag bit: ... and could not settle it (no alignment, a wait that timed out): exact kernel
constexpr int kExtMaxList = 1024;             // utterances of one call the extended-range redo takes (the rest: exact kernel)
constexpr int kExtHalfA = 8192, kExtHalfB = 16384;   // flag bits: the alpha / the beta chains of the utterance have finished ON A WORKGROUP OF THEIR OWN
                                              // (few flagged utterances: see ext_chains); the side that finds the other one's bit combines

__device__ __forceinline__ int x_fix(double m, int e) { return m != 0.0 ? e : kXZero; }
__device__ __forceinline__ void x_norm(double& m, int& e) {          // mantissa back to [0.5, 1); zero stays (0, kXZero)
  e += __builtin_amdgcn_frexp_exp(m);
  m = __builtin_amdgcn_frexp_mant(m);
}
// lane n <- lane n-1 / n+1 of an exponent; the lane without a source gets a zero cell's exponent
__device__ __forceinline__ int x_shift_up_e(int e) { return __builtin_amdgcn_update_dpp(kXZero, e, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int x_shift_down_e(int e) { return __builtin_amdgcn_update_dpp(kXZero, e, 0x130, 0xf, 0xf, false); }

// One step of a label pair, both directions (alpha: slot (blank g, label g), P = label g-1 of the row before; beta with
// emission: slot (blank g, label g-1), P = label g of the row after -- the mirror image, see ctc_loss_fast_h1.hip):
//   B' = (B + P) yb,   L' = (L + B + skip P) yl              ctc_loss.cpp:47-60 / 84-99
// A result is zero exactly when its emission is zero or every term is (then the largest exponent IS kXZero): a term that is
// not zero never underflows (the mantissas are renormalised every fourth step and lose < 2^-127 per step), nothing is
// subtracted.  So the new exponents need only the emissions' zero tests -- known before the step -- and the exponent
// chain (shift, max, select) runs beside the mantissa chain (shift, ldexp, add, multiply) instead of behind its compare.
__device__ __forceinline__ void x_pair_step(double& Bm, int& Be, double& Lm, int& Le, double Pm, int Pe, double sk,
                                            double yb, double yl) {
  const int PeL = sk != 0.0 ? Pe : kXZero;
  const int eB = max(Be, Pe), eL = max(Le, max(Be, PeL));
  const double nB = (ldexp(Bm, Be - eB) + ldexp(Pm, Pe - eB)) * yb;
  const double nL = fma(sk, ldexp(Pm, PeL - eL), ldexp(Lm, Le - eL) + ldexp(Bm, Be - eL)) * yl;
  Bm = nB; Be = yb != 0.0 ? eB : kXZero;
  Lm = nL; Le = yl != 0.0 ? eL : kXZero;
}

struct ExtLds {
  // byte offsets into the workgroup's dynamic LDS
  static constexpr int kRec = 32;                                   // an edge record: Bm, Lm (doubles), Be, Le (ints), padding
  static constexpr int edge = 0;                                    // [2 dirs][4 waves][2 buffers][8 lanes][2 slots] records
  static constexpr int zrec = edge + 2 * 4 * 2 * kXHalo * 2 * kRec; // [2 dirs][2] (m, e as double): the cells of Z
  static constexpr int tiles = zrec + 64;                           // [2 dirs][2 buffers] probability tiles of tile_bytes(V)
  __host__ __device__ static size_t tile_bytes(int V) { return ((size_t)V * 8 * sizeof(float) + 15) & ~(size_t)15; }
  __host__ __device__ static size_t chain_bytes(int V) { return tiles + 4 * tile_bytes(V); }
  // segments: post[16][V + 1] doubles
  __host__ __device__ static size_t seg_bytes(int V) { return sizeof(double) * 16 * ((size_t)V + 1); }
  __host__ __device__ static size_t bytes(int V) { return seg_bytes(V) > chain_bytes(V) ? seg_bytes(V) : chain_bytes(V); }
};

// probabilities of one 16-step segment for a lane's labels: y[tt] of label `lab` (clamped by the caller) out of ytab
__device__ __forceinline__ void x_load_rows(const FastRetry& rt, int b, int seg, int Tmax, int V, int lab, float (&out)[16]) {
  if (rt.ytab_segments) {
    const float4* src = reinterpret_cast<const float4*>(rt.ytab + (((size_t)b * rt.NS + seg) * V + lab) * kFastSeg);
#pragma unroll
    for (int q = 0; q < 4; q++) { const float4 v = src[q]; out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w; }
  } else {
    const float* src = rt.ytab + ((size_t)b * Tmax + (size_t)seg * kFastSeg) * V + lab;
    const int nrow = Tmax - seg * kFastSeg;                      // rows of the table that exist from here
#pragma unroll
    for (int q = 0; q < 16; q++) out[q] = src[(size_t)(q < nrow ? q : 0) * V];
  }
}

// ---- the chains ------------------------------------------------------------------------------------------------------
// All 8 waves of the workgroup call this (wave = 2 w + DIR); waves that hold no cell only keep the barriers.
// global-memory accesses spelled as such: inside this (very large) kernel the compiler no longer proves that the pointers of the
// parameter block are global ones, and a FLAT load counts on lgkmcnt as well -- every wait for LDS then waits for HBM too
typedef __attribute__((address_space(1))) const float g_cfloat;
typedef float x_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const x_f4 g_cfloat4;
typedef __attribute__((address_space(1))) float g_float;
typedef __attribute__((address_space(1))) int g_int;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) x_f4 lds_f32x4;

// Probability tiles: the 8 steps t = 8 m .. 8 m + 7 of every column, [column][8 steps] floats, staged in LDS by the four waves of
// a direction together (256 threads, coalesced out of the fast path's table) one tile ahead of its use, double-buffered.  A
// lane reads its labels' rows of a tile with two 16-byte LDS reads each.  (First form: every lane loaded its own labels' rows
// of the next 16 steps into registers -- the compiler's waits for those loads, and with them for the checkpoint stores, sat in
// front of every segment: 327 us per 1000 frames of which 92 were arithmetic.)
constexpr int kXTile = 8;
constexpr int kXTileLoads = 14;                                     // elements a thread fetches per tile at most (448 columns x 8 / 256)
template <int DIR>
struct XTileFetch {
  x_f4 q;                        // segment-major table (<= 96 columns): one 16-byte piece
  float e[kXTileLoads];          // row-major table: elements idx = d_tid + 256 i of the tile's [8][V]
  // ask for tile m of utterance b (clamped to the utterance's tiles by the caller)
  __device__ __forceinline__ void load(const FastRetry& rt, int b, int m, int Tmax, int V, int d_tid) {
    if (rt.ytab_segments) {
      // [b][seg][V][16]: tile m = half (m & 1) of segment m >> 1; piece j of the tile: column j >> 1, steps 4 (j & 1) .. + 3
      const int j = d_tid < 2 * V ? d_tid : 0;
      g_cfloat4* src = (g_cfloat4*)(rt.ytab + (((size_t)b * rt.NS + (m >> 1)) * V + (j >> 1)) * kFastSeg + 8 * (m & 1) + 4 * (j & 1));
      q = *src;
    } else {
      g_cfloat* src = (g_cfloat*)(rt.ytab + ((size_t)b * Tmax + (size_t)m * kXTile) * V);
      const int cnt = min(kXTile, Tmax - m * kXTile) * V;           // (rows of the table that exist)
#pragma unroll
      for (int i = 0; i < kXTileLoads; i++) { const int idx = d_tid + 256 * i; e[i] = src[idx < cnt ? idx : 0]; }
    }
  }
  // ... and leave it in the LDS buffer `dst` ([V][8] floats)
  __device__ __forceinline__ void store(const FastRetry& rt, unsigned char* dst, int V, int d_tid) const {
    if (rt.ytab_segments) {
      if (d_tid < 2 * V) *(lds_f32x4*)(dst + 16 * d_tid) = q;
    } else {
      const float rV = 1.0f / (float)V;
#pragma unroll
      for (int i = 0; i < kXTileLoads; i++) {
        const int idx = d_tid + 256 * i;
        // (idx / V by a float product: idx + 0.5 is never within 0.5 / V of a multiple of V, f32 rounding is 1e-6 of that)
        if (idx < kXTile * V) { const int tt = (int)(((float)idx + 0.5f) * rV), v = idx - tt * V; *(lds_f32*)(dst + (v * kXTile + tt) * 4) = e[i]; }
      }
    }
  }
};

// ---- the chains ------------------------------------------------------------------------------------------------------
// All 8 waves of the workgroup call this (wave = 2 w + DIR); waves that hold no cell keep the barriers and stage tiles.
// Tiles and exchanges: the 8-step tiles t = 8 m .. 8 m + 7 are walked m = 0 .. M by alpha and M .. 0 by beta (M = (T-1) / 8); at
// every tile boundary the waves publish their edge lanes, meet at a barrier (waiting for LDS only), refill their halo (which
// lasts 8 NP steps: at NP = 2 it is refreshed twice as often as it must be), stage the next tile, ask for the one after.
template <int DIR, int NP, typename LT>
__device__ __forceinline__ void ext_chain_wave(const ExactParams& p, unsigned char* smem, int b, int T, int S, int w, int lane) {
  // (w >= 4: a wave of a workgroup that runs ONE direction on all its eight waves -- it holds no cell, keeps the barriers and leaves
  //  the tiles to the first four)
  const bool stager = __builtin_amdgcn_readfirstlane(w < 4 ? 1 : 0) != 0;
  constexpr int kOwn = NP * kXOwnLanes;                             // pairs a wave owns
  const FastRetry& rt = p.retry;
  const int V = p.V, blank = p.blank, L = 2 * S + 1, Tmax = p.T;
  const int M = (T - 1) >> 3;
  const int W = (S + kOwn) / kOwn;                                  // waves that hold a cell: pairs 0 .. S
  const bool active = __builtin_amdgcn_readfirstlane(w < W ? 1 : 0) != 0;
  const bool cond = (T > 1 || L == 1);                              // ctc_loss.cpp:39,76
  const int64_t* tg = p.targets + (int64_t)b * p.tgt_stride;
  // slot r of the lane: pair g = g0 + r; alpha holds (blank g, label g), beta (blank g, label g - 1)
  const int g0 = kOwn * w + NP * (DIR == 0 ? lane - kXHalo : lane);
  const bool halo = DIR == 0 ? lane < kXHalo : lane >= kXOwnLanes;
  const bool edge = DIR == 0 ? lane >= 64 - kXHalo : lane < kXHalo;
  int lab[NP]; double sk[NP]; bool lvalid[NP];
#pragma unroll
  for (int r = 0; r < NP; r++) {
    const int g = g0 + r, li = DIR == 0 ? g : g - 1;
    lvalid[r] = li >= 0 && li < S;
    const int lv = lvalid[r] ? (int)tg[li] : 0;
    lab[r] = min(max(lv, 0), V - 1);
    if (DIR == 0) {
      const int lpv = li >= 1 && li < S ? (int)tg[li - 1] : -1;
      sk[r] = (lvalid[r] && li >= 1 && lv != blank && lpv != lv) ? 1.0 : 0.0;                 // ctc_loss.cpp:53-57
    } else {
      const int lnv = li >= 0 && li + 1 < S ? (int)tg[li + 1] : -1;                          // P = label li + 1
      sk[r] = (lvalid[r] && li + 1 < S && lv != blank && lnv != lv) ? 1.0 : 0.0;              // ctc_loss.cpp:91-96
    }
  }
  double Bm[NP], Lm[NP]; int Be[NP], Le[NP];
#pragma unroll
  for (int r = 0; r < NP; r++) { Bm[r] = 0.0; Lm[r] = 0.0; Be[r] = kXZero; Le[r] = kXZero; }

  unsigned char* rec_mine = smem + ExtLds::edge + (size_t)((DIR * 4 + w) * 2) * kXHalo * 2 * ExtLds::kRec;
  const int upw = DIR == 0 ? w - 1 : w + 1;
  const bool has_up = active && (DIR == 0 ? w > 0 : w + 1 < W);
  unsigned char* rec_up = smem + ExtLds::edge + (size_t)((DIR * 4 + (has_up ? upw : 0)) * 2) * kXHalo * 2 * ExtLds::kRec;
  unsigned char* tiles = smem + ExtLds::tiles + (size_t)DIR * 2 * ExtLds::tile_bytes(V);      // this direction's two buffers
  const int tile_b = (int)ExtLds::tile_bytes(V);
  const int d_tid = w * 64 + lane;
  g_float* ckm = (g_float*)(const_cast<float*>(DIR == 0 ? rt.ckA : rt.ckQ) + (size_t)b * rt.NS * rt.CELLS);
  g_int* cke = (g_int*)((DIR == 0 ? rt.ckXA : rt.ckXQ) + (size_t)b * rt.NS * rt.CELLS);
  auto tile_of = [&](int q) { const int m = DIR == 0 ? q : M - q; return min(max(m, 0), M); };   // (clamped: past the end, the last one again)

  // tile 0 of the walk into buffer 0
  XTileFetch<DIR> fetch;
  if (stager) { fetch.load(rt, b, tile_of(0), Tmax, V, d_tid); fetch.store(rt, tiles, V, d_tid); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // STEADY: a tile strictly inside the walk -- all 8 rows live, none of them the chain's first
  auto do_tile = [&](int q, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    const int m = DIR == 0 ? q : M - q;
    if (q > 0) {
      // ---- the boundary: edge lanes out, barrier, halo in ----
      if (!(E2E_EXT_ABL & 4)) {
        if (active && edge) {
#pragma unroll
          for (int r = 0; r < NP; r++) {
            unsigned char* a = rec_mine + (size_t)(((q & 1) * kXHalo + (lane & (kXHalo - 1))) * 2 + r) * ExtLds::kRec;
            *reinterpret_cast<double2*>(a) = double2{Bm[r], Lm[r]};
            *reinterpret_cast<int2*>(a + 16) = int2{Be[r], Le[r]};
          }
        }
        // (a barrier that waits for LDS only: __syncthreads() would also drain the tile loads and the checkpoint stores)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (has_up && halo) {
#pragma unroll
          for (int r = 0; r < NP; r++) {
            const unsigned char* a = rec_up + (size_t)(((q & 1) * kXHalo + (lane & (kXHalo - 1))) * 2 + r) * ExtLds::kRec;
            const double2 v = *reinterpret_cast<const double2*>(a);
            const int2 e = *reinterpret_cast<const int2*>(a + 16);
            Bm[r] = v.x; Lm[r] = v.y; Be[r] = e.x; Le[r] = e.y;
          }
        }
      }
    }
    // ---- the tile after this one is asked for now and staged at this tile's end (into the other buffer: its readers left it
    //      before the barrier above): one tile -- 8 steps, ~2 us -- hides the round trip, and no loaded register lives across
    //      the loop's back edge (where the compiler waits for everything outstanding) ----
    if (!(E2E_EXT_ABL & 1) && stager) fetch.load(rt, b, tile_of(q + 1), Tmax, V, d_tid);
    if (!active) { if (!(E2E_EXT_ABL & 1) && stager) fetch.store(rt, tiles + ((q + 1) & 1) * tile_b, V, d_tid); return; }
    // ---- checkpoints: alpha row t = 16 k - 1 -> slot k, beta-with-emission row t = 16 k -> slot k (0 < 16 k < T), cells in lattice
    //      order (blank g = cell 2 g, label g = cell 2 g + 1), the mantissa as f32.  The row is the one the tile before ended
    //      with (freshly normalised; the halo refill above does not touch the lanes that store).  Stored HERE, behind the
    //      request for the next tile: vmcnt counts in order, and a store issued in front of that request would have to
    //      land before the tile can be staged (measured: 30 us per 1000 frames) ----
    if (!(E2E_EXT_ABL & 8) && q > 0) {
      const int tb_ = DIR == 0 ? 8 * m : 8 * m + 8;                   // alpha: the row 8 m - 1 -> slot m / 2; beta: the row 8 (m + 1)
      if ((tb_ & 15) == 0 && tb_ > 0 && tb_ < T) {
        const int slot = tb_ >> 4;
        if (!halo) {
#pragma unroll
          for (int r = 0; r < NP; r++) {
            const int g = g0 + r;
            if (g >= 0 && g <= S) {
              const size_t cb = (size_t)slot * rt.CELLS + 2 * g;
              ckm[cb] = (float)Bm[r]; cke[cb] = Be[r];
              if (DIR == 0) { ckm[cb + 1] = (float)Lm[r]; cke[cb + 1] = Le[r]; }
              else if (g >= 1) { ckm[cb - 1] = (float)Lm[r]; cke[cb - 1] = Le[r]; }
            }
          }
        }
      }
    }
    // ---- this tile's probabilities of the lane's columns ----
    float yb[kXTile], yl[NP][kXTile];
    {
      const unsigned char* tb = tiles + ((E2E_EXT_ABL & 1) ? 0 : (q & 1) * tile_b);
      const x_f4 u0 = *(const lds_f32x4*)(tb + blank * 32), u1 = *(const lds_f32x4*)(tb + blank * 32 + 16);
      yb[0] = u0.x; yb[1] = u0.y; yb[2] = u0.z; yb[3] = u0.w; yb[4] = u1.x; yb[5] = u1.y; yb[6] = u1.z; yb[7] = u1.w;
#pragma unroll
      for (int r = 0; r < NP; r++) {
        const x_f4 v0 = *(const lds_f32x4*)(tb + lab[r] * 32), v1 = *(const lds_f32x4*)(tb + lab[r] * 32 + 16);
        yl[r][0] = v0.x; yl[r][1] = v0.y; yl[r][2] = v0.z; yl[r][3] = v0.w; yl[r][4] = v1.x; yl[r][5] = v1.y; yl[r][6] = v1.z; yl[r][7] = v1.w;
      }
    }
#pragma unroll
    for (int k = 0; k < kXTile; k++) {
      const int tt = DIR == 0 ? k : kXTile - 1 - k;
      const int t = m * kXTile + tt;
      if (!STEADY && t >= T) continue;                              // (uniform: the utterance's last tile may be short)
      const double ybt = (double)yb[tt];
      if (!STEADY && (DIR == 0 ? t == 0 : t == T - 1)) {
        // the first row: ctc_loss.cpp:39-42 (alpha), :76-78 with the emission (beta)
#pragma unroll
        for (int r = 0; r < NP; r++) {
          const int g = g0 + r;
          const bool isB = DIR == 0 ? g == 0 : g == S, isL = DIR == 0 ? (g == 0 && S > 0) : (g == S && S > 0);
          const double vb = (isB && cond) ? ybt : 0.0, vl = isL ? (double)yl[r][tt] : 0.0;
          Bm[r] = vb; Be[r] = x_fix(vb, 0); Lm[r] = vl; Le[r] = x_fix(vl, 0);
        }
      } else if (E2E_EXT_ABL & 2) {
#pragma unroll
        for (int r = 0; r < NP; r++) { Bm[r] += ybt; Lm[r] += (double)yl[r][tt]; }
      } else if (DIR == 0) {
        double Pm = lane_shift_up(Lm[NP - 1]); int Pe = x_shift_up_e(Le[NP - 1]);
#pragma unroll
        for (int r = 0; r < NP; r++) {
          const double om = Lm[r]; const int oe = Le[r];
          x_pair_step(Bm[r], Be[r], Lm[r], Le[r], Pm, Pe, sk[r], ybt, lvalid[r] ? (double)yl[r][tt] : 0.0);
          Pm = om; Pe = oe;
        }
      } else {
        double Pm = lane_shift_down(Lm[0]); int Pe = x_shift_down_e(Le[0]);
#pragma unroll
        for (int r = NP - 1; r >= 0; r--) {
          const double om = Lm[r]; const int oe = Le[r];
          x_pair_step(Bm[r], Be[r], Lm[r], Le[r], Pm, Pe, sk[r], ybt, lvalid[r] ? (double)yl[r][tt] : 0.0);
          Pm = om; Pe = oe;
        }
      }
      if ((k & 3) == 3) {                                           // (includes the checkpoint rows: the last step of every second tile)
#pragma unroll
        for (int r = 0; r < NP; r++) { x_norm(Bm[r], Be[r]); x_norm(Lm[r], Le[r]); }
      }
    }
    if (!(E2E_EXT_ABL & 1)) fetch.store(rt, tiles + ((q + 1) & 1) * tile_b, V, d_tid);
  };
  do_tile(0, std::false_type{});
  for (int q = 1; q < M; q++) do_tile(q, std::true_type{});
  if (M > 0) do_tile(M, std::false_type{});
  // ---- Z from this side: alpha (ctc_loss.cpp:63-70): blank S + label S-1 of the last row; beta: sum_j alpha_0[j] beta_0[j]
  //      = [cond] blank 0 + label 0 of the row t = 0 (with their emissions) ----
  if (active && !halo) {
    double* z = reinterpret_cast<double*>(smem + ExtLds::zrec) + DIR * 4;
#pragma unroll
    for (int r = 0; r < NP; r++) {
      const int g = g0 + r;
      if (DIR == 0) {
        if (g == S) { z[0] = Bm[r]; z[1] = (double)Be[r]; }
        if (g == S - 1) { z[2] = Lm[r]; z[3] = (double)Le[r]; }
      } else {
        if (g == 0) { z[0] = cond ? Bm[r] : 0.0; z[1] = (double)(cond ? Be[r] : kXZero); }
        if (g == 1) { z[2] = Lm[r]; z[3] = (double)Le[r]; }
      }
    }
  }
}

// One utterance's chains on this workgroup.  Ends with the utterance's flag word carrying kExtDone (and kExtBad if the
// partition sum is not a positive number or the two sides disagree), after a release fence.
// (Forced inline, like everything that takes the parameter block by reference.  Out of line -- tried for the chains' sake: a
//  register allocation of their own -- the block's address escapes, the kernel keeps it on its stack, and EVERY wave of EVERY
//  launch copies the 336 bytes per lane to scratch at the kernel's entry, in front of the early exit: the empty flagged launch
//  9.6 instead of 5.3 us, the headline call +6.6 us.  profiles/r05_placement/; tests/test_host_cpu.py audits the built library
//  for it.  The dynamic LDS is found again through its own declaration.)
// half: -1 = both directions on this workgroup (waves (alpha w, beta w): every SIMD carries a wave of either direction); 0 / 1 = only
// the alpha / only the beta chains, the other direction on ANOTHER workgroup (round 6, when the call has so few flagged utterances
// that the launch has workgroups to spare -- 8 mislabelled utterances used 8 of 256 CUs): a chain wave then has its SIMD to itself.
// The two sides leave their cells of Z in the workspace and meet through two bits of the utterance's flag word; whichever comes
// second finishes the utterance.
template <typename IO>
__device__ __forceinline__ void ext_chains(const ExactParams& p, int b, int half = -1) {
  extern __shared__ __align__(16) unsigned char smem[];
  typedef typename LossOf<IO>::type LT;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = (int)p.x_len[b], S = (int)p.t_len[b];
  const int dir = half < 0 ? (wid & 1) : half, w = half < 0 ? (wid >> 1) : wid;
  if (tid < 8) {
    double* z = reinterpret_cast<double*>(smem + ExtLds::zrec);
    z[tid] = (tid & 1) ? (double)kXZero : 0.0;
  }
  __syncthreads();
#ifndef E2E_EXT_FORCE_NP2          // (tools/diag A/B: two pairs per lane whatever the width)
#define E2E_EXT_FORCE_NP2 0
#endif
  if (!E2E_EXT_FORCE_NP2 && S + 1 <= 4 * kXOwnLanes) {
    if (dir == 0) ext_chain_wave<0, 1, LT>(p, smem, b, T, S, w, lane); else ext_chain_wave<1, 1, LT>(p, smem, b, T, S, w, lane);
  } else {
    if (dir == 0) ext_chain_wave<0, 2, LT>(p, smem, b, T, S, w, lane); else ext_chain_wave<1, 2, LT>(p, smem, b, T, S, w, lane);
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    double* z = reinterpret_cast<double*>(smem + ExtLds::zrec);
    bool finish = true;
    if (half >= 0) {
      // this side's cells of Z into the workspace ([B][2] final values, then [B][2 sides][4]); then the ticket
      double* zh = p.retry.extz + 2 * (size_t)p.B + 8 * (size_t)b;
      for (int k = 0; k < 4; k++) zh[4 * half + k] = z[4 * half + k];
      __threadfence();
      const int old = atomicOr(&p.flags[b], half == 0 ? kExtHalfA : kExtHalfB);
      finish = (old & (half == 0 ? kExtHalfB : kExtHalfA)) != 0;
      if (finish) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int k = 0; k < 4; k++) z[4 * (half ^ 1) + k] = __hip_atomic_load(&zh[4 * (half ^ 1) + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (finish) {
    auto sum2 = [&](const double* c, double& m, int& e) {
      const int e0 = (int)c[1], e1 = (int)c[3];
      e = max(e0, e1);
      m = ldexp(c[0], e0 - e) + ldexp(c[2], e1 - e);
      if (m == 0.0) e = kXZero;
      x_norm(m, e);
    };
    double ma, mb; int ea, eb;
    sum2(z, ma, ea); sum2(z + 4, mb, eb);
    const double la = log(ma) + (double)ea * 0.693147180559945309417, lb = log(mb) + (double)eb * 0.693147180559945309417;
    const bool ok = ma > 0.0 && mb > 0.0 && fabs(la - lb) <= 1e-8 * fmax(1.0, fabs(la));
    if (ok) {
      p.retry.extz[2 * b] = ma; p.retry.extz[2 * b + 1] = (double)ea;
      reinterpret_cast<LT*>(p.losses)[b] = (LT)(-la);
    }
    __threadfence();
    atomicOr(&p.flags[b], ok ? kExtDone : (kExtDone | kExtBad));
    }
  }
  __syncthreads();
}

// ---- the segments ----------------------------------------------------------------------------------------------------
// One (utterance, 16-step segment) on this workgroup: wave c takes the label pairs 32 c .. 32 c + 31 (its lanes hold the
// pairs 32 c - 16 .. 32 c + 47: what the 16 steps can reach from either side), c = wid, wid + 8, ...
//
// Round 6.  What an item cost was not its lattice -- with the chunk loop compiled out the call took as long as with only its LDS
// atomics removed (sharp_unrelated: 1 325 against 1 504 us; tools/diag: -DE2E_EXT_ABL=128 / 64) -- but the dependent round trips to
// memory around it, on a CU that holds nothing else to run meanwhile: the wait for the utterance's flag and the acquire fence
// behind it, Z, the targets, the probability rows behind the targets, the checkpoints, the probabilities again for the gradient
// rows; ~15 us of a 17 us item.  So a workgroup now takes a CONTIGUOUS run of the list's items -- mostly segments of ONE utterance
// -- and keeps what belongs to the utterance (XSegUtt: lengths, Z, its waves' labels and skip flags) across them; an item asks
// for everything it reads from memory at its top, in one round trip; the posterior rows are cleared by the pass that reads them.
struct XSegChunk { int g, lab; bool own, lvalid; double skp, skn; };
struct XSegUtt {
  int b, T, S; double Zinv; int Ze;
  XSegChunk ch[2];                    // the wave's chunks c = wid, wid + 8 (targets of up to 447 labels: 14 chunks)
};
template <typename IO>
__device__ __forceinline__ void ext_segment_setup(const ExactParams& p, int b, XSegUtt& u) {
  const FastRetry& rt = p.retry;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int V = p.V, blank = p.blank;
  u.b = b; u.T = (int)p.x_len[b]; u.S = (int)p.t_len[b];
  u.Zinv = 1.0 / rt.extz[2 * b]; u.Ze = (int)rt.extz[2 * b + 1];
  const int64_t* tg = p.targets + (int64_t)b * p.tgt_stride;
  const int S = u.S;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int c = wid + 8 * k;
    XSegChunk& q = u.ch[k];
    q.g = 32 * c - 16 + lane;
    q.own = lane >= 16 && lane < 48 && q.g <= S;
    q.lvalid = q.g >= 0 && q.g < S;
    const int lv = q.lvalid ? (int)tg[q.g] : 0;
    q.lab = min(max(lv, 0), V - 1);
    const int lpv = q.g >= 1 && q.g < S ? (int)tg[q.g - 1] : -1, lnv = q.g >= 0 && q.g + 1 < S ? (int)tg[q.g + 1] : -1;
    q.skp = (q.lvalid && q.g >= 1 && lv != blank && lpv != lv) ? 1.0 : 0.0;           // ctc_loss.cpp:53-57
    q.skn = (q.lvalid && q.g + 1 < S && lv != blank && lnv != lv) ? 1.0 : 0.0;         // ctc_loss.cpp:91-96
  }
}

// `post` ([16][V + 1] doubles in LDS) is all zero on entry and on exit; the caller puts a barrier between two items.
template <typename IO>
__device__ __forceinline__ void ext_segment(const ExactParams& p, const XSegUtt& u, int seg) {
  extern __shared__ __align__(16) unsigned char smem[];
  const FastRetry& rt = p.retry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V, blank = p.blank, Tmax = p.T;
  const int b = u.b, T = u.T, S = u.S, L = 2 * S + 1;
  const int t0 = seg * kFastSeg, n = min(kFastSeg, T - t0);
  const bool cond = (T > 1 || L == 1);
  double* post = reinterpret_cast<double*>(smem);                   // [16][V + 1]
  const double Zinv = u.Zinv;
  const int Ze = u.Ze;
  // the probabilities of the gradient rows this thread will write (alphabets of up to 64 columns: two per thread), asked for now
  constexpr int kEarly = 2;
  const bool early = n * V <= kEarly * kThreads;
  float yv[kEarly];
  auto y_at = [&](int i) -> float {
    const int tt = i / V, v = i - tt * V;
    return rt.ytab_segments ? rt.ytab[(((size_t)b * rt.NS + seg) * V + v) * kFastSeg + tt] : rt.ytab[((size_t)b * Tmax + t0 + tt) * V + v];
  };
#pragma unroll
  for (int k = 0; k < kEarly; k++) yv[k] = (early && tid + k * kThreads < n * V) ? y_at(tid + k * kThreads) : 0.f;
  const int nchunk = (S + 32) / 32;                                 // pairs 0 .. S
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int c = wid + 8 * k;
    if (c >= ((E2E_EXT_ABL & 128) ? 0 : nchunk)) continue;
    const XSegChunk& q = u.ch[k];
    const int g = q.g, lab = q.lab;
    const bool own = q.own, lvalid = q.lvalid;
    const double skp = q.skp, skn = q.skn;
    float yb[16], yl[16];
    x_load_rows(rt, b, seg, Tmax, V, blank, yb);
    x_load_rows(rt, b, seg, Tmax, V, lab, yl);
    // both checkpoint rows, asked for before either is used
    double Bm = 0.0, Lm = 0.0; int Be = kXZero, Le = kXZero;
    double qBm = 0.0, qLm = 0.0; int qBe = kXZero, qLe = kXZero;
    {
      const bool in = g >= 0 && g <= S;
      const bool hasA = seg > 0 && in, hasQ = t0 + n < T && in;
      const size_t ca = ((size_t)b * rt.NS + seg) * rt.CELLS + 2 * (in ? g : 0), cq = ca + rt.CELLS;
      float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f; int ea0 = kXZero, ea1 = kXZero, eq0 = kXZero, eq1 = kXZero;
      if (hasA) { a0 = rt.ckA[ca]; a1 = rt.ckA[ca + 1]; ea0 = rt.ckXA[ca]; ea1 = rt.ckXA[ca + 1]; }
      if (hasQ) { q0 = rt.ckQ[cq]; q1 = rt.ckQ[cq + 1]; eq0 = rt.ckXQ[cq]; eq1 = rt.ckXQ[cq + 1]; }
      if (hasA) { Bm = (double)a0; Lm = (double)a1; Be = x_fix(Bm, ea0); Le = x_fix(Lm, ea1); }
      if (hasQ) {
        qBm = (double)q0; qLm = (double)q1; qBe = x_fix(qBm, eq0); qLe = x_fix(qLm, eq1);
        if (g == S) { qLm = 0.0; qLe = kXZero; }                    // (label S does not exist; its checkpoint cell is never written)
      }
    }
    // ---- alpha rows t0 .. t0 + n - 1 ----
    double aBm[16], aLm[16]; int aBe[16], aLe[16];
#pragma unroll
    for (int tt = 0; tt < 16; tt++) {
      if (tt < n) {
        const double ybt = (double)yb[tt], ylt = lvalid ? (double)yl[tt] : 0.0;
        if (t0 + tt == 0) {
          const double vb = (g == 0 && cond) ? ybt : 0.0, vl = (g == 0 && S > 0) ? ylt : 0.0;
          Bm = vb; Be = x_fix(vb, 0); Lm = vl; Le = x_fix(vl, 0);
        } else {
          const double Pm = lane_shift_up(Lm); const int Pe = x_shift_up_e(Le);
          x_pair_step(Bm, Be, Lm, Le, Pm, Pe, skp, ybt, ylt);
        }
        if ((tt & 3) == 3) { x_norm(Bm, Be); x_norm(Lm, Le); }
      }
      aBm[tt] = Bm; aBe[tt] = Be; aLm[tt] = Lm; aLe[tt] = Le;
    }
    // ---- beta back through the segment (q = beta with its emission; unshifted pairing: slot (blank g, label g)) ----
#pragma unroll
    for (int tt = 15; tt >= 0; tt--) {
      if (tt >= n) continue;
      const int t = t0 + tt;
      // beta of row t WITHOUT its emission (ctc_loss.cpp:72-100): what alpha_t is multiplied with
      double bBm, bLm; int bBe, bLe;
      if (t == T - 1) {
        bBm = (g == S && cond) ? 1.0 : 0.0; bBe = x_fix(bBm, 0);
        bLm = (g == S - 1 && g >= 0) ? 1.0 : 0.0; bLe = x_fix(bLm, 0);
      } else {
        const double nBm = lane_shift_down(qBm), nLm = lane_shift_down(qLm);
        const int nBe = x_shift_down_e(qBe), nLe = x_shift_down_e(qLe);
        bBe = max(qBe, qLe);
        bBm = ldexp(qBm, qBe - bBe) + ldexp(qLm, qLe - bBe);         // (zero exactly when both terms are: then bBe is kXZero already)
        const int nLeS = skn != 0.0 ? nLe : kXZero;
        bLe = max(qLe, max(nBe, nLeS));
        bLm = fma(skn, ldexp(nLm, nLeS - bLe), ldexp(qLm, qLe - bLe) + ldexp(nBm, nBe - bLe));
      }
      // posteriors of the lane's two cells (ordinary doubles: a posterior below 2^-1074 IS zero)
      const double pB = ldexp(aBm[tt] * bBm * Zinv, aBe[tt] + bBe - Ze);
      const double pL = ldexp(aLm[tt] * bLm * Zinv, aLe[tt] + bLe - Ze);
      double pb = own ? pB : 0.0;
      if (!(E2E_EXT_ABL & 32)) pb = wave_sum_lane63(pb);
      if (!(E2E_EXT_ABL & 64)) {
        if (lane == 63 && pb != 0.0) atomicAdd(&post[tt * (V + 1) + blank], pb);
        if (own && lvalid && pL != 0.0) atomicAdd(&post[tt * (V + 1) + lab], pL);
      }
      // q of row t
      const double ybt = (double)yb[tt], ylt = lvalid ? (double)yl[tt] : 0.0;
      qBm = bBm * ybt; qBe = ybt != 0.0 ? bBe : kXZero;
      qLm = bLm * ylt; qLe = ylt != 0.0 ? bLe : kXZero;
      if ((tt & 3) == 0) { x_norm(qBm, qBe); x_norm(qLm, qLe); }
    }
  }
  __syncthreads();
  // ---- the rows: y - posterior (ctc_loss.cpp:102-117); what is read of `post` is cleared for the next item ----
  IO* grads = reinterpret_cast<IO*>(p.grads) + (size_t)b * (size_t)Tmax * (size_t)V;
  if (early) {
#pragma unroll
    for (int k = 0; k < kEarly; k++) {
      const int i = tid + k * kThreads;
      if (i < n * V) {
        const int tt = i / V, v = i - tt * V;
        const double po = post[tt * (V + 1) + v];
        post[tt * (V + 1) + v] = 0.0;
        grads[(size_t)(t0 + tt) * V + v] = (IO)(((double)yv[k] - po) * p.gscale);
      }
    }
  } else {
    for (int i = tid; i < n * V; i += kThreads) {
      const int tt = i / V, v = i - tt * V;
      const double po = post[tt * (V + 1) + v];
      post[tt * (V + 1) + v] = 0.0;
      grads[(size_t)(t0 + tt) * V + v] = (IO)(((double)y_at(i) - po) * p.gscale);
    }
  }
}

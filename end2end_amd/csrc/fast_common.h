// Device code shared by the translation units of the fast CTC path (ctc_loss_fast.hip: single-wave and two-pairs-per-lane
// chains, segment kernel, launchers; ctc_loss_fast_h1.hip: the one-pair-per-lane chain kernel).  The kernels live in separate
// translation units -- i.e. separate code objects -- on purpose: see the note on code placement in DESIGN.md 4.1.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "common.h"

// the scaled lattice is not bit-pinned to the reference: let the compiler fuse multiply-adds here
#pragma clang fp contract(fast)

// Every kernel of the fast path starts on a 64 KB boundary of its code object: where a kernel's code lies relative to such
// boundaries changes its speed by integer factors (the segment kernel: 61 -> 401 us when another kernel's growth moved it
// across one), and with the alignment pinned no kernel's placement depends on its neighbours' sizes.
#ifndef E2E_KERNEL_ALIGN
#define E2E_KERNEL_ALIGN __attribute__((aligned(65536)))
#endif

namespace e2e {
namespace fastk {

constexpr int kSeg = kFastSeg;  // steps per F2 segment == checkpoint spacing (16)
constexpr int kBlk = 8;         // prep -> chain hand-off block and rescale period
constexpr int kRingBlks = 8;    // ring depth (blocks)
constexpr int kRow32 = 12;      // floats per label row of an f32 ring block (ctc_fast_chain_hf_kernel): 8 steps + pad, 48 B
constexpr int kRow = 10;        // doubles per label row of a ring block: 8 steps + pad (80 B spreads the 16-byte gathers over the banks)
constexpr int kMaxSmallV = 96;  // alphabet columns the lattice kernels take (prep: <= 12 columns per lane)
constexpr int kMaxBigV = 224;   // ... and with the wide-row forms of the halo chains and of the segment kernel (ChainF64W: f32 ring of
                                // depth 4, 28 columns per producer lane, four label sets per gradient lane); targets of <= 223 labels
constexpr int kMaxHugeV = 448;  // ... and of the long-transcript kernels (ChainF64LW: ring of depth 2, 56 columns per producer lane,
                                // seven label sets per gradient lane, eight pairs per segment-kernel lane); targets of <= 447 labels
__host__ __device__ inline int lstart_ints(int V) { return 64 * ((V + 64) >> 6) + 2; }   // label-start entries per utterance (>= 130)

struct FastParams {
  const void* x; int xdt;          // logits / log-probabilities and their dtype (E2E_F32 / E2E_F16 / E2E_BF16); the gradient has the same
  int64_t sB, sT, sV; int logprobs;
  const int64_t* targets; int64_t tgt_stride;
  const int64_t* x_len; const int64_t* t_len;
  int B, T, V, Smax, blank;
  float* losses; void* grads;     // (losses: f32 also for 16-bit I/O)
  float* ytab;     // probabilities y_t[v].  Alphabets of <= kMaxSmallV columns: [B][NS][V][16] -- per 16-step segment, label-major,
                   // the form the segment kernel's LDS tile has (one 16-byte copy per label and four steps: staging the
                   // row-major form cost it a seventh of its instructions); beyond (ChainF64W): [B][T][V]
  float* ckA;      // [B][NS][CELLS]  row k: alpha row at t = 16k-1 (k >= 1)
  float* ckQ;      // [B][NS][CELLS]  row k: beta-with-emission row at t = 16k (k >= 1)
  short* ckE;      // [B][NS][2][64]  per-lane exponent of checkpoint row k (0: alpha, 1: beta); -30000 = all zero
  int* cumA;       // [B][NB]   cumA[m]: sum of the exponents the alpha chain removed at steps 8i+7, i < m (cumA[0] = 0)
  int* cumB;       // [B][NB]   cumB[m]: sum of the exponents the beta chain removed at steps 8i, i >= m (0 past the end)
  int* trkA;       // [B][NB]   like cumA / cumB, but of a frame that follows the row's maximum block by block: the segment
  int* trkB;       //           kernel takes the exponents it replays INSIDE a segment from these differences (the multi-wave
                   //           chains' own frame lags by kMwLag blocks; the single-wave chains' frame is such a frame itself)
  double* zt2;     // [B]       log2 of the TILTED partition sum in the alpha chain's final units + what it removed:
                   //           what sum_j alpha_t[j]*beta_t[j] * 2^(cumA + cumB) must equal at every t
  double* logz;    // [B][2]    alpha-side / beta-side log Z
  int* flags;      // [B]       != 0: redo with the exact kernel
  unsigned* segmask; // [B][MW]  bit s of an utterance's words: segment s failed its range / self-check (flag bits 8 / 16): the
  int MW;          //           f64 redo of the flagged-utterance launch takes those segments only; cleared by the chain kernel
  unsigned* cinfo; // [B][CELLS/2]  per label pair: label (10 bits) | sorted slot << 10 (10 bits) | alpha skip << 20 | beta skip << 21
  int* lstart;     // [B][LS]   first label-sorted slot of every label (V+1 entries used); LS = 64 * ceil((V+1)/64) + 2
  int LS;
  int* ctl;        // [4]  0: fallback workgroups that have finished (F1 clears it; the last one reduces the losses);
                   //      1: flagged utterances the f64 redo of the segments could not settle (diagnostics)
  float gscale;    // every gradient element is multiplied by this as it is written (e2e_ctc_loss_opts.grad_scale)
  float ztol;      // |log2| tolerance of the segment kernel's self-check (kZTol with f64 chains, kZTolF32 with f32 chains)
  int chains;      // host side: E2E_CHAINS_* of the call
  int NS, NB, CELLS;
};

// ---- cross-lane helpers (wave64) ------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);
}
// lane n <- lane n-1 (lane 0 keeps 0)
// (bound_ctrl: the lane without a source reads 0 and the compiler need not materialise an `old` operand)
template <int CTRL>
__device__ __forceinline__ int dpp_z(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ double from_prev_lane(double v) {
  const int lo = dpp_z<0x138>(__double2loint(v)), hi = dpp_z<0x138>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float from_prev_lane(float v) {
  return __int_as_float(dpp_z<0x138>(__float_as_int(v)));
}
// lane n <- lane n+1 (lane 63 keeps 0)
__device__ __forceinline__ double from_next_lane(double v) {
  const int lo = dpp_z<0x130>(__double2loint(v)), hi = dpp_z<0x130>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float from_next_lane(float v) {
  return __int_as_float(dpp_z<0x130>(__float_as_int(v)));
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
// The two cross-row steps of a wave-wide DPP scan / reduction.  Written as fused DPP instructions by hand: lanes of
// the rows that are masked off keep their value (dst is also the second source), which the update_dpp builtin can
// only express with an extra zeroed register and a separate add.  (s_nop 1: VALU write -> DPP read hazard.)
#define E2E_ROW_BCAST_STEPS(OP, v)                                                              \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
               "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v))

// maximum over the wave, all VALU (no ds_bpermute round trips); the result is wave-uniform
__device__ __forceinline__ int wave_max(int v) {
  v = max(v, dpp_i<0xB1>(0, v));
  v = max(v, dpp_i<0x4E>(0, v));
  v = max(v, dpp_i<0x141>(0, v));
  v = max(v, dpp_i<0x140>(0, v));                                                   // every lane: its row's maximum
  E2E_ROW_BCAST_STEPS("v_max_i32_dpp", v);                                          // lane 63: the wave's
  return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum over the 64 lanes, all DPP
__device__ __forceinline__ float wave_scan(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));   // row_shr:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));   // row_shr:8
  E2E_ROW_BCAST_STEPS("v_add_f32_dpp", v);
  return v;
}
// eight inclusive prefix sums at once, step by step across the eight: a DPP instruction needs two wait states after
// the VALU write of its source, which the other seven values' instructions fill (scanning them one after the other
// costs an s_nop per step -- a fifth of the instructions of the F2 scan phase)
__device__ __forceinline__ void wave_scan8(float (&v)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x111, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x112, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x114, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x118, 0xf, 0xf, true));
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
__device__ __forceinline__ int wave_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  E2E_ROW_BCAST_STEPS("v_add_u32_dpp", v);
  return v;
}
// inclusive prefix maximum over the 64 lanes, all DPP (lanes without a source keep their own value)
__device__ __forceinline__ int wave_scan_max(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  E2E_ROW_BCAST_STEPS("v_max_i32_dpp", v);
  return v;
}

// Wave-wide sums of 8 per-lane values at once (result: wave-uniform).  Instead of 8 reductions of 6 steps, the
// first two butterfly steps HALVE the number of values a lane carries (each lane pair / quad shares the rows between
// its members), so the whole thing is ~30 operations plus 8 readlanes.
__device__ __forceinline__ void wave_sum8(const float (&v)[8], float (&tot)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float w[4], u[2];
#pragma unroll
  for (int j = 0; j < 4; j++) {            // partner lane^1: even lanes keep rows 0..3, odd lanes rows 4..7
    const float keep = b0 ? v[j + 4] : v[j], send = b0 ? v[j] : v[j + 4];
    w[j] = keep + __int_as_float(dpp_i<0xB1>(0, __float_as_int(send)));
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {            // partner lane^2: row = 4*b0 + 2*b1 + j
    const float keep = b1 ? w[j + 2] : w[j], send = b1 ? w[j] : w[j + 2];
    u[j] = keep + __int_as_float(dpp_i<0x4E>(0, __float_as_int(send)));
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {            // the 4 quads of a 16-lane row: rotations by 4 and 8 keep (b1, b0)
    u[j] += __int_as_float(dpp_i<0x124>(0, __float_as_int(u[j])));     // row_ror:4
    u[j] += __int_as_float(dpp_i<0x128>(0, __float_as_int(u[j])));     // row_ror:8
  }
#pragma unroll
  for (int j = 0; j < 2; j++) u[j] += __shfl_xor(u[j], 16, 64);        // the 4 rows (row_bcast would mix the classes)
#pragma unroll
  for (int j = 0; j < 2; j++) u[j] += __shfl_xor(u[j], 32, 64);
#pragma unroll
  for (int c = 0; c < 4; c++) {            // lane c holds the rows of class c = (b1, b0)
    const int row = 4 * (c & 1) + 2 * (c >> 1);
    tot[row] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u[0]), c));
    tot[row + 1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u[1]), c));
  }
}

// all-reduce inside each 16-lane DPP row (pure VALU, no LDS crossbar)
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __int_as_float(dpp_i<0xB1>(0, __float_as_int(v))));
  v = fmaxf(v, __int_as_float(dpp_i<0x4E>(0, __float_as_int(v))));
  v = fmaxf(v, __int_as_float(dpp_i<0x141>(0, __float_as_int(v))));
  v = fmaxf(v, __int_as_float(dpp_i<0x140>(0, __float_as_int(v))));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(dpp_i<0xB1>(0, __float_as_int(v)));
  v += __int_as_float(dpp_i<0x4E>(0, __float_as_int(v)));
  v += __int_as_float(dpp_i<0x141>(0, __float_as_int(v)));
  v += __int_as_float(dpp_i<0x140>(0, __float_as_int(v)));
  return v;
}

// all-reduce inside each group of 8 lanes
__device__ __forceinline__ float row8_max(float v) {
  v = fmaxf(v, __int_as_float(dpp_i<0xB1>(0, __float_as_int(v))));
  v = fmaxf(v, __int_as_float(dpp_i<0x4E>(0, __float_as_int(v))));
  v = fmaxf(v, __int_as_float(dpp_i<0x141>(0, __float_as_int(v))));     // row_half_mirror: lane i <-> 7-i
  return v;
}
__device__ __forceinline__ float row8_sum(float v) {
  v += __int_as_float(dpp_i<0xB1>(0, __float_as_int(v)));
  v += __int_as_float(dpp_i<0x4E>(0, __float_as_int(v)));
  v += __int_as_float(dpp_i<0x141>(0, __float_as_int(v)));
  return v;
}

// Hand-off words live in LDS and guard LDS data only.  The LDS executes one wave's operations in order, so the
// producer needs no wait between its data writes and the flag write, and the consumer only has to keep the
// compiler from hoisting its data reads above the flag read.  (A workgroup-scope release fence would also drain
// the wave's outstanding GLOBAL stores -- checkpoints, probability rows -- once per 8-step block.)
#ifdef E2E_FAST_PROFILE
// (diagnostic builds only; every translation unit that includes this header has its own copy)
static __device__ unsigned long long g_prof[256 * 4 * 4];    // [wg][wave][total, spin, nspin, -]
static __device__ unsigned long long g_prof2[16384 * 8];     // F2 phase cycles per workgroup (first 16384)
static __device__ unsigned long long g_prof3[256 * 16 * 4];  // halo chains: [wg][dir*8 + wave][total, probability-ring wait, neighbour wait, frame wait]
static __device__ float g_zdev[16384];                        // F2 self-check: log2 deviation per workgroup
static __device__ unsigned long long g_tl[256 * 2 * 12];     // lean halo chains, first wave of a direction: [wg][dir][s_memtime, 100 MHz clock] x (kernel entry,
                                                              //  behind the entry barrier, first block's probabilities there, last step done, exit)
__shared__ unsigned long long s_prof_prev;
__shared__ unsigned long long s_prof_acc[8];
#define F2_STAMP(i) { unsigned long long _t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); \
    if (threadIdx.x == 0) { if ((i) < 0) { for (int _k = 0; _k < 8; _k++) s_prof_acc[_k] = 0; } else s_prof_acc[(i) < 0 ? 0 : (i)] += _t - s_prof_prev; } \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); if (threadIdx.x == 0) s_prof_prev = _t; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define F2_FLUSH { const unsigned _wg = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x < 8 && _wg < 16384) g_prof2[_wg * 8 + threadIdx.x] = s_prof_acc[threadIdx.x]; }
#define PROF_SPIN_BEGIN unsigned long long _t0 = __builtin_amdgcn_s_memtime();
#define PROF_SPIN_END(acc) acc += __builtin_amdgcn_s_memtime() - _t0;
#else
#define PROF_SPIN_BEGIN
#define PROF_SPIN_END(acc)
#define F2_STAMP(i)
#define F2_FLUSH
#endif
__host__ __device__ inline size_t align_up_dev(size_t x, size_t a) { return (x + a - 1) / a * a; }
typedef __attribute__((address_space(3))) int lds_int;
// (the flags must be addressed as LDS: through a generic pointer the poll becomes a flat load with sc0 sc1 and
// an s_waitcnt vmcnt(0) that again drains the global stores)
__device__ __forceinline__ void spin_until(volatile int* p, int want) {
  volatile lds_int* q = (volatile lds_int*)p;
  while (*q != want) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void spin_until_ge(volatile int* p, int want) {
  volatile lds_int* q = (volatile lds_int*)p;
  while (*q < want) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void publish(volatile int* p, int v) {
  asm volatile("" ::: "memory");          // (compiler ordering only: the LDS itself runs a wave's operations in order)
  *(volatile lds_int*)p = v;
}
__device__ __forceinline__ int peek(volatile int* p) { return *(volatile lds_int*)p; }
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef double h_d2 __attribute__((ext_vector_type(2)));
typedef int h_i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) h_d2 lds_d2;      // (every LDS access of these waves is a DS operation: a wave's DS
typedef __attribute__((address_space(3))) double lds_f64;   //  operations execute in order, which the hand-off words rely on)
typedef __attribute__((address_space(3))) h_i4 lds_i4;

// smallest / largest of the eight words at a 32-byte aligned LDS address (wave-uniform)
__device__ __forceinline__ int lds_min8(lds_u8* a) {
  const h_i4 u = *(volatile lds_i4*)a, v = *(volatile lds_i4*)(a + 16);
  return min(min(min(u.x, u.y), min(u.z, u.w)), min(min(v.x, v.y), min(v.z, v.w)));
}
__device__ __forceinline__ int lds_max8(lds_u8* a) {
  const h_i4 u = *(volatile lds_i4*)a, v = *(volatile lds_i4*)(a + 16);
  return max(max(max(u.x, u.y), max(u.z, u.w)), max(max(v.x, v.y), max(v.z, v.w)));
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
  __device__ void set_tilt(int S, int T) {
    // the number of alignments of t frames to i labels grows by ~((t-i)/(2i))^2 per extra label, so the
    // untilted maximum sits at i = t/3; r = 2*rho/(1-rho) moves it to i = rho*t
    r = fast_tilt(S, T);
  }
  // the packed form F1 leaves in the workspace for F2 (one word per label pair; `slot` = label-sorted position)
  __device__ static unsigned pack(int lab, int slot, float skp, float skn) {
    return (unsigned)lab | ((unsigned)slot << 10) | (skp != 0.f ? 1u << 20 : 0u) | (skn != 0.f ? 1u << 21 : 0u);
  }
  __device__ void unpack(const unsigned* w, int S, int T, int (&slot)[PPL]) {
    set_tilt(S, T);
    has_blank_label = false;
#pragma unroll
    for (int q = 0; q < PPL; q++) {
      lab[q] = (int)(w[q] & 0x3ffu);
      slot[q] = (int)((w[q] >> 10) & 0x3ffu);
      skp[q] = (w[q] >> 20) & 1u ? r * r : 0.f;
      skn[q] = (w[q] >> 21) & 1u ? r * r : 0.f;
    }
  }
  __device__ void load(const int64_t* tg, int S, int T, int V, int blank, int lane) {
    has_blank_label = false;
    set_tilt(S, T);
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
// F1 device code
// ============================================================================================
// LDS of F1.  Per direction a ring of kRingBlks blocks; a block holds the probabilities of its 8 steps TRANSPOSED:
// [label v][step] doubles, plus an all-zero row V for lattice cells past the utterance's labels.
// What costs on this machine is the number of LDS instructions (a wave pays >= 12 cycles for each, whatever its
// width), so the per-lane gather is arranged to pull 4 consecutive time steps of the lane's label per instruction:
// 2 reads per label cell and block instead of 8, and the producers need no gather at all.
struct F1Lds {
  double* ring;      // [2][kRingBlks][V+1][kRow]  (f64: saves the chains' conversions, 2.5 % of the step at B = 256.  It
                     //  costs them 30 VGPRs, though: at 146 a second workgroup does not fit on the CU, which an f32 ring
                     //  (114) allows -- measured +7 % at B = 1024 with the second workgroup's roles rotated onto SIMDs 1/3)
  int* filled;       // [2][kRingBlks]   probability block n of a direction is complete (== n+1)
  int* took;         // [2]              the direction's chain has read the probabilities of blocks < took
  int* sortcnt;      // [130] counting-sort scratch of the cell-info wave (V <= 96 here: two chunks of 64 labels + 2)
  int blk_elems;
  static constexpr int kSyncInts = 2 * kRingBlks + 2;
  // rb: the ring's depth in blocks -- kRingBlks, or 4 where that lets a second workgroup onto the CU (ctc_fast_chain_kernel's RB)
  __device__ F1Lds(unsigned char* smem, int V, int rb = kRingBlks) {
    blk_elems = (V + 1) * kRow;
    ring = reinterpret_cast<double*>(smem);
    filled = reinterpret_cast<int*>(ring + 2 * rb * blk_elems);
    took = filled + 2 * kRingBlks;
    sortcnt = took + 2;
  }
  __host__ __device__ static size_t bytes(int V, int rb = kRingBlks) { return sizeof(double) * 2 * rb * (V + 1) * kRow + sizeof(int) * (kSyncInts + 130); }
};

// Block geometry shared by prep and chain.  Both directions work in blocks of 8 steps that are ALIGNED in
// absolute time (t = 8m .. 8m+7), so that the rescale phase of a step is its position in the block:
//   alpha: block n covers t = 8n + tt;              beta: block n covers t = 8(M-n) + 7 - tt,  M = (T-1)/8
// (beta's first block may start with rows t >= T, which are dead).
__device__ __forceinline__ int block_time(int dir, int n, int tt, int T) {
  return dir == 0 ? n * kBlk + tt : (((T - 1) >> 3) - n) * kBlk + 7 - tt;
}

// exp(x) for x <= ~0 (softmax numerators, log-probabilities): two-constant range reduction + v_exp_f32 + ldexp,
// ~1 ulp like expf but without its overflow / underflow selects (ldexp saturates to 0 by itself).
__device__ __forceinline__ float exp_le0(float x) {
  x = fmaxf(x, -200.f);                                    // -inf (padding, log 0) -> exactly 0 instead of NaN
  const float t = x * 1.44269504088896340736f;
  const float n = rintf(t);
  float f = fmaf(x, 1.44269504088896340736f, -n);          // exact product residual
  f = fmaf(x, 1.92596299112661746e-8f, f);                 // log2(e) low part
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// Probability rows for one chain: a block is ONE pass -- each group of 8 lanes takes one of the block's 8 time steps,
// a lane holds the columns v = l8 + 8k, k < NV = ceil(V/8); max / sum by DPP all-reduce inside the group of 8.
// (Four steps per pass with 16 lanes each was the first form: the reductions, the reciprocal and the address
// arithmetic are paid per pass, and at V = 29 that was ~300 VALU instructions per block against ~110 here.  The
// producers share their SIMDs with the chain waves, so their instruction count is the chains' speed too.)
// MODE 0: f64 ring read by one chain wave (`took`); 1 / 2: f64 / f32 ring of ctc_fast_chain_hf_kernel -- label rows and one
// row of (blank probability, tilted blank probability) pairs, read by several waves whose progress words replace `took`.
template <int NV, int MODE = 0, int RB = kRingBlks>
__device__ __forceinline__ void prep_wave(const FastParams& p, int b, int T, int dir, int first, int stride,
                                          unsigned char* myring_bytes, int blk_bytes, volatile int* myfilled, volatile int* took,
                                          int lane, lds_u8* prog = nullptr, double rr2 = 0.0) {
  constexpr bool HALO = MODE != 0;
  const int V = p.V;
  const int nblk = (T + kBlk - 1) / kBlk;
  const int64_t xo = (int64_t)b * p.sB;
  float* ytab = p.ytab + (size_t)b * p.NS * kSeg * V;       // [segment][label][16 steps]
  const int tt = lane >> 3, l8 = lane & 7;
  const float ninf = -__builtin_huge_valf();
  bool col_live[NV];
  int64_t col_off[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) { col_live[k] = l8 + 8 * k < V; col_off[k] = (int64_t)(col_live[k] ? l8 + 8 * k : 0) * p.sV; }
  unsigned long long prof_spin = 0, prof_t0 = __builtin_amdgcn_s_memtime();
  (void)prof_spin; (void)prof_t0;
  // The logits of a block are requested two of this wave's blocks before they are worked on: an HBM miss under load
  // (2-4 thousand cycles) would otherwise sit in front of every block, and a chain takes ~1 000 cycles per block.  The
  // loads are unconditional (clamped addresses; what a dead row or column reads is replaced when it is used): a
  // conditional load becomes a branch per element and a full wait behind it.  Three register sets rotate through a loop
  // unrolled three times, so that no set is ever copied (a copy waits for the load it copies).
  // (F32IN: the logits are f32 -- the loop below exists twice, so that no test of the dtype sits between the loads; 16-bit
  //  logits take the copy whose loads choose between the two 16-bit types)
  auto load_block = [&](auto f32_tag, int n, float (&out)[NV]) {
    constexpr bool F32IN = decltype(f32_tag)::value;
    const int t = block_time(dir, n, tt, T);
    const bool row_live = n < nblk && t < T;
    const int64_t xr = xo + (int64_t)(row_live ? t : 0) * p.sT;
    if (F32IN) {
#pragma unroll
      for (int k = 0; k < NV; k++) out[k] = reinterpret_cast<const float*>(p.x)[xr + col_off[k]];
    } else {
      unsigned short h[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) h[k] = reinterpret_cast<const unsigned short*>(p.x)[xr + col_off[k]];
      const bool bf = p.xdt == E2E_BF16;
#pragma unroll
      for (int k = 0; k < NV; k++)
        out[k] = bf ? __uint_as_float((unsigned)h[k] << 16) : (float)__builtin_bit_cast(f16_t, h[k]);
    }
  };
  int consumed = 0;                 // blocks the ring's readers are known to have finished with (HALO)
  float lpmin = 0.f;                // smallest FINITE log-probability this wave has seen (alpha-side producers)
  auto process = [&](int n, const float (&xraw)[NV]) {
    if (n >= nblk) return;
    float xv[NV];
    {
      const bool row_in = block_time(dir, n, tt, T) < T;
#pragma unroll
      for (int k = 0; k < NV; k++) xv[k] = (row_in && col_live[k]) ? xraw[k] : ninf;
    }
    const int slot = n % RB;
    if (n >= RB) {
      PROF_SPIN_BEGIN
      if (HALO) {
        // (the readers' progress is looked at again only when the last look does not cover this block)
        while (consumed < n - RB + 1) {
          consumed = __builtin_amdgcn_readfirstlane(lds_min8(prog));
          if (consumed < n - RB + 1) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
      } else spin_until_ge(took, n - RB + 1);
      PROF_SPIN_END(prof_spin)
    }
    double* blk = reinterpret_cast<double*>(myring_bytes + (size_t)slot * blk_bytes);
    float* blk32 = reinterpret_cast<float*>(myring_bytes + (size_t)slot * blk_bytes);
    const int t = block_time(dir, n, tt, T);
    const bool row_live = t < T;
    float y[NV];
    if (p.logprobs) {
#pragma unroll
      for (int k = 0; k < NV; k++) {
        y[k] = exp_le0(xv[k]);
        lpmin = fminf(lpmin, xv[k] > ninf ? xv[k] : 0.f);      // (-inf: an impossible symbol, exact; so are dead rows)
      }
    } else {
      float m = xv[0];
#pragma unroll
      for (int k = 1; k < NV; k++) m = fmaxf(m, xv[k]);
      m = row8_max(m);
      float ssum = 0.f;
#pragma unroll
      for (int k = 0; k < NV; k++) {
        y[k] = exp_le0(xv[k] - m); ssum += y[k];
        lpmin = fminf(lpmin, xv[k] > ninf ? xv[k] - m : 0.f);  // (>= the log-probability)
      }
      ssum = row8_sum(ssum);
      float inv = __builtin_amdgcn_rcpf(ssum);
      inv = fmaf(fmaf(-ssum, inv, 1.0f), inv, inv);        // one Newton step: ~0.5 ulp
#pragma unroll
      for (int k = 0; k < NV; k++) y[k] *= inv;
    }
    float* yrow = ytab + ((size_t)((row_live ? t : 0) >> 4) * V + l8) * kSeg + (t & (kSeg - 1));
#pragma unroll
    for (int k = 0; k < NV; k++) {
      if (col_live[k]) {
        if (MODE == 2) {
          blk32[(l8 + 8 * k) * kRow32 + tt] = row_live ? y[k] : 0.f;       // transposed: [label][step]
          if (l8 + 8 * k == p.blank) {
            float2 yw; yw.x = row_live ? y[k] : 0.f; yw.y = row_live ? (float)rr2 * y[k] : 0.f;
            *reinterpret_cast<float2*>(blk32 + (V + 1) * kRow32 + 2 * tt) = yw;
          }
        } else if (MODE == 1) {
          blk[(l8 + 8 * k) * kRow + tt] = row_live ? (double)y[k] : 0.0;
          if (l8 + 8 * k == p.blank) {
            double2 yw; yw.x = row_live ? (double)y[k] : 0.0; yw.y = row_live ? rr2 * (double)y[k] : 0.0;
            *reinterpret_cast<double2*>(blk + (V + 1) * kRow + 2 * tt) = yw;
          }
        } else {
          blk[(l8 + 8 * k) * kRow + tt] = row_live ? (double)y[k] : 0.0;   // transposed: [label][step]
        }
        if (dir == 0 && row_live) yrow[8 * k * kSeg] = y[k];
      }
    }
    // every lane stores the same word: no divergence, one LDS write.  (MODE 1, 2: one word per producer, "my blocks up to n
    // are there" -- the readers keep the minimum of the two in a scalar and look again only when they catch up.)
    if (MODE != 0) publish(&myfilled[first], n + stride);
    else publish(&myfilled[slot], n + 1);
  };
  auto run = [&](auto f32_tag) {
    float xa[NV], xb[NV], xc[NV];
    load_block(f32_tag, first, xa);
    load_block(f32_tag, first + stride, xb);
    for (int n = first; n < nblk; n += 3 * stride) {       // this wave fills every `stride`-th block
      load_block(f32_tag, n + 2 * stride, xc); process(n, xa);
      load_block(f32_tag, n + 3 * stride, xa); process(n + stride, xb);
      load_block(f32_tag, n + 4 * stride, xb); process(n + 2 * stride, xc);
    }
  };
  if (p.xdt == E2E_F32) run(std::true_type{}); else run(std::false_type{});
  // Probabilities are f32: below ~2^-126 they are flushed, and a chain that ran through such frames carries a loss that
  // is off by the flushed amount (a symbol with log-probability -inf is exactly impossible and does not count).  Reason bit 64 ("emissions near the end of f32"): such an utterance is recomputed
  // entirely by the exact kernel, never by the f64 redo of the segments alone, which would keep the chains' loss.
  // (bit 256 on top: a finite log-probability below -78 -- within a few bits of where an f32 probability stops being a normal
  //  number -- , which the extended-range redo cannot take from the table either: the exact kernel's own softmax)
  { const bool t1 = __any(lpmin < -69.f), t2 = __any(lpmin < -78.f);      // (both votes by the whole wave, outside the lane test)
    if (dir == 0 && t1 && lane == 0) atomicOr(&p.flags[b], t2 ? 64 | 256 : 64); }       // e^-69 = 2^-100
#ifdef E2E_FAST_PROFILE
  if (lane == 0 && b < 256 && first == 0) { g_prof[(b * 4 + 2 + dir) * 4 + 0] = __builtin_amdgcn_s_memtime() - prof_t0; g_prof[(b * 4 + 2 + dir) * 4 + 1] = prof_spin; }
#endif
}

// Hand-off waits are bounded: a protocol error flags the utterance (bit 128 -> the exact kernel redoes it) instead of
// hanging the GPU.  (~2^20 polls of >= 64 cycles: far beyond any legitimate wait.)
#define HALO_WAIT(cond)                                                                          \
  do {                                                                                           \
    if (!(cond)) {                                                                               \
      int _spins = 0;                                                                            \
      do { __builtin_amdgcn_s_sleep(1); if (++_spins > (1 << 20)) { atomicOr(&p.flags[b], 128); break; } } while (!(cond)); \
    }                                                                                            \
    asm volatile("" ::: "memory");                                                               \
  } while (0)

// The halo chains' producers for alphabets of 97..224 columns (ChainF64W): prep_wave's MODE 2 layout -- an f32 ring of label rows
// and one row of (blank probability, tilted blank probability) pairs --, a ring of RING blocks (the rows are 2.3 times as long),
// and NO arithmetic: with 28 columns per lane a block cost a producer ~700 instructions, the exponentials were computed twice
// (alpha side, beta side) and the producers, not the chains, set the kernel's speed (510 cycles per step against 150).  The
// probabilities are therefore computed ONCE, by the whole chip, in a launch of their own (ctc_fast_prob_kernel, which fills
// ytab); a producer only moves a block's rows from ytab into the ring, transposed.
constexpr float kTinyProb = 1e-37f;       // ytab marker: a FINITE log-probability below -78 (e^-78 / 448 = 3e-37; the table keeps the true
                                          // probability down to there: what lies below 1e-30 = e^-69 raises reason bit 64, the marker 256 as well)
template <int NV, int RING, int SETS = 2>
__device__ __forceinline__ void prep_wave_big(const FastParams& p, int b, int T, int dir, int first, int stride,
                                              unsigned char* myring_bytes, int blk_bytes, volatile int* myfilled, int lane,
                                              lds_u8* prog, double rr2) {
  const int V = p.V;
  const int nblk = (T + kBlk - 1) / kBlk;
  const float* ytab = p.ytab + (size_t)b * p.T * V;
  const int tt = lane >> 3, l8 = lane & 7;
  // two register sets in flight: the rows of this wave's next block are requested before the current one is stored
  // (unconditional loads from clamped addresses; what a dead row or column reads is replaced when it is used)
  auto load_block = [&](int n, float (&out)[NV]) {
    const int t = block_time(dir, n, tt, T);
    const bool row_live = n < nblk && t < T;
    const float* yr = ytab + (size_t)(row_live ? t : 0) * V + l8;
#pragma unroll
    for (int k = 0; k < NV; k++) out[k] = yr[l8 + 8 * k < V ? 8 * k : 0];
  };
  int consumed = 0;
  float ymin = 1.f;                 // smallest POSITIVE probability this wave has moved (one select + one minimum per element)
  auto process = [&](int n, const float (&yraw)[NV]) {
    if (n >= nblk) return;
    const int t = block_time(dir, n, tt, T);
    const bool row_live = t < T;
    if (n >= RING && consumed < n - RING + 1)       // (bounded: a protocol error flags the utterance instead of hanging)
      HALO_WAIT((consumed = __builtin_amdgcn_readfirstlane(lds_min8(prog))) >= n - RING + 1);
    float* blk32 = reinterpret_cast<float*>(myring_bytes + (size_t)(n % RING) * blk_bytes);
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const int v = l8 + 8 * k;
      if (v < V) {
        const float y = row_live ? yraw[k] : 0.f;
        ymin = fminf(ymin, y > 0.f ? y : 1.f);
        blk32[v * kRow32 + tt] = y;                                        // transposed: [label][step]
      }
    }
    // the blank's pairs: read back from the row this wave has just written (the LDS runs a wave's operations in order)
    if (l8 == 0) {
      const float yb = blk32[p.blank * kRow32 + tt];
      float2 yw; yw.x = yb; yw.y = (float)rr2 * yb;
      *reinterpret_cast<float2*>(blk32 + (V + 1) * kRow32 + 2 * tt) = yw;
    }
    publish(&myfilled[first], n + stride);
  };
  if constexpr (SETS == 2) {
    float xa[NV], xb[NV];
    load_block(first, xa);
    for (int n = first; n < nblk; n += 2 * stride) {
      load_block(n + stride, xb); process(n, xa);
      load_block(n + 2 * stride, xa); process(n + stride, xb);
    }
  } else {
    // one set (56 columns per lane: two do not fit 128 registers): the rows are requested, then the wave waits for its ring
    // slot -- with a ring of two blocks that wait is what hides the loads
    for (int n = first; n < nblk; n += stride) { float xa[NV]; load_block(n, xa); process(n, xa); }
  }
  // Probabilities are f32: below ~2^-126 they are flushed (see prep_wave: reason bit 64, the exact kernel recomputes the
  // utterance); the launch that filled ytab left kTinyProb wherever a finite log-probability lay below -69
  { const bool t1 = __any(ymin < 1e-30f), t2 = __any(ymin < 2e-37f);      // (both votes by the whole wave, outside the lane test)
    if (dir == 0 && t1 && lane == 0) atomicOr(&p.flags[b], t2 ? 64 | 256 : 64); }
}

// Per-utterance lattice description for F2, computed once here instead of once per 16-step segment there (63x at
// T = 1000): label, skip flags and the label-sorted slot of every label pair, plus the first slot of every label.
// Counting sort of the label cells by label: cell i -> start[label] + (its order inside the label); cells past
// the utterance's S labels keep slot i (they only ever hold zeros).
template <int PPL>
__device__ __forceinline__ void cellinfo_wave(const FastParams& p, int b, int T, int S, int* cnt, int lane) {
  const int V = p.V;
  const int nch = (V + 64) >> 6;             // chunks of 64 labels that hold 0..V (two for V <= 127; cnt: [64 * nch + 2] = lstart_ints(V))
  for (int c = 0; c < nch; c++) cnt[64 * c + lane] = 0;
  if (lane < 2) cnt[64 * nch + lane] = 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  LaneCells<PPL> lc;
  lc.load(p.targets + (int64_t)b * p.tgt_stride, S, T, V, p.blank, lane);
  int rank[PPL];
#pragma unroll
  for (int r = 0; r < PPL; r++) {
    const int i = PPL * lane + r;
    rank[r] = (i < S && lc.lab[r] < V) ? atomicAdd(&cnt[lc.lab[r]], 1) : 0;    // ds_add_rtn_u32
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int* ls = p.lstart + (size_t)b * p.LS;
  {
    // exclusive prefix over the label counts, chunk by chunk
    int carry = 0;
    for (int c = 0; c < nch; c++) {
      const int cc = cnt[64 * c + lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int inc = wave_scan(cc) + carry;
      cnt[64 * c + lane] = inc - cc; ls[64 * c + lane] = inc - cc;
      carry = __builtin_amdgcn_readlane(inc, 63);
    }
    if (lane == 63) { cnt[64 * nch] = carry; ls[64 * nch] = carry; ls[64 * nch + 1] = carry; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned* ci = p.cinfo + (size_t)b * (p.CELLS / 2) + PPL * lane;
#pragma unroll
  for (int r = 0; r < PPL; r++) {
    const int i = PPL * lane + r;
    const int slot = (i < S && lc.lab[r] < V) ? cnt[lc.lab[r]] + rank[r] : i;
    ci[r] = LaneCells<PPL>::pack(lc.lab[r], slot, lc.skp[r], lc.skn[r]);
  }
}

// ============================================================================================
// F1, halo form: each chain runs on several waves that exchange cells once per 16 steps (packed f32: the caller's option)
// ============================================================================================
// Cell j of row t depends on cells j, j-1, j-2 of row t-1 only: mass moves by at most one label pair per step.  So a
// wave that OWNS a range of pairs and also carries the h pairs next to them on the upstream side (the halo: the first
// lanes for alpha, the last for beta) can run h steps without hearing from its neighbour -- each step one more halo pair
// goes stale, and after the h-th exactly the owned lanes are still right.  Then the halo is refilled from the
// neighbour's published edge lanes.  Nothing is exchanged per step; h/128 of the arithmetic is redundant.
//   * hand-over: wave w publishes its edge lanes and `prog[w]` at the end of a block; its downstream neighbour waits for
//     that before it needs them.  The same word releases the producers' ring slot.
//   * the power-of-two rescale stays COMMON to the whole row (one frame per direction: edge lanes need no conversion, and
//     the segment kernel sees cumA / cumB / zt2 as with the single-wave chains).  At the end of block n every lane leaves
//     the high word of its largest cell in LDS; a FRAME WAVE per direction reduces the words of block n once all chain
//     waves have passed it and publishes the exponent to remove at the end of block n+2: the absolute exponent of the
//     row's maximum at block n less what has been removed since (absolute: a delayed RELATIVE correction oscillates).
//     A chain wave reads one tagged word per block.  Two blocks of slack keep the waves from running in lock-step.
//   * beta hands ONE value down per step, like alpha hands one up: a lane prepares what the label cell of the pair below
//     takes from its pair (blank * tilted blank probability + label * skip), instead of shipping both cells.
// (An f64 form of this -- one pair per lane, five waves per direction -- was built first: parity-green and exactly as fast
// as the single-wave chains, 91.6 against 94.6 us; see DESIGN.md 4.1c.  It is not in the tree any more.)
constexpr int kHaloSlots = 8;                 // ring depth (blocks) of the published exponents and edge lanes
constexpr int kHaloLag = 2;                   // blocks between measuring the row's exponent and removing it
constexpr int kHaloIdle = 0x3fffffff;         // prog[] of a wave that holds no cell of the utterance

// The frame wave of a direction decides the exponent every chain wave removes at the end of block n: the absolute
// exponent of the row's largest cell at the end of block n - kHaloLag, less what has been removed up to block n-1, so
// that the frame after block n is that absolute exponent.  It also writes the cum exponents for the segment kernel.
// (The chain waves only leave one word per lane and read one word per block; the reductions happen here, two blocks
// ahead of where they are needed.)
// bias: the chain waves keep their cells 2^bias above the frame (f32 cells: the lagged frame leaves the row ~70 bits
// below its unit, and the cells need room under the row's maximum as well).
// LAG: blocks between the measurement and its use.  Two leave the waves a block of slack against each other; one keeps the
// row within 8 steps' decay of its unit (f32 cells), at the price of the waves meeting at every block's end.
template <int DIR, bool F32 = false, int LAG = kHaloLag, bool TRACK = false>
__device__ __forceinline__ void halo_frame_wave(const FastParams& p, int b, int T, lds_u8* L0, int prog_off, int exw_off, int mxl_off,
                                                int maxw, int lane, int W, int bias = 0) {
  static_assert(LAG == 1 || LAG == 2, "the exponents in flight are kept in two variables");
  __builtin_amdgcn_s_setprio(3);                     // (little work, but the chain waves wait for its word every block)
  const int nblk = (T + kBlk - 1) / kBlk;
  const int nres = DIR == 0 ? T / kBlk : nblk;       // blocks whose step 7 is live (alpha's last block may be short)
  const int M = (T - 1) >> 3;
  int* cum = (DIR == 0 ? p.cumA : p.cumB) + (size_t)b * p.NB;
  // TRACK: the row's TRUE exponent per block as well (known here one block late, which does not matter to a kernel that
  // runs afterwards): the segment kernel takes its in-segment rescales from these, the frame of a checkpoint from cum
  int* trk = (DIR == 0 ? p.trkA : p.trkB) + (size_t)b * p.NB;
  lds_u8* prog = L0 + prog_off + DIR * 32;
  lds_u8* exw = L0 + exw_off + DIR * (kHaloSlots * 4);
  lds_u8* mxl = L0 + mxl_off + (DIR * kHaloSlots * maxw * 64 + lane) * 4;
  constexpr int kExMax = F32 ? 100 : 1000;           // (f32 cells: the whole exponent range is 2^+-126)
  if (lane == 0) {
    if (DIR == 0) { cum[0] = 0; if (TRACK) trk[0] = 0; }
    else { cum[M + 1] = 0; cum[M + 2] = 0; if (TRACK) { trk[M + 1] = 0; trk[M + 2] = 0; } }
    for (int n = 0; n < LAG && n < nres; n++) cum[DIR == 0 ? n + 1 : M - n] = 0;      // (their words were set with the flags)
  }
  int through = 0;                    // sum of ex[k], k < n + LAG: the frame after block n + LAG - 1
  int ex1 = 0, ex2 = 0;               // ex[n + 1], ex[n] (LAG 2); ex[n] (LAG 1)
  int absolute = 0;                   // exponent of the row's maximum at the end of block n, in absolute terms
  for (int n = 0; n < (TRACK ? nres : nres - LAG); n++) {
    HALO_WAIT(__builtin_amdgcn_readfirstlane(lds_min8(prog)) >= n + 1);
    int m = 0;
    for (int w = 0; w < W; w++) m = max(m, *(volatile lds_int*)(mxl + ((n & (kHaloSlots - 1)) * maxw + w) * 256));
    m = wave_max(m);                  // positive floating-point numbers order like ints
    if (m > 0) {
      const int e = (F32 ? ((m >> 23) & 0xff) - 127 : ((m >> 20) & 0x7ff) - 1023) - bias;
      absolute = e + (through - ex1 - (LAG == 2 ? ex2 : 0));     // block n was measured before ex[n] was removed
    }
    if (TRACK && lane == 0) trk[DIR == 0 ? n + 1 : M - n] = absolute;
    if (n + LAG < nres) {
      const int ex = m > 0 ? max(min(absolute - through, kExMax), -kExMax) : 0;
      through += ex; ex2 = ex1; ex1 = ex;
      const int nn = n + LAG;
      *(volatile lds_int*)(exw + 4 * (nn & (kHaloSlots - 1))) = (nn << 12) | (ex + 2048);
      if (lane == 0) cum[DIR == 0 ? nn + 1 : M - nn] = through;
    }
  }
}


// the one-pair-per-lane chain kernel (ctc_loss_fast_h1.hip); h1_supported: whether it takes the shape
bool h1_supported(int V, int Smax, int ppl);
int launch_fast_h1_chain(const FastParams& p, int ppl, hipStream_t stream);      // (the caller launches the segment kernel behind it)

}  // namespace fastk
}  // namespace e2e

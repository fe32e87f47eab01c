/* Diagnostics of libe2e_ctc.so -- NOT part of the drop-in contract (that is include/e2e_ctc.h).
 *
 * These entry points exist so that bench.py and tools/diag/ can look inside a call without a second
 * library: they synchronise the device and copy small records to the host, so they never belong in a
 * training step.  A caller that only wants the reference's behaviour never needs this header.
 * Every exported e2e_* symbol of the library is declared in one of the two headers
 * (tests/test_host_cpu.py checks both directions).
 */
#ifndef E2E_CTC_DEBUG_H
#define E2E_CTC_DEBUG_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dst[0..bytes) = src[0..bytes) by a plain streaming kernel on `stream` (bytes % 16 == 0): the on-box
 * copy rate bench.py reports as roofline.peak_measured beside the 8 TB/s spec. */
int e2e_debug_stream_copy(void* dst, const void* src, size_t bytes, void* stream);

/* Keeps `workgroups` workgroups of `threads` threads, each holding `lds_bytes` of LDS, resident for `nanoseconds` (<= 0.1 s) on
 * `stream`, doing nothing: tests/test_gpu_fuzz.py runs the loss call's flagged launch beside it (its bounded waits must end in
 * right results whether they run out or not). */
int e2e_debug_occupy(int workgroups, int threads, int lds_bytes, long long nanoseconds, void* stream);

/* After an e2e_ctc_loss_fwd_bwd(ALGO_AUTO / ALGO_FAST) call that took the fast path: the per-utterance
 * flag words (0 = served by the fast path; bits: 1 bad lengths, 2 blank in targets, 4 infeasible,
 * 8 range / self-check, 16 non-finite, 32 log Z mismatch, 64 emissions near the end of f32) and the
 * alpha-side / beta-side log Z of the chains, read out of `workspace`.  Synchronises. */
int e2e_debug_fast_state(const void* workspace, int B, int T, int V, int Smax, int* flags_host, double* logz_host);

/* How many flagged utterances of that call the f64 redo of the segments could not settle (they were
 * recomputed by the exact kernel).  Synchronises. */
int e2e_debug_fast_redo_failures(const void* workspace, int B, int T, int V, int Smax, int* count_host);

/* The flagged-utterance launch of that call as its workgroup 0 saw it, microseconds since the launch's start (100 MHz clock):
 * us[0] end of its f64 redos of single segments, [1] of the wait for the other workgroups, [2] of its extended-range chains,
 * [3] of its extended-range segments, [4] end of the launch's last workgroup, [5] when that workgroup learnt it was the last
 * (us_host holds six).  Zeros when nothing was flagged.  Synchronises. */
int e2e_debug_flagged_phases(const void* workspace, int B, int T, int V, int Smax, double* us_host);

/* Bounded waits of that call's flagged-utterance launch that ran out (0 on an idle GPU; what they left undone was recomputed by the
 * exact kernel) and f64 redos of single segments that failed (handed to the extended-range redo).  Synchronises. */
int e2e_debug_flagged_counters(const void* workspace, int B, int T, int V, int Smax, int* timeouts_host, int* failed_redos_host);

#ifdef E2E_FAST_PROFILE   /* only in builds made by tools/diag/build_profile_lib.sh */
int e2e_debug_fast_zdev(float* host, int reset);
int e2e_debug_fast_profile(unsigned long long* host, int n);
int e2e_debug_fast_profile2(unsigned long long* host, int reset);
int e2e_debug_fast_profile3(unsigned long long* host, int reset);
#endif
#ifdef E2E_BEAM_PROFILE
int e2e_debug_beam_profile(unsigned long long* host);
int e2e_debug_beam_sigs(unsigned long long* host, int cap);   /* LM states of utterance 0 that had to ask, in order; resets the list */
#endif

#ifdef __cplusplus
}
#endif
#endif

/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference CTC
 * loss / greedy decode / prefix beam search of artbataev/end2end.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker (or the timed CPU baseline).
 * The product path (end2end_amd/) never links, imports or calls it.
 *
 * Parity pins:
 *   - loss/grad: checked against the reference's own compiled engine
 *     (oracle/_ref/cpp_ctc_loss.so, built from /root/reference/src/losses by
 *     oracle/Makefile) and against the reference tests' known answers
 *     (tests/test_ctc.py:69-165) -- see tests/test_oracle_*.py.
 *   - greedy / beam (no LM): pinned on the reference tests' known answers
 *     (tests/test_ctc_decoder.py:44-59,86-166) and on exhaustive enumeration of
 *     every alignment of 20 short utterances (tests/golden/make_beam_golden.py:
 *     a beam that never prunes must return argmax log P - wip * num_words).  The
 *     reference decoder cannot be compiled here (needs KenLM's lm/model.hh,
 *     absent), so pruned-beam behaviour (quirk Q7, tie order) is pinned by
 *     restatement only.
 *   - LM scorer: pinned to the definition of the ARPA format (expected
 *     BaseScore of every (context, word) of two models of 527 / 427 entries,
 *     derived in pure Python by tests/golden/make_lm_golden.py, no code shared
 *     with this file).  Agreement with a KenLM BINARY on a real corpus model is
 *     unpinned: KenLM and its ARPA fixture are absent from /root/reference and
 *     the reference's only LM tests print or are skipped
 *     (tests/test_ctc_decoder.py:62-83,168-181).
 *   - forced alignment: pinned on outputs of the reference's own functions
 *     (pytorch_end2end/utils/alignment.py, run by tests/golden/make_align_golden.py).
 *
 * All arithmetic is IEEE double, as in the reference (scalar_t = double,
 * src/losses/ctc_loss.cpp:10; decode: src/decoders/ctc_decoder.cpp:157).
 */
#ifndef CTC_ORACLE_H
#define CTC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Two-argument log-sum-exp, src/utils/math_utils.h:8-16. */
double oracle_log_sum_exp(double a, double b);

/* Batch driver + per-utterance lattice.
 * Follows src/losses/forward_backward.cpp:7-59 (driver) and
 * src/losses/ctc_loss.cpp:15-118 (compute_2d).
 *   lp       : (B,T,V) log-probabilities, element strides sB,sT,sV (doubles)
 *   targets  : (B,*) int64 rows with stride tgt_stride; first t_len[b] read
 *   losses   : (B) out
 *   grads    : (B,T,V) contiguous out (full Tmax slab incl. padded frames, Q1)
 *   n_threads: 0 = one OS thread per utterance (what the reference does,
 *              forward_backward.cpp:37); k>0 = k worker threads.
 * returns 0, or -1 on bad arguments. */
int oracle_ctc_loss(const double* lp, int64_t sB, int64_t sT, int64_t sV,
                    const int64_t* targets, int64_t tgt_stride,
                    const int64_t* x_len, const int64_t* t_len,
                    int B, int T, int V, int blank,
                    double* losses, double* grads, int n_threads);

/* Greedy decode, src/decoders/ctc_decoder.cpp:443-490.
 *   x        : (B,T,V) logits or log-probs (argmax only)
 *   out      : (B,T) int64, zero-filled then written from the left (Q5)
 *   out_len  : (B) */
int oracle_ctc_greedy(const double* x, int64_t sB, int64_t sT, int64_t sV,
                      const int64_t* x_len, int B, int T, int V, int blank,
                      int64_t* out, int64_t* out_len, int n_threads);

/* Optional n-gram LM used by the beam search (ARPA back-off semantics; stands
 * where KenLM's ProbingModel stands in the reference). */
typedef struct oracle_lm oracle_lm;
oracle_lm* oracle_lm_load_arpa(const char* path, char* err, int errlen);
void oracle_lm_free(oracle_lm* lm);
int oracle_lm_order(const oracle_lm* lm);
/* word -> index (0 = <unk> / not found), exact-case lookup. */
uint32_t oracle_lm_word_index(const oracle_lm* lm, const char* word);
/* log10 p(word | ctx) with back-off; ctx = most-recent-first word ids,
 * ctx_len <= order-1.  Writes the new context (most-recent-first) to out_ctx,
 * returns its length through out_ctx_len. */
double oracle_lm_base_score(const oracle_lm* lm, const uint32_t* ctx, int ctx_len,
                            uint32_t word, uint32_t* out_ctx, int* out_ctx_len);

/* Prefix beam search, src/decoders/ctc_decoder.cpp:153-201 (driver),
 * :353-441 (decode_sentence), :247-312 (get_next_prefix), :314-318 (score).
 *   lp        : (B,T,V) log-probs
 *   labels    : V strings (UTF-8) or NULL (needed only with an LM)
 *   space_id  : index of " " in labels, or -1 (ctc_decoder.cpp:55-59)
 *   lm        : NULL for no LM (then lmwt is forced to 0, :72-74)
 *   out       : (B,max_out) int64 zero-filled; out_len (B): result lengths
 *               (the empty prefix winning yields [-1], length 1 -- Q6)
 * returns 0, -1 bad args, -2 if a result is longer than max_out. */
int oracle_ctc_beam(const double* lp, int64_t sB, int64_t sT, int64_t sV,
                    const int64_t* x_len, int B, int T, int V, int blank,
                    int beam_width, const char* const* labels, int space_id,
                    const oracle_lm* lm, int case_sensitive,
                    double lmwt, double wip, double oov_penalty,
                    int64_t* out, int64_t max_out, int64_t* out_len,
                    int n_threads);

/* Viterbi forced alignment, pytorch_end2end/utils/alignment.py:50-106 (CTC), :10-47 (ASG, is_ctc = 0),
 * driver :109-138.  lp: (B,T,V) log-probs; out: (B,T) int64 pre-filled by the caller (upstream: -100);
 * row b receives its labelling in out[b, :x_len[b]]. */
int oracle_ctc_align(const double* lp, int64_t sB, int64_t sT, int64_t sV,
                     const int64_t* targets, int64_t tgt_stride, const int64_t* x_len, const int64_t* t_len,
                     int B, int T, int V, int blank, int is_ctc, int64_t* out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif

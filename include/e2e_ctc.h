/*
 * e2e_ctc.h -- C ABI of the MI355X-native CTC loss-and-decode library
 * (libe2e_ctc.so, built from end2end_amd/csrc/ with hipcc --offload-arch=gfx950).
 *
 * This is the drop-in boundary for the hot path of artbataev/end2end: every
 * entry point replaces one method of the reference's pybind11 engines
 * (citations are file:line in the reference tree).  Plain pointers and sizes
 * only; no torch / pybind types.  All data pointers are DEVICE pointers on the
 * current HIP device unless a parameter says "host".  Every call is
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null
 * stream), allocates nothing and never synchronises, so it can be captured in
 * a hipGraph; the caller owns every buffer.
 *
 * Return value: 0 on success, a negative E2E_ERR_* code otherwise;
 * e2e_last_error() then returns a thread-local message.  No C++ exception
 * crosses this boundary.
 */
#ifndef E2E_CTC_H
#define E2E_CTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2E_CTC_ABI_VERSION 3

/* element types of the logits / log-prob tensor (losses and grads use the same) */
#define E2E_F32 0
#define E2E_F64 1
/* 16-bit inputs (e2e_ctc_loss_fwd_bwd(_opt) under E2E_ALGO_AUTO / E2E_ALGO_FAST, e2e_ctc_greedy): the logits are read in
 * their own dtype, the lattice runs in f32 (fast / wide path; reference: src/losses/forward_backward.cpp:15 converts once
 * to double), and the gradient is written in the source dtype.  `losses` and `reduced` are FLOAT32 for these two -- a
 * loss of a few hundred has three digits in bf16 --; the Python engine converts the B losses at the end.  Shapes the
 * fast and wide paths do not take return E2E_ERR_UNSUPPORTED for these dtypes (up-cast and call again). */
#define E2E_F16 2
#define E2E_BF16 3

#define E2E_OK 0
#define E2E_ERR_ARG (-1)          /* bad argument (null pointer, bad size/dtype) */
#define E2E_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels support */
#define E2E_ERR_WORKSPACE (-3)    /* workspace too small */
#define E2E_ERR_HIP (-4)          /* a HIP launch / runtime call failed */
#define E2E_ERR_IO (-5)           /* LM file could not be read / parsed (host) */

/* which CTC loss algorithm to run */
#define E2E_ALGO_AUTO 0    /* fast scaled path for f32 where valid; otherwise the exact kernel -- for f32 in its rescaled
                            * f64 probability-domain form, for f64 in the reference's log domain */
#define E2E_ALGO_EXACT 1   /* f64 log-domain lattice, the reference's arithmetic */
#define E2E_ALGO_FAST 2    /* scaled linear-domain lattice, flags invalid utterances */

int e2e_ctc_abi_version(void);
const char* e2e_last_error(void);

/* ------------------------------------------------------------------------
 * CTC loss forward + backward.
 * Replaces cpp_ctc_loss.CTCLossEngine(blank_idx).compute(logits, targets,
 * logits_lengths, targets_lengths) -> (losses[B], grads[B,T,V])
 *   src/losses/ctc_loss_py.cpp:8-16, src/losses/forward_backward.cpp:7-59,
 *   src/losses/ctc_loss.cpp:15-118.
 *
 *   x            (B,T,V) tensor with element strides sB,sT,sV (a time-major
 *                permuted view is fine); dtype E2E_F32, E2E_F64, or -- where e2e_ctc_loss_takes_dtype() says so --
 *                E2E_F16 / E2E_BF16 (read as they are: no f32 copy exists)
 *   input_is_logprobs
 *                1: x holds log-probabilities -- exactly the reference engine:
 *                   grads = exp(x) - posterior over the full (T,V) slab, rows
 *                   t >= x_len[b] come out as exp(x) (reference quirk Q1);
 *                0: x holds raw logits; log_softmax(dim=V) is fused in and
 *                   grads are d loss / d logits (= softmax - posterior for
 *                   t < x_len[b], 0 for padded rows), i.e. what
 *                   pytorch_end2end/modules/ctc_loss.py:37-40 + autograd give.
 *   targets      (B,*) int64, row stride tgt_stride, first t_len[b] entries used
 *   x_len,t_len  (B) int64 (1 <= x_len[b] <= T, 0 <= t_len[b] <= Smax).  An utterance whose lengths are
 *                outside these ranges, or whose first t_len[b] targets contain a value outside [0,V), gets
 *                loss = NaN and a NaN gradient slab (the reference reads out of bounds there); the other
 *                utterances of the batch are unaffected and the call still returns 0.
 *   losses       (B)  same dtype as x -- f32 for 16-bit x;  +inf for an infeasible alignment (Q2)
 *   grads        (B,T,V) contiguous, same dtype as x (16-bit x: 16-bit gradient); NaN slab when infeasible
 *   workspace    >= e2e_ctc_loss_workspace_bytes(...) bytes, 256-B aligned
 *   algo         E2E_ALGO_*
 */
size_t e2e_ctc_loss_workspace_bytes(int B, int T, int V, int Smax, int dtype, int algo);

/* 1 if a loss call with these logits runs as it is; 0 if the caller has to up-cast them to f32 first (then the call would
 * return E2E_ERR_UNSUPPORTED).  Always 1 for E2E_F32 / E2E_F64.  16-bit logits are read natively by the lattice kernels
 * (alphabets of <= 448 columns, targets of <= 447 labels: any strides) and by the wide-alphabet path -- in one pass when its rows
 * are contiguous (sV == 1), V % 8 == 0, V <= 8192, strides of whole 16-byte pieces, x and grads 16-byte aligned; element by
 * element in two passes otherwise (since ABI 3).  What is left for 0: shapes only the exact kernel takes (E2E_ALGO_EXACT, or
 * targets beyond 447 labels on an alphabet the wide path does not compact). */
int e2e_ctc_loss_takes_dtype(int dtype, int algo, int T, int V, int Smax, int64_t sB, int64_t sT, int64_t sV,
                             const void* x, const void* grads);

int e2e_ctc_loss_fwd_bwd(const void* x, int dtype, int input_is_logprobs,
                         int64_t sB, int64_t sT, int64_t sV,
                         const int64_t* targets, int64_t tgt_stride,
                         const int64_t* x_len, const int64_t* t_len,
                         int B, int T, int V, int Smax, int blank,
                         void* losses, void* grads,
                         void* workspace, size_t workspace_bytes,
                         int algo, void* stream);

/* The same call with options (NULL = the plain call):
 *   grad_scale  every gradient element is multiplied by it as it is written (NaN slabs stay NaN).  The module passes
 *               1/B for reduce=True, size_average=True, so that the autograd backward of the mean
 *               (pytorch_end2end/modules/ctc_loss.py:52-56 + functions/forward_backward.py:33) has nothing left to do.
 *   reduced     NULL, or one element of x's dtype that receives the sum (E2E_REDUCE_SUM) or mean (E2E_REDUCE_MEAN) of
 *               the B losses (+inf / NaN propagate as they would through torch.sum), in a fixed order (deterministic).
 *               On the f32 small-alphabet path it is written by the launch that also looks for flagged utterances, so
 *               the call has no launch more than without it.
 *   chains      arithmetic of the lattice chains on the f32 small-alphabet path.  E2E_CHAINS_F64 (0, the default): every
 *               result within ~1e-6 of the reference's f64 (gradient elements: 2e-6 absolute).  E2E_CHAINS_F32: the
 *               chains run in packed f32 where that is faster (targets longer than 127 labels: ~10 % on the step at
 *               B=256, T=1000, V=29, S<=200); losses still within 2e-6 relative, gradient elements within 2e-5 absolute -- inside the
 *               1e-4 the drop-in promises, for callers that train in f32 / bf16 anyway.  Elsewhere it changes nothing. */
#define E2E_REDUCE_NONE 0
#define E2E_REDUCE_SUM 1
#define E2E_REDUCE_MEAN 2
#define E2E_CHAINS_F64 0
#define E2E_CHAINS_F32 1
typedef struct e2e_ctc_loss_opts {
  double grad_scale;
  void* reduced;
  int reduction;
  int chains;
} e2e_ctc_loss_opts;

int e2e_ctc_loss_fwd_bwd_opt(const void* x, int dtype, int input_is_logprobs,
                             int64_t sB, int64_t sT, int64_t sV,
                             const int64_t* targets, int64_t tgt_stride,
                             const int64_t* x_len, const int64_t* t_len,
                             int B, int T, int V, int Smax, int blank,
                             void* losses, void* grads,
                             void* workspace, size_t workspace_bytes,
                             int algo, void* stream, const e2e_ctc_loss_opts* opts);

/* grads[b,:,:] *= scale[b]  in place (rows whose factor is exactly 1 are not touched): the multiply of the autograd backward
 * (pytorch_end2end/functions/forward_backward.py:33) without a second (B,T,V) tensor. */
int e2e_ctc_scale_grads(void* grads, int dtype, const void* scale /* (B) same dtype */,
                        int B, int64_t row_elems /* T*V */, void* stream);

/* ------------------------------------------------------------------------
 * Greedy decode.  Replaces cpp_ctc_decoder.CTCDecoder.decode_greedy
 *   src/decoders/ctc_decoder.cpp:443-490 (argmax + blank/repeat collapse).
 *   x        (B,T,V) logits or log-probs, strides sB,sT,sV, f32/f64
 *   x_len    (B) int64
 *   out      (B,T) int64, written zero-padded on the right (quirk Q5)
 *   out_len  (B) int64
 */
int e2e_ctc_greedy(const void* x, int dtype, int64_t sB, int64_t sT, int64_t sV,
                   const int64_t* x_len, int B, int T, int V, int blank,
                   int64_t* out, int64_t* out_len, void* stream);

/* ------------------------------------------------------------------------
 * n-gram language model (stands where KenLM's ProbingModel stands in the
 * reference: src/decoders/ctc_decoder.cpp:60-71 load, :77-88 get_idx,
 * :275-278,:291-294 BaseScore).  Host-side ARPA reader (plain or .gz) that
 * builds a device-resident hash table.  `labels` are the decoder's V label
 * strings (UTF-8): words are spelled with them.
 *
 * Hashed matching: on the device a word is found by the 64-bit FNV hash of its
 * spelling and an n-gram by a 64-bit signature of its word ids; the entries keep
 * the hash, not the spelling / ids.  What the model lists is always found.  A
 * query the model does NOT list -- an out-of-vocabulary spelling, an unseen
 * n-gram -- is taken for a listed one if the two hashes collide: about 2^-64
 * per probe, i.e. correct with overwhelming probability, not by construction
 * (KenLM's probing model matches on 64-bit hashes in the same way).  The host
 * scorer below (e2e_lm_score) compares ids and is exact.  A model that lists an
 * n-gram without its context (SRILM-pruned ARPA files) is served from id-keyed
 * tables, which are slower.
 */
typedef struct e2e_lm e2e_lm;
/* Reads the model and uploads its tables to the CURRENT HIP device (this call allocates and synchronises; it is
 * the one entry point that does).  A model serves calls on that device only: load it once per device. */
int e2e_lm_load_arpa(const char* path /* host */, const char* const* labels /* host */, int V,
                     int case_sensitive, e2e_lm** out);
void e2e_lm_free(e2e_lm* lm);
int e2e_lm_order(const e2e_lm* lm);
int e2e_lm_device(const e2e_lm* lm);   /* HIP device index of the tables; -1 = host tables only (no GPU at load) */
/* host-side scoring helpers (testing / print_scores_for_sentence,
 * src/decoders/ctc_decoder.cpp:141-151): word index (0 = <unk>) and
 * log10 p(word | most-recent-first context ids). */
uint32_t e2e_lm_word_index(const e2e_lm* lm, const char* word /* host */);
double e2e_lm_score(const e2e_lm* lm, const uint32_t* ctx /* host */, int ctx_len, uint32_t word);

/* ------------------------------------------------------------------------
 * Prefix beam search.  Replaces cpp_ctc_decoder.CTCDecoder.decode
 *   src/decoders/ctc_decoder.cpp:153-201 (driver), :353-441 (decode_sentence),
 *   :247-312 (get_next_prefix), :314-318 (score).
 *   lp        (B,T,V) LOG-PROBABILITIES, strides sB,sT,sV, f32 / f64 / f16 / bf16 (16-bit values are read as they are: the
 *             search is the one of their f32 images, bit for bit)
 *   space_id  index of " " among the labels, or -1 (:55-59)
 *   lm        NULL for none (lmwt then counts as 0, :72-74); else a model loaded on the current device
 *             with exactly V labels (E2E_ERR_ARG otherwise)
 *   out       (B,max_out) int64, zero-filled; out_len (B) int64.  When the
 *             empty prefix wins the result is the single id -1 (quirk Q6).
 *             Per-utterance status rides on out_len (no separate call, nothing synchronises):
 *               0 <= out_len[b] <= max_out   the sentence is out[b, :out_len[b]]
 *               out_len[b] > max_out         the sentence has out_len[b] ids, only the first max_out were
 *                                            written (max_out = x_len[b] + 1 can never be exceeded)
 *               out_len[b] == -1             the prefix-tree node pool ran out (cannot happen with a workspace
 *                                            of e2e_ctc_beam_workspace_bytes(); the row must not be used)
 *   workspace >= e2e_ctc_beam_workspace_bytes(...)
 * Limits (E2E_ERR_UNSUPPORTED beyond them; the reference has none): beam_width <= e2e_ctc_beam_max_width() = 512,
 * language models of order <= 6.
 */
size_t e2e_ctc_beam_workspace_bytes(int B, int T, int V, int beam_width);
/* The same for a caller that knows whether the call will carry a language model (with_lm 0 / 1): only what that call needs --
 * without a model the general kernel's rows of LM answers are left out, and nothing of the general kernel's is counted where
 * the one-workgroup kernel takes the call.  e2e_ctc_beam_workspace_bytes() is the larger of the two. */
size_t e2e_ctc_beam_workspace_bytes_lm(int B, int T, int V, int beam_width, int with_lm);
/* The largest beam_width e2e_ctc_beam accepts for an alphabet of V columns, with or without a language model: 512 for
 * any V.  Two kernels stand behind the call: the fast one keeps everything that scales with beam_width * V in one
 * workgroup's LDS (V = 29: widths up to 150, 103 with an LM; V = 80: 81 / 47); beyond that the general kernel keeps the
 * candidate keys and the LM's answers in the workspace (any V, e.g. the reference's default width 100 at V = 8000) --
 * same result, slower per step; beyond ~256 hypotheses it also keeps the beam members' state there.  Host callers check the width at construction instead of failing at the first decode. */
int e2e_ctc_beam_max_width(int V, int with_lm);

int e2e_ctc_beam(const void* lp, int dtype, int64_t sB, int64_t sT, int64_t sV,
                 const int64_t* x_len, int B, int T, int V, int blank,
                 int beam_width, int space_id, const e2e_lm* lm,
                 double lmwt, double wip, double oov_penalty,
                 int64_t* out, int64_t max_out, int64_t* out_len,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Viterbi forced alignment on the same lattice (max-plus instead of sum).
 * Replaces pytorch_end2end/utils/alignment.py:50-106 (_get_alignment_ctc_1d), :10-47
 * (_get_alignment_asg_1d, is_ctc = 0: no blanks) and the batch driver :109-138
 * (get_alignment_3d), which run as numba-jitted Python on the host upstream.
 *   lp        (B,T,V) LOG-PROBABILITIES, strides sB,sT,sV, f32 / f64 / f16 / bf16 (alpha is f64 either way, as upstream)
 *   targets   (B,*) int64, first t_len[b] entries used; x_len, t_len (B) int64
 *   blank     the blank id (upstream hard-codes 0, :57); ignored when is_ctc = 0
 *   out       (B,T) int64: out[b, t] = the label (or blank) frame t is aligned to for t < x_len[b], pad_value beyond
 *             (upstream fills -100, :132).  Ties keep the earlier candidate in the order stay, previous cell, skip, as
 *             upstream's strict ">" does.  Too few frames for the labelling is not an error upstream and is not one
 *             here (same walk over -inf cells).  An utterance with invalid lengths or a target outside [0,V) gets a
 *             row of pad_value.
 *   workspace >= e2e_ctc_align_workspace_bytes(...): one back-pointer byte per lattice cell and frame
 */
size_t e2e_ctc_align_workspace_bytes(int B, int T, int V, int Smax, int is_ctc);

int e2e_ctc_align(const void* lp, int dtype, int64_t sB, int64_t sT, int64_t sV,
                  const int64_t* targets, int64_t tgt_stride,
                  const int64_t* x_len, const int64_t* t_len,
                  int B, int T, int V, int Smax, int blank, int is_ctc,
                  int64_t* out, int64_t pad_value,
                  void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E2E_CTC_H */

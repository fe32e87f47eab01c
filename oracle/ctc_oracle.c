/*
 * TEST INFRASTRUCTURE ONLY -- see ctc_oracle.h.  Plain-C restatement of the
 * reference algorithms; every function cites the reference lines it follows.
 * Not used, linked or imported by the product path.
 */
#define _POSIX_C_SOURCE 200809L
#include "ctc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define NEG_INF (-INFINITY)

/* src/utils/math_utils.h:8-16 -- note log(1.0 + x), not log1p. */
double oracle_log_sum_exp(double a, double b) {
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  if (a > b) return a + log(1.0 + exp(b - a));
  return b + log(1.0 + exp(a - b));
}
#define LSE oracle_log_sum_exp

/* ------------------------------------------------------------------ */
/* thread-per-utterance driver (src/utils/threadpool.cpp:7-44: N threads
 * draining a FIFO of B tasks; the reference always uses N = B).       */
typedef void (*utt_fn)(void* ctx, int b);
typedef struct {
  utt_fn fn; void* ctx; int B; int next; pthread_mutex_t mu;
} pool_t;

static void* pool_worker(void* p_) {
  pool_t* p = (pool_t*)p_;
  for (;;) {
    pthread_mutex_lock(&p->mu);
    int b = p->next++;
    pthread_mutex_unlock(&p->mu);
    if (b >= p->B) break;
    p->fn(p->ctx, b);
  }
  return NULL;
}

static void run_pool(utt_fn fn, void* ctx, int B, int n_threads) {
  if (B <= 0) return;
  int n = n_threads > 0 ? n_threads : B;
  if (n > B) n = B;
  pool_t p; p.fn = fn; p.ctx = ctx; p.B = B; p.next = 0;
  pthread_mutex_init(&p.mu, NULL);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n);
  int started = 0;
  pthread_attr_t at; pthread_attr_init(&at);
  pthread_attr_setstacksize(&at, 1 << 20);
  for (int i = 0; i < n; i++) {
    if (pthread_create(&th[started], &at, pool_worker, &p) == 0) started++;
  }
  if (started == 0) pool_worker(&p); /* degrade to serial */
  for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
  pthread_attr_destroy(&at);
  free(th);
  pthread_mutex_destroy(&p.mu);
}

/* ------------------------------------------------------------------ */
/* CTC loss                                                            */
typedef struct {
  const double* lp; int64_t sB, sT, sV;
  const int64_t* targets; int64_t tgt_stride;
  const int64_t* x_len; const int64_t* t_len;
  int B, T, V, blank;
  double* losses; double* grads;
} loss_ctx;

/* src/losses/ctc_loss.cpp:15-118 (compute_2d) for utterance b. */
static void loss_one(void* c_, int b) {
  loss_ctx* c = (loss_ctx*)c_;
  const int T = (int)c->x_len[b];            /* seq_len */
  const int S = (int)c->t_len[b];            /* targets_len */
  const int L = 2 * S + 1;                   /* ext_targets_len, :25 */
  const int V = c->V, Tmax = c->T, blank = c->blank;
  const double* lp = c->lp + (int64_t)b * c->sB;
#define LP(t, v) lp[(int64_t)(t) * c->sT + (int64_t)(v) * c->sV]

  int64_t* ext = (int64_t*)malloc(sizeof(int64_t) * (size_t)L);
  for (int j = 0; j < L; j++) ext[j] = blank;                     /* :26-28 */
  for (int i = 0; i < S; i++) ext[2 * i + 1] = c->targets[(int64_t)b * c->tgt_stride + i]; /* :30-31 */

  /* label-major [j][t] storage as in the reference (:34-36) */
  size_t cells = (size_t)L * (size_t)(T > 0 ? T : 1);
  double* alpha = (double*)malloc(sizeof(double) * cells);
  double* beta = (double*)malloc(sizeof(double) * cells);
  for (size_t i = 0; i < cells; i++) { alpha[i] = NEG_INF; beta[i] = NEG_INF; }
#define A(j, t) alpha[(size_t)(j) * (size_t)T + (size_t)(t)]
#define Bt(j, t) beta[(size_t)(j) * (size_t)T + (size_t)(t)]

  /* alpha, :38-61 */
  if (T > 1 || L == 1) A(0, 0) = LP(0, ext[0]);
  if (L > 1) A(1, 0) = LP(0, ext[1]);
  for (int t = 1; t < T; t++) {
    int start = L - 2 * (T - t); if (start < 0) start = 0;
    int end = t * 2 + 2; if (end > L) end = L;
    for (int j = start; j < end; j++) {
      double a = A(j, t - 1);
      int64_t cur = ext[j];
      if (j > 0) {
        a = LSE(a, A(j - 1, t - 1));
        if (cur != blank && j - 2 >= 0 && ext[j - 2] != cur) a = LSE(a, A(j - 2, t - 1));
      }
      A(j, t) = a + LP(t, cur);
    }
  }
  /* loss, :63-70 */
  double loss;
  if (L > 1) loss = -LSE(A(L - 1, T - 1), A(L - 2, T - 1));
  else loss = -A(L - 1, T - 1);
  c->losses[b] = loss;
  const double loss_forward = -loss;

  /* beta, :72-100 (beta excludes the emission at t) */
  if (T > 1 || L == 1) Bt(L - 1, T - 1) = 0;
  if (L > 1) Bt(L - 2, T - 1) = 0;
  for (int t = T - 2; t >= 0; t--) {
    int start = L - 2 * (T - t); if (start < 0) start = 0;
    int end = t * 2 + 2; if (end > L) end = L;
    for (int j = start; j < end; j++) {
      int64_t cur = ext[j];
      double v = Bt(j, t + 1) + LP(t + 1, ext[j]);
      if (j < L - 1) {
        v = LSE(v, Bt(j + 1, t + 1) + LP(t + 1, ext[j + 1]));
        if (cur != blank && j + 2 < L && ext[j + 2] != cur)
          v = LSE(v, Bt(j + 2, t + 1) + LP(t + 1, ext[j + 2]));
      }
      Bt(j, t) = v;
    }
  }

  /* gradient, :102-117: prob_sum over the FULL (Tmax,V) slab stays -inf for
   * t >= seq_len, so padded rows come out as exp(lp) (quirk Q1). */
  double* prob_sum = (double*)malloc(sizeof(double) * (size_t)Tmax * (size_t)V);
  for (size_t i = 0; i < (size_t)Tmax * (size_t)V; i++) prob_sum[i] = NEG_INF;
  for (int i = 0; i < L; i++) {
    int64_t cur = ext[i];
    for (int j = 0; j < T; j++) {
      double ab = A(i, j) + Bt(i, j);
      double* ps = &prob_sum[(size_t)j * (size_t)V + (size_t)cur];
      *ps = LSE(*ps, ab);
    }
  }
  double* g = c->grads + (size_t)b * (size_t)Tmax * (size_t)V;
  for (int t = 0; t < Tmax; t++)
    for (int v = 0; v < V; v++)
      g[(size_t)t * (size_t)V + (size_t)v] =
          exp(LP(t, v)) - exp(prob_sum[(size_t)t * (size_t)V + (size_t)v] - loss_forward);

  free(prob_sum); free(alpha); free(beta); free(ext);
#undef A
#undef Bt
#undef LP
}

int oracle_ctc_loss(const double* lp, int64_t sB, int64_t sT, int64_t sV,
                    const int64_t* targets, int64_t tgt_stride,
                    const int64_t* x_len, const int64_t* t_len,
                    int B, int T, int V, int blank,
                    double* losses, double* grads, int n_threads) {
  if (!lp || !x_len || !t_len || !losses || !grads || B < 0 || T < 1 || V < 1) return -1;
  for (int b = 0; b < B; b++)
    if (x_len[b] < 1 || x_len[b] > T || t_len[b] < 0) return -1;
  loss_ctx c = {lp, sB, sT, sV, targets, tgt_stride, x_len, t_len, B, T, V, blank, losses, grads};
  run_pool(loss_one, &c, B, n_threads);
  return 0;
}

/* ------------------------------------------------------------------ */
/* greedy decode, src/decoders/ctc_decoder.cpp:443-490                 */
typedef struct {
  const double* x; int64_t sB, sT, sV; const int64_t* x_len;
  int B, T, V, blank; int64_t* out; int64_t* out_len;
} greedy_ctx;

static void greedy_one(void* c_, int b) {
  greedy_ctx* c = (greedy_ctx*)c_;
  const double* x = c->x + (int64_t)b * c->sB;
  int64_t* o = c->out + (int64_t)b * c->T;
  int64_t prev = c->blank, n = 0;
  for (int t = 0; t < (int)c->x_len[b]; t++) {
    /* argmax(-1), :451 -- first maximum wins on ties; NaN counts as maximal
     * (torch CPU argmax semantics). */
    const double* r = x + (int64_t)t * c->sT;
    int best = 0; double bv = r[0];
    for (int v = 1; v < c->V; v++) {
      double xv = r[(int64_t)v * c->sV];
      if (!(bv != bv) && (xv > bv || xv != xv)) { bv = xv; best = v; }
    }
    if (best != c->blank && prev != best) o[n++] = best;   /* :475-480 */
    prev = best;                                           /* :481 */
  }
  c->out_len[b] = n;
}

int oracle_ctc_greedy(const double* x, int64_t sB, int64_t sT, int64_t sV,
                      const int64_t* x_len, int B, int T, int V, int blank,
                      int64_t* out, int64_t* out_len, int n_threads) {
  if (!x || !x_len || !out || !out_len || B < 0 || T < 1 || V < 1) return -1;
  memset(out, 0, sizeof(int64_t) * (size_t)B * (size_t)T);  /* zeros_like, :452 */
  greedy_ctx c = {x, sB, sT, sV, x_len, B, T, V, blank, out, out_len};
  run_pool(greedy_one, &c, B, n_threads);
  return 0;
}

/* ------------------------------------------------------------------ */
/* ARPA back-off LM (stands where KenLM stands; PARITY UNPINNED)       */
#define LM_MAX_ORDER 6      /* KENLM_MAX_ORDER=6, CMakeLists.txt:36 */

typedef struct {
  uint32_t ids[LM_MAX_ORDER]; int n; float prob; float backoff; int used;
} ngram_slot;

typedef struct { char* word; uint32_t id; int used; } vocab_slot;

struct oracle_lm {
  int order;
  ngram_slot* tab; size_t cap;
  vocab_slot* vtab; size_t vcap; uint32_t n_words;
  vocab_slot* ltab;           /* lower-cased words (case-insensitive lookup) */
  uint32_t bos, eos;
};

static uint64_t fnv64(const void* p, size_t n, uint64_t h) {
  const unsigned char* s = (const unsigned char*)p;
  for (size_t i = 0; i < n; i++) { h ^= s[i]; h *= 1099511628211ULL; }
  return h;
}

static vocab_slot* table_find(vocab_slot* tab, size_t vcap, const char* w, int insert) {
  uint64_t h = fnv64(w, strlen(w), 1469598103934665603ULL);
  size_t i = (size_t)(h % vcap);
  for (;;) {
    vocab_slot* s = &tab[i];
    if (!s->used) return insert ? s : NULL;
    if (strcmp(s->word, w) == 0) return s;
    i = (i + 1) % vcap;
  }
}
static vocab_slot* vocab_find(const oracle_lm* lm, const char* w, int insert) {
  return table_find(lm->vtab, lm->vcap, w, insert);
}

static ngram_slot* ngram_find(const oracle_lm* lm, const uint32_t* ids, int n, int insert) {
  uint64_t h = fnv64(ids, sizeof(uint32_t) * (size_t)n, 1469598103934665603ULL ^ (uint64_t)n);
  size_t i = (size_t)(h % lm->cap);
  for (;;) {
    ngram_slot* s = &lm->tab[i];
    if (!s->used) return insert ? s : NULL;
    if (s->n == n && memcmp(s->ids, ids, sizeof(uint32_t) * (size_t)n) == 0) return s;
    i = (i + 1) % lm->cap;
  }
}

static uint32_t vocab_intern(oracle_lm* lm, const char* w) {
  vocab_slot* s = vocab_find(lm, w, 1);
  if (!s->used) { s->used = 1; s->word = strdup(w); s->id = lm->n_words++; }
  return s->id;
}

oracle_lm* oracle_lm_load_arpa(const char* path, char* err, int errlen) {
  FILE* f = fopen(path, "r");
  if (!f) { if (err) snprintf(err, (size_t)errlen, "cannot open %s", path); return NULL; }
  oracle_lm* lm = (oracle_lm*)calloc(1, sizeof(oracle_lm));
  size_t counts[LM_MAX_ORDER + 1] = {0}, total = 0;
  char* line = NULL; size_t cap = 0; ssize_t len;
  int section = 0; /* 0 header, k = k-grams */
  /* pass 1: counts */
  while ((len = getline(&line, &cap, f)) >= 0) {
    int k; size_t n;
    if (sscanf(line, "ngram %d=%zu", &k, &n) == 2 && k >= 1 && k <= LM_MAX_ORDER) {
      counts[k] = n; total += n; if (k > lm->order) lm->order = k;
    }
    if (line[0] == '\\' && strstr(line, "-grams:")) break;
  }
  if (lm->order == 0) {
    if (err) snprintf(err, (size_t)errlen, "%s: no \\data\\ header", path);
    free(line); fclose(f); free(lm); return NULL;
  }
  lm->cap = total * 2 + 64; lm->tab = (ngram_slot*)calloc(lm->cap, sizeof(ngram_slot));
  lm->vcap = counts[1] * 2 + 64; lm->vtab = (vocab_slot*)calloc(lm->vcap, sizeof(vocab_slot));
  vocab_intern(lm, "<unk>");                     /* index 0 == NotFound() */
  rewind(f);
  while ((len = getline(&line, &cap, f)) >= 0) {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    if (len == 0) continue;
    if (line[0] == '\\') {
      int k;
      if (sscanf(line, "\\%d-grams:", &k) == 1) section = k; else if (strncmp(line, "\\end\\", 5) == 0) break;
      continue;
    }
    if (section == 0) continue;
    /* logprob \t w1 ... wk [\t backoff] */
    char* save = NULL; char* tok = strtok_r(line, " \t", &save);
    if (!tok) continue;
    float prob = strtof(tok, NULL);
    uint32_t ids[LM_MAX_ORDER]; int ok = 1;
    for (int i = 0; i < section; i++) {
      tok = strtok_r(NULL, " \t", &save);
      if (!tok) { ok = 0; break; }
      ids[i] = vocab_intern(lm, tok);
    }
    if (!ok) continue;
    tok = strtok_r(NULL, " \t", &save);
    float bo = tok ? strtof(tok, NULL) : 0.0f;
    ngram_slot* s = ngram_find(lm, ids, section, 1);
    s->used = 1; s->n = section; memcpy(s->ids, ids, sizeof(uint32_t) * (size_t)section);
    s->prob = prob; s->backoff = bo;
  }
  free(line); fclose(f);
  { /* <unk> absent from the file: KenLM default unknown_missing_logprob = -100 */
    uint32_t z = 0;
    if (!ngram_find(lm, &z, 1, 0)) {
      ngram_slot* s = ngram_find(lm, &z, 1, 1);
      s->used = 1; s->n = 1; s->ids[0] = 0; s->prob = -100.0f; s->backoff = 0.0f;
    }
  }
  /* lower-cased map, ctc_decoder.cpp:68-70 (which entry wins when two words
   * lower-case to the same string is unspecified there; here: lowest id) */
  lm->ltab = (vocab_slot*)calloc(lm->vcap, sizeof(vocab_slot));
  {
    vocab_slot** by_id = (vocab_slot**)calloc(lm->n_words, sizeof(vocab_slot*));
    for (size_t i = 0; i < lm->vcap; i++) if (lm->vtab[i].used) by_id[lm->vtab[i].id] = &lm->vtab[i];
    for (uint32_t i = 0; i < lm->n_words; i++) {
      char* lw = strdup(by_id[i]->word);
      for (char* p = lw; *p; p++) *p = (char)tolower((unsigned char)*p);
      vocab_slot* sl = table_find(lm->ltab, lm->vcap, lw, 1);
      if (!sl->used) { sl->used = 1; sl->word = lw; sl->id = i; } else free(lw);
    }
    free(by_id);
  }
  vocab_slot* vs = vocab_find(lm, "<s>", 0); lm->bos = vs ? vs->id : 0;
  vs = vocab_find(lm, "</s>", 0); lm->eos = vs ? vs->id : 0;
  return lm;
}

void oracle_lm_free(oracle_lm* lm) {
  if (!lm) return;
  for (size_t i = 0; i < lm->vcap; i++) if (lm->vtab[i].used) free(lm->vtab[i].word);
  for (size_t i = 0; i < lm->vcap; i++) if (lm->ltab[i].used) free(lm->ltab[i].word);
  free(lm->ltab); free(lm->vtab); free(lm->tab); free(lm);
}

int oracle_lm_order(const oracle_lm* lm) { return lm->order; }

uint32_t oracle_lm_word_index(const oracle_lm* lm, const char* word) {
  vocab_slot* s = vocab_find(lm, word, 0);
  return s ? s->id : 0;
}

/* ARPA back-off: p(w|c_1..c_n) = p(c..w) if listed, else bo(c) + p(w|c_2..c_n).
 * ctx is most-recent-first.  Float accumulation in KenLM's order (lm/model.cc FullScore: the float prob of the
 * longest listed n-gram, then the float back-off weights of the longer contexts added shortest context first),
 * returned as float widened to double. */
double oracle_lm_base_score(const oracle_lm* lm, const uint32_t* ctx, int ctx_len,
                            uint32_t word, uint32_t* out_ctx, int* out_ctx_len) {
  int n = ctx_len; if (n > lm->order - 1) n = lm->order - 1;
  uint32_t ids[LM_MAX_ORDER];
  float bo[LM_MAX_ORDER + 1];
  float result = 0.0f; int found_k = -1;
  for (int k = n; k >= 0; k--) {
    /* k context words, oldest first, then the word */
    for (int i = 0; i < k; i++) ids[i] = ctx[k - 1 - i];
    ids[k] = word;
    ngram_slot* s = ngram_find(lm, ids, k + 1, 0);
    if (s) { result = s->prob; found_k = k; break; }
    bo[k] = 0.0f;
    if (k > 0) {
      ngram_slot* c = ngram_find(lm, ids, k, 0);   /* the context n-gram */
      if (c) bo[k] = c->backoff;
    }
  }
  if (found_k < 0) { uint32_t z = 0; result = ngram_find(lm, &z, 1, 0)->prob; found_k = 0; }
  for (int k = found_k + 1; k <= n; k++) result += bo[k];
  if (out_ctx) {
    int m = n + 1; if (m > lm->order - 1) m = lm->order - 1;
    uint32_t tmp[LM_MAX_ORDER];
    if (m > 0) tmp[0] = word;
    for (int i = 1; i < m; i++) tmp[i] = ctx[i - 1];
    memcpy(out_ctx, tmp, sizeof(uint32_t) * (size_t)m);
    *out_ctx_len = m;
  }
  return (double)result;
}

/* ------------------------------------------------------------------ */
/* prefix beam search                                                  */
typedef struct { uint32_t w[LM_MAX_ORDER]; int n; } lm_state;

/* CTCDecoder::Prefix, src/decoders/ctc_decoder.h:69-96; ctor .cpp:320-331.
 * shared_ptr/weak_ptr ownership is restated as an explicit reference count:
 * owners are (a) membership in the beam vector, (b) each live child's
 * `parent`.  `children[c]` is the weak next_data entry: it is cleared when the
 * child dies, and a live-but-pruned child is still found (quirk Q7). */
typedef struct Prefix {
  double pb, pnb, prev_pb, prev_pnb;
  int last_char;
  double lm_score, lm_score_before_last;
  int num_words, num_oov, num_oov_before_last;
  int* last_word; int last_word_len;
  lm_state st_before_last, st;
  struct Prefix* parent;
  struct Prefix** children;   /* V weak slots, lazily allocated */
  int refs;
} Prefix;

typedef struct {
  const double* lp; int64_t sB, sT, sV; const int64_t* x_len;
  int B, T, V, blank, W; const char* const* labels; int space_id;
  const oracle_lm* lm; int case_sensitive; double lmwt, wip, oov;
  int64_t* out; int64_t max_out; int64_t* out_len; int status;
} beam_ctx;

static Prefix* prefix_new(void) {
  Prefix* p = (Prefix*)calloc(1, sizeof(Prefix));
  p->pb = p->pnb = p->prev_pb = p->prev_pnb = NEG_INF;
  p->last_char = -1;
  return p;
}

static void prefix_release(Prefix* p, int V) {
  while (p && --p->refs == 0) {
    Prefix* par = p->parent;
    if (par && par->children) par->children[p->last_char] = NULL; /* weak_ptr expires */
    (void)V;
    free(p->children); free(p->last_word); free(p);
    p = par;
  }
}

/* get_prev_full_prob, .cpp:333-335 */
static double prev_full(const Prefix* p) { return LSE(p->prev_pnb, p->prev_pb); }

/* get_prev_full_prob_with_lmwt, .cpp:314-318 */
static double score_of(const beam_ctx* c, const Prefix* p) {
  return prev_full(p) + p->lm_score * c->lmwt - p->num_words * c->wip + p->num_oov * c->oov;
}

/* get_idx(vector<int>), .cpp:77-88: concatenate labels, (lower-case unless
 * case_sensitive), vocabulary lookup, miss -> 0. */
static uint32_t word_idx(const beam_ctx* c, const int* chars, int n) {
  char buf[1024]; size_t o = 0;
  for (int i = 0; i < n; i++) {
    const char* s = c->labels[chars[i]];
    size_t l = strlen(s);
    if (o + l + 1 >= sizeof(buf)) return 0;
    memcpy(buf + o, s, l); o += l;
  }
  buf[o] = 0;
  if (c->case_sensitive) return oracle_lm_word_index(c->lm, buf);
  for (size_t i = 0; i < o; i++) buf[i] = (char)tolower((unsigned char)buf[i]);
  vocab_slot* sl = table_find(c->lm->ltab, c->lm->vcap, buf, 0);
  return sl ? sl->id : 0;
}

/* get_next_prefix, .cpp:247-312.  Returns the child and sets *is_new. */
static Prefix* next_prefix(const beam_ctx* c, Prefix* p, int ch, int* is_new) {
  if (p->children && p->children[ch]) { *is_new = 0; return p->children[ch]; }   /* :250-252 */
  Prefix* n = prefix_new();
  if (!p->children) p->children = (Prefix**)calloc((size_t)c->V, sizeof(Prefix*));
  p->children[ch] = n;
  n->last_char = ch;
  n->num_words = p->num_words;
  int new_word = ch != c->space_id && (p->num_words == 0 || p->last_char == c->space_id); /* :258-259 */
  if (new_word) n->num_words++;
  if (c->lm) {
    const double kLogE10 = log(10.0);
    if (new_word) {                                                   /* :265-281 */
      n->last_word = (int*)malloc(sizeof(int)); n->last_word[0] = ch; n->last_word_len = 1;
      uint32_t wi = word_idx(c, n->last_word, 1);
      n->st_before_last = p->st; n->lm_score_before_last = p->lm_score;
      double s = oracle_lm_base_score(c->lm, n->st_before_last.w, n->st_before_last.n, wi, n->st.w, &n->st.n);
      n->lm_score += n->lm_score_before_last + s / kLogE10;           /* Q8: divides by ln10 */
      n->num_oov_before_last = p->num_oov;
      n->num_oov = p->num_oov + (wi == 0);
    } else if (ch != c->space_id) {                                   /* :282-297 */
      n->last_word_len = p->last_word_len + 1;
      n->last_word = (int*)malloc(sizeof(int) * (size_t)n->last_word_len);
      if (p->last_word_len) memcpy(n->last_word, p->last_word, sizeof(int) * (size_t)p->last_word_len);
      n->last_word[p->last_word_len] = ch;
      uint32_t wi = word_idx(c, n->last_word, n->last_word_len);
      n->st_before_last = p->st_before_last; n->lm_score_before_last = p->lm_score_before_last;
      double s = oracle_lm_base_score(c->lm, n->st_before_last.w, n->st_before_last.n, wi, n->st.w, &n->st.n);
      n->lm_score += n->lm_score_before_last + s / kLogE10;
      n->num_oov_before_last = p->num_oov_before_last;
      n->num_oov = n->num_oov_before_last + (wi == 0);
    } else {                                                          /* :299-307 */
      n->last_word_len = p->last_word_len;
      if (p->last_word_len) {
        n->last_word = (int*)malloc(sizeof(int) * (size_t)p->last_word_len);
        memcpy(n->last_word, p->last_word, sizeof(int) * (size_t)p->last_word_len);
      }
      n->lm_score = p->lm_score; n->lm_score_before_last = p->lm_score_before_last;
      n->num_oov = p->num_oov; n->num_oov_before_last = p->num_oov_before_last;
      n->st = p->st; n->st_before_last = p->st_before_last;
    }
  }
  n->parent = p; p->refs++;                                           /* :310 */
  *is_new = 1;
  return n;
}

typedef struct { Prefix* p; double score; size_t pos; } ranked;

/* score descending; ties by position in the pre-selection vector (the
 * reference's nth_element / sort leave ties unspecified, quirk Q9). */
static int ranked_cmp(const void* a_, const void* b_) {
  const ranked* a = (const ranked*)a_; const ranked* b = (const ranked*)b_;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return a->pos < b->pos ? -1 : (a->pos > b->pos ? 1 : 0);
}

/* decode_sentence, .cpp:353-441 */
static void beam_one(void* c_, int b) {
  beam_ctx* c = (beam_ctx*)c_;
  const int V = c->V, W = c->W, len = (int)c->x_len[b];
  const double* lp = c->lp + (int64_t)b * c->sB;
  /* the beam holds <= W prefixes at the start of every step (it starts at 1
   * and is cut back to W whenever it exceeds W), so one step adds <= W*(V-1) */
  size_t cap = (size_t)W * (size_t)V + (size_t)W + 16, n = 0, n_new = 0;
  Prefix** beam = (Prefix**)malloc(sizeof(Prefix*) * cap);
  Prefix** fresh = (Prefix**)malloc(sizeof(Prefix*) * cap);
  ranked* rk = (ranked*)malloc(sizeof(ranked) * cap);

  Prefix* root = prefix_new();                        /* get_initial_prefix, :222-230 */
  root->prev_pb = 0.0;
  if (c->lm) {
    root->st.n = 1; root->st.w[0] = c->lm->bos;
    root->st_before_last = root->st;
  }
  root->refs = 1; beam[n++] = root;

  for (int t = 0; t < len; t++) {
    const double* row = lp + (int64_t)t * c->sT;
    for (int ch = 0; ch < V; ch++) {                  /* char outer, prefix inner, :370-395 */
      double cur = row[(int64_t)ch * c->sV];
      for (size_t i = 0; i < n; i++) {
        Prefix* p = beam[i];
        if (ch == c->blank) {
          p->pb = LSE(p->pb, cur + prev_full(p));
        } else {
          int is_new; Prefix* q = next_prefix(c, p, ch, &is_new);
          if (is_new) { q->refs++; fresh[n_new++] = q; }
          if (ch == p->last_char) {
            q->pnb = LSE(q->pnb, cur + p->prev_pb);
            p->pnb = LSE(p->pnb, cur + p->prev_pnb);
          } else {
            q->pnb = LSE(q->pnb, cur + prev_full(p));
          }
        }
      }
    }
    memcpy(beam + n, fresh, sizeof(Prefix*) * n_new); n += n_new; n_new = 0;   /* :397-401 */
    for (size_t i = 0; i < n; i++) {                  /* next_step, :337-342 */
      Prefix* p = beam[i];
      p->prev_pb = p->pb; p->prev_pnb = p->pnb; p->pb = NEG_INF; p->pnb = NEG_INF;
    }
    if (n > (size_t)W) {                              /* :405-415 */
      for (size_t i = 0; i < n; i++) { rk[i].p = beam[i]; rk[i].score = score_of(c, beam[i]); rk[i].pos = i; }
      qsort(rk, n, sizeof(ranked), ranked_cmp);
      for (size_t i = 0; i < (size_t)W; i++) beam[i] = rk[i].p;
      for (size_t i = (size_t)W; i < n; i++) prefix_release(rk[i].p, V);
      n = (size_t)W;
    }
  }
  /* final sort, :418-424; take [0] */
  for (size_t i = 0; i < n; i++) { rk[i].p = beam[i]; rk[i].score = score_of(c, beam[i]); rk[i].pos = i; }
  qsort(rk, n, sizeof(ranked), ranked_cmp);
  Prefix* best = rk[0].p;
  /* get_sentence, :232-245: own last_char, then every ancestor's except the root's */
  int64_t m = 0;
  for (Prefix* q = best; q; q = q->parent) if (q == best || q->parent) m++;
  int64_t* o = c->out + (int64_t)b * c->max_out;
  int64_t idx = m;
  for (Prefix* q = best; q; q = q->parent)
    if (q == best || q->parent) { idx--; if (idx < c->max_out) o[idx] = q->last_char; }
  if (m > c->max_out) c->status = -2;
  c->out_len[b] = m;
  for (size_t i = 0; i < n; i++) prefix_release(beam[i], V);
  free(beam); free(fresh); free(rk);
}

int oracle_ctc_beam(const double* lp, int64_t sB, int64_t sT, int64_t sV,
                    const int64_t* x_len, int B, int T, int V, int blank,
                    int beam_width, const char* const* labels, int space_id,
                    const oracle_lm* lm, int case_sensitive,
                    double lmwt, double wip, double oov_penalty,
                    int64_t* out, int64_t max_out, int64_t* out_len,
                    int n_threads) {
  if (!lp || !x_len || !out || !out_len || B < 0 || T < 1 || V < 1 || beam_width < 1 || max_out < 1) return -1;
  if (lm && !labels) return -1;
  memset(out, 0, sizeof(int64_t) * (size_t)B * (size_t)max_out);
  beam_ctx c = {lp, sB, sT, sV, x_len, B, T, V, blank, beam_width, labels, space_id,
                lm, case_sensitive, lm ? lmwt : 0.0 /* .cpp:72-74 */, wip, oov_penalty,
                out, max_out, out_len, 0};
  run_pool(beam_one, &c, B, n_threads);
  return c.status;
}

/* ------------------------------------------------------------------ */
/* Viterbi forced alignment                                            */
/* pytorch_end2end/utils/alignment.py:50-106 (_get_alignment_ctc_1d), :10-47
 * (_get_alignment_asg_1d), batch driver :109-138 (get_alignment_3d).
 * Restated statement by statement, including what the Python leaves implicit:
 * path_alpha is zero-initialised (cells outside the band point at cell 0),
 * comparisons are strict ">" in the order stay, i-1, i-2, the skip needs
 * "i - 2 > 0" (:86), alpha is float64 whatever the input, the blank id is the
 * argument here (upstream hard-codes 0, :57). */
typedef struct {
  const double* lp; int64_t sB, sT, sV; const int64_t* targets; int64_t tgt_stride;
  const int64_t* x_len; const int64_t* t_len; int B, T, V, blank, is_ctc; int64_t* out;
} align_ctx;

static void align_one(void* vctx, int b) {
  align_ctx* c = (align_ctx*)vctx;
  const int T = (int)c->x_len[b], S = (int)c->t_len[b];
  const double* lp = c->lp + b * c->sB;
  const int64_t* tg = c->targets + (int64_t)b * c->tgt_stride;
  int64_t* best = c->out + (int64_t)b * c->T;
  if (T < 1) return;
#define LP(t, v) lp[(int64_t)(t) * c->sT + (int64_t)(v) * c->sV]
  if (c->is_ctc) {
    const int L = 2 * S + 1;
    for (int k = 0; k < T; k++) best[k] = 0;                        /* np.zeros, :63 */
    if (L == 1 || T == 1) {                                         /* :65-70 */
      if (L == 1) return;
      best[0] = tg[0];
      return;
    }
    int64_t* ext = (int64_t*)malloc(sizeof(int64_t) * (size_t)L);
    for (int i = 0; i < L; i++) ext[i] = (i & 1) ? tg[i >> 1] : c->blank;
    double* alpha = (double*)malloc(sizeof(double) * (size_t)L * (size_t)T);
    int* path = (int*)calloc((size_t)L * (size_t)T, sizeof(int));
    for (size_t q = 0; q < (size_t)L * (size_t)T; q++) alpha[q] = NEG_INF;
#define A(i, k) alpha[(size_t)(i) * (size_t)T + (size_t)(k)]
#define P(i, k) path[(size_t)(i) * (size_t)T + (size_t)(k)]
    A(0, 0) = LP(0, ext[0]);
    A(1, 0) = LP(0, ext[1]);
    for (int k = 1; k < T; k++) {
      int start = L - 2 * (T - k); if (start < 0) start = 0;
      int end = k * 2 + 2; if (end > L) end = L;
      for (int i = start; i < end; i++) { A(i, k) = A(i, k - 1); P(i, k) = i; }
      for (int i = start; i < end; i++) {
        const int64_t lab = ext[i];
        if (i > 0) {
          if (A(i - 1, k - 1) > A(i, k)) { A(i, k) = A(i - 1, k - 1); P(i, k) = i - 1; }
          if (lab != c->blank && i - 2 > 0 && ext[i - 2] != lab && A(i - 2, k - 1) > A(i, k)) {
            A(i, k) = A(i - 2, k - 1); P(i, k) = i - 2;
          }
        }
        A(i, k) += LP(k, lab);
      }
    }
    int i = L - 1;
    if (A(i - 1, T - 1) > A(i, T - 1)) i = i - 1;
    for (int k = T - 1; k >= 0; k--) { best[k] = ext[i]; i = P(i, k); }
    free(ext); free(alpha); free(path);
  } else {
    for (int k = 0; k < T; k++) best[k] = 0;
    if (S < 1) return;                                              /* (upstream would index targets[0]) */
    if (T == 1) { best[0] = tg[0]; return; }                        /* :22-24 */
    double* alpha = (double*)malloc(sizeof(double) * (size_t)S * (size_t)T);
    int* path = (int*)calloc((size_t)S * (size_t)T, sizeof(int));
    for (size_t q = 0; q < (size_t)S * (size_t)T; q++) alpha[q] = NEG_INF;
#define A2(i, k) alpha[(size_t)(i) * (size_t)T + (size_t)(k)]
#define P2(i, k) path[(size_t)(i) * (size_t)T + (size_t)(k)]
    A2(0, 0) = LP(0, tg[0]);
    for (int k = 1; k < T; k++) {
      int start = S - (T - k); if (start < 0) start = 0;
      int end = k + 1; if (end > S) end = S;
      for (int i = start; i < end; i++) { A2(i, k) = A2(i, k - 1); P2(i, k) = i; }
      for (int i = start; i < end; i++) {
        if (i > 0 && A2(i - 1, k - 1) > A2(i, k)) { A2(i, k) = A2(i - 1, k - 1); P2(i, k) = i - 1; }
        A2(i, k) += LP(k, tg[i]);
      }
    }
    int i = S - 1;
    for (int k = T - 1; k >= 0; k--) { best[k] = tg[i]; i = P2(i, k); }
    free(alpha); free(path);
  }
}

/* out: (B,T) int64, pre-filled by the caller (upstream: -100, :132); rows b get out[b, :x_len[b]] */
int oracle_ctc_align(const double* lp, int64_t sB, int64_t sT, int64_t sV,
                     const int64_t* targets, int64_t tgt_stride, const int64_t* x_len, const int64_t* t_len,
                     int B, int T, int V, int blank, int is_ctc, int64_t* out, int n_threads) {
  if (!lp || !x_len || !t_len || !out || B < 0 || T < 1 || V < 1) return -1;
  align_ctx c = {lp, sB, sT, sV, targets, tgt_stride, x_len, t_len, B, T, V, blank, is_ctc, out};
  run_pool(align_one, &c, B, n_threads);
  return 0;
}

// Prefix beam search (Hannun et al. 2014) with optional n-gram LM scoring, on the GPU.
// Restates src/decoders/ctc_decoder.cpp:153-201 (driver), :353-441 (decode_sentence), :247-312
// (get_next_prefix), :314-318 (score), :232-245 (get_sentence).
//
// One 256-thread workgroup per utterance; time is serial, the W*V (prefix, character) pairs of a step are
// spread over the threads.  The prefix tree lives in a per-utterance node pool in HBM (L2 resident); the
// reference's shared_ptr / weak_ptr ownership is restated as explicit reference counts:
//   * a node is owned by its membership in the beam and by each live child (`parent` pointer);
//   * the weak `next_data` map of a prefix is a V-entry child table that exists only while the prefix is in the
//     beam (only beam members are ever asked for a child); a dying child clears its entry -- so a pruned child that
//     is kept alive by a descendant is still found, receives probability, and is NOT re-added (quirk Q7).
// All scores are IEEE doubles with the reference's two-argument log-sum-exp.  Top-W selection is a full bitonic
// sort in LDS on (score descending, position ascending): the reference's nth_element leaves ties unspecified
// (quirk Q9); the oracle uses the same total order.
//
// The language model stands where KenLM stands upstream (src/decoders/ctc_decoder.cpp:60-71,77-88,264-308):
// host-side ARPA reader (plain or gzip), device-resident open-addressing tables, standard back-off scoring.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace e2e {

constexpr int kLmMaxOrder = 6;     // KENLM_MAX_ORDER=6, CMakeLists.txt:36
constexpr int kCtx = kLmMaxOrder - 1;

struct NgSlot { uint32_t ids[kLmMaxOrder]; int32_t n; float prob; float backoff; };

struct LmView {                    // what the kernel sees (device pointers) / what the host scorer sees
  int order;
  const uint64_t* vkeys; const uint32_t* vvals; uint32_t vmask;
  const NgSlot* ng; uint32_t ngmask;
  uint32_t bos;
  const unsigned char* label_bytes; const int* label_off;   // label c spells bytes [off[c], off[c+1])
  int fold_case;
};

__host__ __device__ inline uint64_t fnv_step(uint64_t h, unsigned char b) { return (h ^ b) * 1099511628211ULL; }
constexpr uint64_t kFnvInit = 1469598103934665603ULL;

__host__ __device__ inline uint64_t ngram_hash(const uint32_t* ids, int n) {
  uint64_t h = kFnvInit ^ (uint64_t)n;
  for (int i = 0; i < n; i++)
    for (int b = 0; b < 4; b++) h = fnv_step(h, (unsigned char)(ids[i] >> (8 * b)));
  return h;
}

__host__ __device__ inline uint32_t lm_word_lookup(const LmView& lm, uint64_t h) {
  if (h == 0) h = 1;
  for (uint32_t i = (uint32_t)h & lm.vmask;; i = (i + 1) & lm.vmask) {
    const uint64_t k = lm.vkeys[i];
    if (k == h) return lm.vvals[i];
    if (k == 0) return 0;                       // NotFound() == <unk> == 0
  }
}

__host__ __device__ inline const NgSlot* lm_ngram_find(const LmView& lm, const uint32_t* ids, int n) {
  for (uint32_t i = (uint32_t)ngram_hash(ids, n) & lm.ngmask;; i = (i + 1) & lm.ngmask) {
    const NgSlot* s = &lm.ng[i];
    if (s->n == 0) return nullptr;
    if (s->n == n) {
      bool eq = true;
      for (int k = 0; k < n; k++) eq = eq && s->ids[k] == ids[k];
      if (eq) return s;
    }
  }
}

// log10 p(word | ctx) with ARPA back-off; ctx is most-recent-first.  Float accumulation like KenLM.
__host__ __device__ inline float lm_base_score(const LmView& lm, const uint32_t* ctx, int ctx_len, uint32_t word,
                                               uint32_t* out_ctx, int* out_len) {
  int n = ctx_len; if (n > lm.order - 1) n = lm.order - 1;
  uint32_t ids[kLmMaxOrder];
  float acc = 0.f, result = 0.f; bool found = false;
  for (int k = n; k >= 0 && !found; k--) {
    for (int i = 0; i < k; i++) ids[i] = ctx[k - 1 - i];
    ids[k] = word;
    const NgSlot* s = lm_ngram_find(lm, ids, k + 1);
    if (s) { result = acc + s->prob; found = true; break; }
    if (k > 0) { const NgSlot* c = lm_ngram_find(lm, ids, k); if (c) acc += c->backoff; }
  }
  if (!found) { const uint32_t z = 0; const NgSlot* u = lm_ngram_find(lm, &z, 1); result = acc + (u ? u->prob : -100.f); }
  if (out_ctx) {
    int m = n + 1; if (m > lm.order - 1) m = lm.order - 1;
    uint32_t tmp[kLmMaxOrder];
    if (m > 0) tmp[0] = word;
    for (int i = 1; i < m; i++) tmp[i] = ctx[i - 1];
    for (int i = 0; i < m; i++) out_ctx[i] = tmp[i];
    *out_len = m;
  }
  return result;
}

}  // namespace e2e

// ------------------------------------------------------------------------------------------------------
// host side of the LM
// ------------------------------------------------------------------------------------------------------
struct e2e_lm {
  int order = 0;
  int fold_case = 0;
  std::vector<uint64_t> vkeys; std::vector<uint32_t> vvals;
  std::vector<e2e::NgSlot> ng;
  std::vector<unsigned char> label_bytes; std::vector<int> label_off;
  std::unordered_map<std::string, uint32_t> exact;       // word -> id, exact case (GetVocabulary().Index)
  uint32_t bos = 0;
  // device copies
  uint64_t* d_vkeys = nullptr; uint32_t* d_vvals = nullptr; e2e::NgSlot* d_ng = nullptr;
  unsigned char* d_label_bytes = nullptr; int* d_label_off = nullptr;
  e2e::LmView host_view() const {
    return {order, vkeys.data(), vvals.data(), (uint32_t)vkeys.size() - 1, ng.data(), (uint32_t)ng.size() - 1, bos,
            label_bytes.data(), label_off.data(), fold_case};
  }
  e2e::LmView dev_view() const {
    return {order, d_vkeys, d_vvals, (uint32_t)vkeys.size() - 1, d_ng, (uint32_t)ng.size() - 1, bos,
            d_label_bytes, d_label_off, fold_case};
  }
};

namespace e2e {
namespace {

uint64_t word_hash(const std::string& w) {
  uint64_t h = kFnvInit;
  for (unsigned char c : w) h = fnv_step(h, c);
  return h == 0 ? 1 : h;
}

size_t pow2_at_least(size_t n) { size_t p = 16; while (p < n) p <<= 1; return p; }

std::string lower(const std::string& s) {
  std::string r = s;
  for (auto& c : r) c = (char)::tolower((unsigned char)c);      // str_to_lower, ctc_decoder.cpp:32-36
  return r;
}

}  // namespace
}  // namespace e2e

using namespace e2e;

extern "C" int e2e_lm_load_arpa(const char* path, const char* const* labels, int V, int case_sensitive, e2e_lm** out) {
  if (out) *out = nullptr;
  if (!path || !out || V < 0 || (V > 0 && !labels)) { set_error("e2e_lm_load_arpa: bad argument"); return E2E_ERR_ARG; }
  gzFile f = gzopen(path, "rb");
  if (!f) { set_error("cannot open language model %s", path); return E2E_ERR_IO; }
  e2e_lm* lm = new e2e_lm();
  lm->fold_case = case_sensitive ? 0 : 1;
  std::vector<std::string> words;                          // id -> word; id 0 is <unk>
  auto intern = [&](const std::string& w) -> uint32_t {
    auto it = lm->exact.find(w);
    if (it != lm->exact.end()) return it->second;
    const uint32_t id = (uint32_t)words.size();
    words.push_back(w); lm->exact.emplace(w, id);
    return id;
  };
  intern("<unk>");
  struct Entry { uint32_t ids[kLmMaxOrder]; int n; float prob, bo; };
  std::vector<Entry> entries;
  std::vector<char> buf(1 << 16);
  int section = 0; bool saw_data = false;
  while (gzgets(f, buf.data(), (int)buf.size())) {
    char* line = buf.data();
    size_t len = strlen(line);
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    if (len == 0) continue;
    if (line[0] == '\\') {
      int k;
      if (strncmp(line, "\\data\\", 6) == 0) saw_data = true;
      else if (sscanf(line, "\\%d-grams:", &k) == 1) { section = k; if (k > lm->order) lm->order = k; }
      else if (strncmp(line, "\\end\\", 5) == 0) break;
      continue;
    }
    if (section == 0 || section > kLmMaxOrder) continue;
    char* save = nullptr;
    char* tok = strtok_r(line, " \t", &save);
    if (!tok) continue;
    Entry e; e.n = section; e.prob = strtof(tok, nullptr); e.bo = 0.f;
    bool ok = true;
    for (int i = 0; i < section; i++) { tok = strtok_r(nullptr, " \t", &save); if (!tok) { ok = false; break; } e.ids[i] = intern(tok); }
    if (!ok) continue;
    tok = strtok_r(nullptr, " \t", &save);
    if (tok) e.bo = strtof(tok, nullptr);
    entries.push_back(e);
  }
  gzclose(f);
  if (!saw_data || lm->order == 0) { delete lm; set_error("%s: not an ARPA file (no \\data\\ / n-gram sections)", path); return E2E_ERR_IO; }
  if (lm->order > kLmMaxOrder) { delete lm; set_error("%s: order %d > %d", path, lm->order, kLmMaxOrder); return E2E_ERR_UNSUPPORTED; }
  {  // <unk> absent from the file: KenLM's default unknown_missing_logprob = -100
    bool has_unk = false;
    for (const auto& e : entries) if (e.n == 1 && e.ids[0] == 0) { has_unk = true; break; }
    if (!has_unk) { Entry e; e.n = 1; e.ids[0] = 0; e.prob = -100.f; e.bo = 0.f; entries.push_back(e); }
  }
  // n-gram table
  lm->ng.assign(pow2_at_least(entries.size() * 2 + 16), NgSlot{{0, 0, 0, 0, 0, 0}, 0, 0.f, 0.f});
  const uint32_t ngmask = (uint32_t)lm->ng.size() - 1;
  for (const auto& e : entries) {
    uint32_t i = (uint32_t)ngram_hash(e.ids, e.n) & ngmask;
    for (;; i = (i + 1) & ngmask) {
      NgSlot& s = lm->ng[i];
      if (s.n == 0) { s.n = e.n; for (int k = 0; k < e.n; k++) s.ids[k] = e.ids[k]; s.prob = e.prob; s.backoff = e.bo; break; }
      if (s.n == e.n && memcmp(s.ids, e.ids, sizeof(uint32_t) * e.n) == 0) { s.prob = e.prob; s.backoff = e.bo; break; }
    }
  }
  // vocabulary table keyed by the hash of the (optionally lower-cased) spelling; when two words fold to the same
  // string the reference keeps whichever its unordered_map iteration visits last (unspecified) -- here: lowest id
  lm->vkeys.assign(pow2_at_least(words.size() * 2 + 16), 0);
  lm->vvals.assign(lm->vkeys.size(), 0);
  const uint32_t vmask = (uint32_t)lm->vkeys.size() - 1;
  for (uint32_t id = 0; id < words.size(); id++) {
    const uint64_t h = word_hash(lm->fold_case ? lower(words[id]) : words[id]);
    for (uint32_t i = (uint32_t)h & vmask;; i = (i + 1) & vmask) {
      if (lm->vkeys[i] == h) break;
      if (lm->vkeys[i] == 0) { lm->vkeys[i] = h; lm->vvals[i] = id; break; }
    }
  }
  { auto it = lm->exact.find("<s>"); lm->bos = it != lm->exact.end() ? it->second : 0; }
  lm->label_off.assign(1, 0);
  for (int c = 0; c < V; c++) {
    for (const char* s = labels[c]; *s; s++) lm->label_bytes.push_back((unsigned char)*s);
    lm->label_off.push_back((int)lm->label_bytes.size());
  }
  if (lm->label_bytes.empty()) lm->label_bytes.push_back(0);
  // upload
  auto up = [](void** d, const void* h, size_t bytes) -> bool {
    if (hipMalloc(d, bytes) != hipSuccess) return false;
    return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = up((void**)&lm->d_vkeys, lm->vkeys.data(), lm->vkeys.size() * sizeof(uint64_t)) &&
            up((void**)&lm->d_vvals, lm->vvals.data(), lm->vvals.size() * sizeof(uint32_t)) &&
            up((void**)&lm->d_ng, lm->ng.data(), lm->ng.size() * sizeof(NgSlot)) &&
            up((void**)&lm->d_label_bytes, lm->label_bytes.data(), lm->label_bytes.size()) &&
            up((void**)&lm->d_label_off, lm->label_off.data(), lm->label_off.size() * sizeof(int));
  if (!ok) {
    // no usable GPU: keep the host tables (e2e_lm_word_index / e2e_lm_score still work); e2e_ctc_beam refuses it
    (void)hipGetLastError();
    (void)hipFree(lm->d_vkeys); (void)hipFree(lm->d_vvals); (void)hipFree(lm->d_ng);
    (void)hipFree(lm->d_label_bytes); (void)hipFree(lm->d_label_off);
    lm->d_vkeys = nullptr; lm->d_vvals = nullptr; lm->d_ng = nullptr; lm->d_label_bytes = nullptr; lm->d_label_off = nullptr;
  }
  *out = lm;
  return E2E_OK;
}

extern "C" void e2e_lm_free(e2e_lm* lm) {
  if (!lm) return;
  (void)hipFree(lm->d_vkeys); (void)hipFree(lm->d_vvals); (void)hipFree(lm->d_ng);
  (void)hipFree(lm->d_label_bytes); (void)hipFree(lm->d_label_off);
  delete lm;
}

extern "C" int e2e_lm_order(const e2e_lm* lm) { return lm ? lm->order : 0; }

// get_idx(string), ctc_decoder.cpp:77-82: exact lookup when case sensitive, else lower-cased lookup
extern "C" uint32_t e2e_lm_word_index(const e2e_lm* lm, const char* word) {
  if (!lm || !word) return 0;
  const std::string w = lm->fold_case ? lower(word) : std::string(word);
  return lm_word_lookup(lm->host_view(), word_hash(w));
}

extern "C" double e2e_lm_score(const e2e_lm* lm, const uint32_t* ctx, int ctx_len, uint32_t word) {
  if (!lm || ctx_len < 0 || ctx_len > kCtx || (ctx_len > 0 && !ctx)) return 0.0;
  return (double)lm_base_score(lm->host_view(), ctx, ctx_len, word, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------------
namespace e2e {
namespace {

constexpr int kThreads = 1024;
constexpr int kMaxCand = 8192;     // W*V + W must fit the LDS sort

struct BeamNode {
  double pb, pnb, ppb, ppnb;               // cur / prev log-probs ending in blank / non-blank
  double lm_score, lm_before;
  int parent, last_char, refs, tab;
  int num_words, num_oov, num_oov_before, word_len;
  unsigned long long word_hash;            // hash of the spelled last word (get_idx(vector<int>), :84-88)
  unsigned int st[kCtx], stb[kCtx];        // LM context after / before the last word, most recent first
  int st_n, stb_n;
};

struct BeamParams {
  const void* lp; int64_t sB, sT, sV; const int64_t* x_len;
  int B, T, V, blank, W, space_id;
  int has_lm; LmView lm; double lmwt, wip, oov;
  int64_t* out; int64_t max_out; int64_t* out_len;
  // per-utterance workspace
  BeamNode* nodes; int* free_nodes; int* ctab; int* free_tabs; int* cand; int* cand2; int* status;
  int NCAP, TCAP, CMAX, NP2;
};

__device__ __forceinline__ double ninf() { return -__builtin_huge_val(); }

// src/utils/math_utils.h:8-16
__device__ __forceinline__ double lse2(double a, double b) {
  if (a == ninf()) return b;
  if (b == ninf()) return a;
  if (a > b) return a + log(1.0 + exp(b - a));
  return b + log(1.0 + exp(a - b));
}

// (score desc, position asc): does a come before b?
__device__ __forceinline__ bool before(double sa, int ia, double sb, int ib) {
  return sa > sb || (sa == sb && ia < ib);
}

__device__ void bitonic_sort(double* key, int* idx, int n2, int tid) {
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = tid; e < n2; e += kThreads) {
        const int partner = e ^ j;
        if (partner > e) {
          const bool up = (e & k) == 0;           // this run ends up "before-ordered"
          const double ka = key[e], kb = key[partner];
          const int ia = idx[e], ib = idx[partner];
          const bool a_first = before(ka, ia, kb, ib);
          if (up ? !a_first : a_first) { key[e] = kb; key[partner] = ka; idx[e] = ib; idx[partner] = ia; }
        }
      }
      __syncthreads();
    }
  }
}

template <typename IO>
__global__ __launch_bounds__(kThreads) void ctc_beam_kernel(BeamParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* skey = reinterpret_cast<double*>(smem);          // [NP2]
  double* srow = skey + p.NP2;                             // [V]
  int* sidx = reinterpret_cast<int*>(srow + p.V);          // [NP2]
  __shared__ int s_free_nodes, s_free_tabs, s_new, s_err;
  __shared__ int s_part[kThreads];

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V, W = p.W, blank = p.blank, space = p.space_id;
  BeamNode* nodes = p.nodes + (size_t)b * p.NCAP;
  int* free_nodes = p.free_nodes + (size_t)b * p.NCAP;
  int* ctab = p.ctab + (size_t)b * p.TCAP * V;
  int* free_tabs = p.free_tabs + (size_t)b * p.TCAP;
  int* cand = p.cand + (size_t)b * p.CMAX;
  int* cand2 = p.cand2 + (size_t)b * p.CMAX;
  const IO* lp = reinterpret_cast<const IO*>(p.lp) + (int64_t)b * p.sB;
  int64_t Tq = p.x_len[b];
  const int T = Tq < 0 ? 0 : (Tq > p.T ? p.T : (int)Tq);
  const double kLogE10 = log(10.0);

  // ---- pools, root prefix (get_initial_prefix, :222-230) ----
  for (int i = tid; i < p.NCAP; i += kThreads) free_nodes[i] = p.NCAP - 1 - i;     // pop order 0,1,2,...
  for (int i = tid; i < p.TCAP; i += kThreads) free_tabs[i] = p.TCAP - 1 - i;
  if (tid == 0) { s_free_nodes = p.NCAP - 1; s_free_tabs = p.TCAP - 1; s_err = 0; }   // node 0 / table 0 = root's
  for (int c = tid; c < V; c += kThreads) ctab[c] = -1;
  if (tid == 0) {
    BeamNode& r = nodes[0];
    r.pb = ninf(); r.pnb = ninf(); r.ppb = 0.0; r.ppnb = ninf();
    r.lm_score = 0.0; r.lm_before = 0.0;
    r.parent = -1; r.last_char = -1; r.refs = 1; r.tab = 0;
    r.num_words = 0; r.num_oov = 0; r.num_oov_before = 0; r.word_len = 0; r.word_hash = kFnvInit;
    r.st_n = 0; r.stb_n = 0;
    if (p.has_lm) { r.st[0] = p.lm.bos; r.st_n = 1; r.stb[0] = p.lm.bos; r.stb_n = 1; }
    cand[0] = 0;
  }
  __syncthreads();
  int n = 1;

  for (int t = 0; t < T; t++) {
    for (int c = tid; c < V; c += kThreads) srow[c] = (double)lp[(int64_t)t * p.sT + (int64_t)c * p.sV];
    if (tid == 0) s_new = 0;
    __syncthreads();
    // pairs in the reference's order: character outer, prefix inner (:370-395): q = c*n + i
    const int npairs = n * V;
    const int chunk = (npairs + kThreads - 1) / kThreads;
    const int q0 = min(tid * chunk, npairs), q1 = min(q0 + chunk, npairs);
    // pass 1: which pairs create a prefix?  (weak child lookup, :250-252)
    int my_new = 0;
    for (int q = q0; q < q1; q++) {
      const int c = q / n, i = q - c * n;
      if (c == blank) continue;
      const BeamNode& pr = nodes[cand[i]];
      if (ctab[pr.tab * V + c] < 0) my_new++;
    }
    // exclusive scan of my_new over the threads (pairs are chunked in order, so this is the reference's order)
    int incl = my_new;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) s_part[wid] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wid; w++) base += s_part[w];
    int total_new = 0;
    for (int w = 0; w < kThreads / 64; w++) total_new += s_part[w];
    int pos = n + base + incl - my_new;
    __syncthreads();
    // pass 2: blank and child updates; creation
    for (int q = q0; q < q1; q++) {
      const int c = q / n, i = q - c * n;
      const int pi = cand[i];
      BeamNode& pr = nodes[pi];
      const double cur = srow[c];
      if (c == blank) {                                                          // :374-376
        pr.pb = lse2(pr.pb, cur + lse2(pr.ppnb, pr.ppb));
        continue;
      }
      int k = ctab[pr.tab * V + c];
      if (k < 0) {                                                               // make_shared<Prefix>, :254-310
        const int fi = atomicSub(&s_free_nodes, 1) - 1;
        if (fi < 0) { s_err = 1; continue; }
        k = free_nodes[fi];
        BeamNode& nn = nodes[k];
        nn.pb = ninf(); nn.pnb = ninf(); nn.ppb = ninf(); nn.ppnb = ninf();
        nn.last_char = c; nn.parent = pi; nn.refs = 1; nn.tab = -1;
        const bool new_word = c != space && (pr.num_words == 0 || pr.last_char == space);     // :258-259
        nn.num_words = pr.num_words + (new_word ? 1 : 0);
        nn.lm_score = 0.0; nn.lm_before = 0.0; nn.num_oov = 0; nn.num_oov_before = 0;
        nn.word_len = 0; nn.word_hash = kFnvInit; nn.st_n = 0; nn.stb_n = 0;
        if (p.has_lm) {
          if (c != space) {
            uint64_t h = new_word ? kFnvInit : pr.word_hash;
            for (int bi = p.lm.label_off[c]; bi < p.lm.label_off[c + 1]; bi++) {
              unsigned char ch = p.lm.label_bytes[bi];
              if (p.lm.fold_case && ch >= 'A' && ch <= 'Z') ch += 32;
              h = fnv_step(h, ch);
            }
            nn.word_hash = h; nn.word_len = (new_word ? 0 : pr.word_len) + 1;
            const uint32_t wi = lm_word_lookup(p.lm, h);
            if (new_word) {                                                       // :265-281
              for (int s = 0; s < pr.st_n; s++) nn.stb[s] = pr.st[s];
              nn.stb_n = pr.st_n; nn.lm_before = pr.lm_score; nn.num_oov_before = pr.num_oov;
            } else {                                                              // :282-297
              for (int s = 0; s < pr.stb_n; s++) nn.stb[s] = pr.stb[s];
              nn.stb_n = pr.stb_n; nn.lm_before = pr.lm_before; nn.num_oov_before = pr.num_oov_before;
            }
            int on = 0;
            const float sc = lm_base_score(p.lm, nn.stb, nn.stb_n, wi, nn.st, &on);
            nn.st_n = on;
            nn.lm_score = nn.lm_before + (double)sc / kLogE10;                     // quirk Q8: divides by ln 10
            nn.num_oov = nn.num_oov_before + (wi == 0 ? 1 : 0);
          } else {                                                                // :299-307 copy
            nn.word_hash = pr.word_hash; nn.word_len = pr.word_len;
            nn.lm_score = pr.lm_score; nn.lm_before = pr.lm_before;
            nn.num_oov = pr.num_oov; nn.num_oov_before = pr.num_oov_before;
            for (int s = 0; s < pr.st_n; s++) nn.st[s] = pr.st[s];
            for (int s = 0; s < pr.stb_n; s++) nn.stb[s] = pr.stb[s];
            nn.st_n = pr.st_n; nn.stb_n = pr.stb_n;
          }
        }
        atomicAdd(&pr.refs, 1);
        ctab[pr.tab * V + c] = k;
        cand[pos++] = k;
      }
      BeamNode& ch = nodes[k];
      if (c == pr.last_char) ch.pnb = lse2(ch.pnb, cur + pr.ppb);                  // :383-385 (child part)
      else ch.pnb = lse2(ch.pnb, cur + lse2(pr.ppnb, pr.ppb));                    // :389-391
    }
    __threadfence_block();
    __syncthreads();
    // repeated character, the prefix's own share (:386-387); after the child updates because a prefix can be the
    // child of another beam member -- log_sum_exp is symmetric, so the order of the two updates does not matter
    for (int i = tid; i < n; i += kThreads) {
      BeamNode& pr = nodes[cand[i]];
      if (pr.last_char >= 0 && pr.last_char != blank) pr.pnb = lse2(pr.pnb, srow[pr.last_char] + pr.ppnb);
    }
    __threadfence_block();
    __syncthreads();
    const int ntot = n + total_new;
    // next_step (:337-342) on every member, old and new
    for (int i = tid; i < ntot; i += kThreads) {
      BeamNode& nd = nodes[cand[i]];
      nd.ppb = nd.pb; nd.ppnb = nd.pnb; nd.pb = ninf(); nd.pnb = ninf();
    }
    __threadfence_block();
    __syncthreads();
    if (ntot > W) {                                                              // :405-415
      int n2 = 1; while (n2 < ntot) n2 <<= 1;
      for (int i = tid; i < n2; i += kThreads) {
        if (i < ntot) {
          const BeamNode& nd = nodes[cand[i]];
          skey[i] = lse2(nd.ppnb, nd.ppb) + nd.lm_score * p.lmwt - nd.num_words * p.wip + nd.num_oov * p.oov;   // :314-318
        } else {
          skey[i] = ninf();
        }
        sidx[i] = i;
      }
      __syncthreads();
      bitonic_sort(skey, sidx, n2, tid);
      for (int i = tid; i < ntot; i += kThreads) cand2[i] = cand[sidx[i]];
      __threadfence_block();
      __syncthreads();
      // leaving the beam: give the child table back, drop the beam's reference, cascade (shared_ptr release)
      for (int i = W + tid; i < ntot; i += kThreads) {
        int k = cand2[i];
        if (nodes[k].tab >= 0) { free_tabs[atomicAdd(&s_free_tabs, 1)] = nodes[k].tab; nodes[k].tab = -1; }
        while (k >= 0) {
          if (atomicSub(&nodes[k].refs, 1) != 1) break;
          const int par = nodes[k].parent;
          if (par >= 0) { const int pt = nodes[par].tab; if (pt >= 0) ctab[pt * V + nodes[k].last_char] = -1; }   // weak_ptr expires
          free_nodes[atomicAdd(&s_free_nodes, 1)] = k;
          k = par;
        }
      }
      __threadfence_block();
      __syncthreads();
      for (int i = tid; i < W; i += kThreads) cand[i] = cand2[i];
      n = W;
    } else {
      n = ntot;
    }
    __syncthreads();
    // members without a child table (the new ones) get one
    for (int i = tid; i < n; i += kThreads) {
      BeamNode& nd = nodes[cand[i]];
      if (nd.tab < 0) {
        const int ti = atomicSub(&s_free_tabs, 1) - 1;
        if (ti < 0) { s_err = 2; continue; }
        nd.tab = free_tabs[ti];
        for (int c = 0; c < V; c++) ctab[nd.tab * V + c] = -1;
      }
    }
    __threadfence_block();
    __syncthreads();
    if (s_err) break;
  }

  // ---- final sort (:418-424), best prefix, its sentence (:232-245) ----
  {
    int n2 = 1; while (n2 < n) n2 <<= 1;
    for (int i = tid; i < n2; i += kThreads) {
      if (i < n) {
        const BeamNode& nd = nodes[cand[i]];
        skey[i] = lse2(nd.ppnb, nd.ppb) + nd.lm_score * p.lmwt - nd.num_words * p.wip + nd.num_oov * p.oov;
      } else skey[i] = ninf();
      sidx[i] = i;
    }
    __syncthreads();
    bitonic_sort(skey, sidx, n2, tid);
  }
  int64_t* out = p.out + (int64_t)b * p.max_out;
  for (int64_t i = tid; i < p.max_out; i += kThreads) out[i] = 0;
  __syncthreads();
  if (tid == 0) {
    const int best = cand[sidx[0]];
    int64_t m = 0;
    for (int k = best; k >= 0; k = nodes[k].parent) if (k == best || nodes[k].parent >= 0) m++;
    int64_t at = m;
    for (int k = best; k >= 0; k = nodes[k].parent)
      if (k == best || nodes[k].parent >= 0) { at--; if (at < p.max_out) out[at] = nodes[k].last_char; }
    p.out_len[b] = m;
    int st = s_err;
    if (m > p.max_out) st = 3;
    p.status[b] = st;
  }
}

struct BeamLayout { size_t nodes, free_nodes, ctab, free_tabs, cand, cand2, status, total; int NCAP, TCAP, CMAX, NP2; };

BeamLayout beam_layout(int B, int T, int V, int W) {
  BeamLayout l;
  l.CMAX = W * V + W + 8;
  // live nodes: the beam, its ancestors (at most one root path of length <= T per member) and one step's candidates
  l.NCAP = W * V + W * (T + 2) + 8;
  l.TCAP = 2 * W + 8;
  l.NP2 = 1; while (l.NP2 < l.CMAX) l.NP2 <<= 1;
  size_t o = 0;
  l.nodes = o; o += align_up((size_t)B * l.NCAP * sizeof(BeamNode), 256);
  l.free_nodes = o; o += align_up((size_t)B * l.NCAP * sizeof(int), 256);
  l.ctab = o; o += align_up((size_t)B * l.TCAP * V * sizeof(int), 256);
  l.free_tabs = o; o += align_up((size_t)B * l.TCAP * sizeof(int), 256);
  l.cand = o; o += align_up((size_t)B * l.CMAX * sizeof(int), 256);
  l.cand2 = o; o += align_up((size_t)B * l.CMAX * sizeof(int), 256);
  l.status = o; o += align_up((size_t)B * sizeof(int), 256);
  l.total = o;
  return l;
}

}  // namespace
}  // namespace e2e

extern "C" size_t e2e_ctc_beam_workspace_bytes(int B, int T, int V, int beam_width) {
  if (B < 0 || T < 1 || V < 1 || beam_width < 1) return 0;
  return beam_layout(B, T, V, beam_width).total + 256;
}

extern "C" int e2e_ctc_beam(const void* lp, int dtype, int64_t sB, int64_t sT, int64_t sV,
                            const int64_t* x_len, int B, int T, int V, int blank,
                            int beam_width, int space_id, const e2e_lm* lm,
                            double lmwt, double wip, double oov_penalty,
                            int64_t* out, int64_t max_out, int64_t* out_len,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64) { set_error("dtype must be E2E_F32 or E2E_F64"); return E2E_ERR_ARG; }
  if (B < 0 || T < 1 || V < 1 || beam_width < 1 || max_out < 1) { set_error("bad sizes"); return E2E_ERR_ARG; }
  if (blank < 0 || blank >= V) { set_error("blank=%d outside [0,%d)", blank, V); return E2E_ERR_ARG; }
  if (B > 0 && (!lp || !x_len || !out || !out_len)) { set_error("null pointer argument"); return E2E_ERR_ARG; }
  if (lm && !lm->d_ng) { set_error("the language model has no device tables (it was loaded without a GPU)"); return E2E_ERR_HIP; }
  const BeamLayout l = beam_layout(B, T, V, beam_width);
  if (l.CMAX > kMaxCand) {
    set_error("beam_width*alphabet = %d candidates per step exceed the %d the LDS sort holds", l.CMAX, kMaxCand);
    return E2E_ERR_UNSUPPORTED;
  }
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const uintptr_t aligned = (base + 255) & ~(uintptr_t)255;
  if (!workspace || workspace_bytes < l.total + (aligned - base)) { set_error("workspace too small: need %zu", l.total + 256); return E2E_ERR_WORKSPACE; }
  if (B == 0) return E2E_OK;
  char* ws = reinterpret_cast<char*>(aligned);
  BeamParams p;
  p.lp = lp; p.sB = sB; p.sT = sT; p.sV = sV; p.x_len = x_len;
  p.B = B; p.T = T; p.V = V; p.blank = blank; p.W = beam_width; p.space_id = space_id;
  p.has_lm = lm ? 1 : 0;
  if (lm) p.lm = lm->dev_view(); else memset(&p.lm, 0, sizeof(p.lm));
  p.lmwt = lm ? lmwt : 0.0;                       // ctc_decoder.cpp:72-74
  p.wip = wip; p.oov = oov_penalty;
  p.out = out; p.max_out = max_out; p.out_len = out_len;
  p.nodes = reinterpret_cast<BeamNode*>(ws + l.nodes); p.free_nodes = reinterpret_cast<int*>(ws + l.free_nodes);
  p.ctab = reinterpret_cast<int*>(ws + l.ctab); p.free_tabs = reinterpret_cast<int*>(ws + l.free_tabs);
  p.cand = reinterpret_cast<int*>(ws + l.cand); p.cand2 = reinterpret_cast<int*>(ws + l.cand2);
  p.status = reinterpret_cast<int*>(ws + l.status);
  p.NCAP = l.NCAP; p.TCAP = l.TCAP; p.CMAX = l.CMAX; p.NP2 = l.NP2;
  const size_t lds = sizeof(double) * ((size_t)l.NP2 + V) + sizeof(int) * (size_t)l.NP2;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == E2E_F32) {
    E2E_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_beam_kernel<float>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    hipLaunchKernelGGL(ctc_beam_kernel<float>, dim3(B), dim3(kThreads), lds, s, p);
  } else {
    E2E_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_beam_kernel<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    hipLaunchKernelGGL(ctc_beam_kernel<double>, dim3(B), dim3(kThreads), lds, s, p);
  }
  E2E_HIP_CHECK(hipGetLastError(), "ctc_beam_kernel launch");
  return E2E_OK;
}

// pool-exhaustion / truncation report of the last call that used `workspace` (synchronises): 0 = ok
extern "C" int e2e_ctc_beam_status(const void* workspace, int B, int T, int V, int beam_width) {
  const BeamLayout l = beam_layout(B, T, V, beam_width);
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const char* ws = reinterpret_cast<const char*>((base + 255) & ~(uintptr_t)255);
  std::vector<int> st((size_t)B);
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  if (B && hipMemcpy(st.data(), ws + l.status, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost) != hipSuccess) return E2E_ERR_HIP;
  for (int v : st) if (v) { set_error("beam search: utterance status %d (1 node pool, 2 table pool, 3 output truncated)", v); return E2E_ERR_UNSUPPORTED; }
  return E2E_OK;
}

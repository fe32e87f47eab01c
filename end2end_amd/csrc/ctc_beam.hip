// Prefix beam search (Hannun et al. 2014) with optional n-gram LM scoring, on the GPU.
// Restates src/decoders/ctc_decoder.cpp:153-201 (driver), :353-441 (decode_sentence), :247-312
// (get_next_prefix), :314-318 (score), :232-245 (get_sentence).
//
// One 1024-thread workgroup per utterance; time is serial, the W*V (prefix, character) pairs of a step are
// spread over the threads.  The prefix tree lives in a per-utterance node pool in HBM (parent, character: written once,
// read only for the final sentence).  The reference's shared_ptr / weak_ptr ownership -- a prefix lives while it is in
// the beam or has a living child; the weak `next_data` entry of its parent expires when it dies -- is restated without
// reference counts:
//   * only beam members are ever asked for a child, so what has to be known is, for every member P and character c,
//     whether the child (P, c) is alive, i.e. is a member or an ancestor of one (a pruned child that is kept alive by
//     a descendant is still found, receives probability that nobody reads, and is NOT re-added: quirk Q7);
//   * every member j carries its GUARD: the first prefix on its way to the root (itself included) whose parent is a
//     member, as (owner = that parent's position in the beam, character, node id).  Every alive child of a member is
//     the guard of some member (of itself, if it is one), so the members' child tables are REBUILT each step from the
//     guards: cleared, then one LDS write per member.  When a guard's owner leaves the beam, the guard is inherited
//     from the owner's own guard (the next alive prefix up the path whose parent is still a member).
//   No global atomics, no reads of the node pool inside the step loop.
// All scores are IEEE doubles with the reference's two-argument log-sum-exp.  The beam (probabilities, LM state, child
// tables) lives in LDS; prefixes are created LAZILY: a step scores all W*V would-be prefixes, selects the W survivors
// (radix select on an order-preserving 64-bit key started below the bits the best and the worst surviving score share --
// one pass in practice -- then a rank-by-counting of the <= W+63 gathered candidates into (score descending, position
// ascending) order: the reference's nth_element leaves ties unspecified, quirk Q9; the oracle uses the same total order)
// and only then allocates nodes for the new ones.  Upstream a new prefix that does not survive its first pruning is
// destroyed at once, so this is equivalent and saves ~W*V allocations, LM-state writes and frees per step.
//
// The language model stands where KenLM stands upstream (src/decoders/ctc_decoder.cpp:60-71,77-88,264-308):
// host-side ARPA reader (plain or gzip), device-resident open-addressing tables, standard back-off scoring.  A beam
// member's V answers are looked up once per LM state when it enters the beam and kept in LDS (see lm_query).
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <stdlib.h>

#include "common.h"

namespace e2e {

constexpr int kLmMaxOrder = 6;     // KENLM_MAX_ORDER=6, CMakeLists.txt:36
constexpr int kCtx = kLmMaxOrder - 1;

struct NgSlot { uint32_t ids[kLmMaxOrder]; int32_t n; float prob; float backoff; };
// The same tables in the form the kernel probes: one 16-byte load per probe, matched by the n-gram's 64-bit hash
// (the loader checks that no two n-grams of the model share one; a queried n-gram that is NOT in the model would have
// to collide in all 64 bits with the entry at its probe position to be mistaken for it).
struct NgSig {                                                   // sig 0: empty
  uint64_t sig; float prob; float backoff;
  // second half, read for CONTEXT lookups only: one bit per continuation word of this n-gram (bit cont_bit(w) is set if the
  // model lists (this n-gram, w)).  A query's longer n-grams are only looked up when their context says they may exist:
  // what a beam step asks is bound by the number of distinct cache lines its waves touch, and for most
  // (context, word) pairs a beam search tries there is no such n-gram.  No false negatives; contexts with many
  // continuations fill their 64 bits and every lookup is made, as before.
  uint64_t cont; uint64_t pad;
};
// Unigrams are not hashed at all: one 16-byte entry per word id (prob > 0: the model has no such unigram).  A few hundred
// KB that stay in L2, where the hashed table (megabytes, one slot per random line) is a trip past it for every probe.
struct UniEntry { float prob; float backoff; uint64_t cont; };
__host__ __device__ inline int cont_bit(uint32_t w) { return (int)(((uint64_t)w * 0x9E3779B97F4A7C15ULL) >> 58); }
struct VEntry { uint64_t key; uint32_t val; float prob; };       // key 0: empty; prob: the word's unigram log10 p (> 0: none listed)
// The kernel's vocabulary table is a TWO-CHOICE (cuckoo) table: a spelling sits in one of the two slots its hash names, so a
// probe is two loads issued together and never a second round (with linear probing the slowest of a wave's 64 lanes needed
// three).  It is small -- 64 bytes per word -- and stays in L2, where a second line per probe is cheap.  (The n-gram table
// keeps linear probing: since the continuation bits its probes are rare or shared by a state's characters.)
__host__ __device__ inline void two_slots(uint64_t h, uint32_t mask, uint32_t& i1, uint32_t& i2) {
  // (the second slot from a re-mixed hash: bits 32.. of an FNV hash of a short spelling are far from uniform -- 7 191
  //  distinct values for the bench model's 10 003 words -- and cuckoo insertion then fails)
  i1 = (uint32_t)h & mask; i2 = (uint32_t)((h * 0x9E3779B97F4A7C15ULL) >> 32) & mask;
  if (i2 == i1) i2 = i1 ^ 1u;
}

struct LmView {                    // what the kernel sees (device pointers) / what the host scorer sees
  int order;
  const uint64_t* vkeys; const uint32_t* vvals; uint32_t vmask;
  const NgSlot* ng; uint32_t ngmask;
  uint32_t bos;
  const unsigned char* label_bytes; const int* label_off;   // label c spells bytes [off[c], off[c+1])
  int fold_case;
  const NgSig* ngs; const VEntry* vt;                        // device only (null: n-gram hashes collide, use ng / vkeys)
  const UniEntry* uni; uint32_t nwords;                      // device only, with ngs
  float unk_prob;                                            // p(<unk>) (KenLM's -100 if the model has none)
};

__host__ __device__ inline uint64_t fnv_step(uint64_t h, unsigned char b) { return (h ^ b) * 1099511628211ULL; }
constexpr uint64_t kFnvInit = 1469598103934665603ULL;

// hash of an n-gram of word ids (table placement and signature; internal to this file): one multiply per id
__host__ __device__ inline uint64_t ng_mix(uint64_t h, uint32_t id) { h = (h ^ id) * 0x9E3779B97F4A7C15ULL; return h ^ (h >> 32); }
__host__ __device__ inline uint64_t ng_finish(uint64_t h, int n) { h = ng_mix(h, 0x51ED2700u + (uint32_t)n); return h == 0 ? 1 : h; }
__host__ __device__ inline uint64_t ngram_hash(const uint32_t* ids, int n) {
  uint64_t h = kFnvInit;
  for (int i = 0; i < n; i++) h = ng_mix(h, ids[i]);
  return ng_finish(h, n);
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

// log10 p(word | ctx) with ARPA back-off; ctx is most-recent-first.  Float accumulation in KenLM's order: the prob of
// the longest listed n-gram, then the back-off weights of the longer contexts, shortest context first.
__host__ __device__ inline float lm_base_score(const LmView& lm, const uint32_t* ctx, int ctx_len, uint32_t word,
                                               uint32_t* out_ctx, int* out_len) {
  int n = ctx_len; if (n > lm.order - 1) n = lm.order - 1;
  uint32_t ids[kLmMaxOrder];
  float bo[kLmMaxOrder + 1];
  float result = 0.f; int found_k = -1;
  for (int k = n; k >= 0; k--) {
    for (int i = 0; i < k; i++) ids[i] = ctx[k - 1 - i];
    ids[k] = word;
    const NgSlot* s = lm_ngram_find(lm, ids, k + 1);
    if (s) { result = s->prob; found_k = k; break; }
    bo[k] = 0.f;
    if (k > 0) { const NgSlot* c = lm_ngram_find(lm, ids, k); if (c) bo[k] = c->backoff; }
  }
  if (found_k < 0) { const uint32_t z = 0; const NgSlot* u = lm_ngram_find(lm, &z, 1); result = u ? u->prob : -100.f; found_k = 0; }
  for (int k = found_k + 1; k <= n; k++) result += bo[k];
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
  e2e::NgSig* d_ngs = nullptr; e2e::VEntry* d_vt = nullptr;     // (d_ngs stays null if two n-grams share a hash)
  e2e::UniEntry* d_uni = nullptr; uint32_t nwords = 0;
  float unk_prob = -100.f;
  int device = -1;                                       // HIP device that holds the tables (-1: host only)
  e2e::LmView host_view() const {
    return {order, vkeys.data(), vvals.data(), (uint32_t)vkeys.size() - 1, ng.data(), (uint32_t)ng.size() - 1, bos,
            label_bytes.data(), label_off.data(), fold_case, nullptr, nullptr, nullptr, nwords, unk_prob};
  }
  e2e::LmView dev_view() const {
    return {order, d_vkeys, d_vvals, (uint32_t)vkeys.size() - 1, d_ng, (uint32_t)ng.size() - 1, bos,
            d_label_bytes, d_label_off, fold_case, d_ngs, d_ngs ? d_vt : nullptr, d_ngs ? d_uni : nullptr, nwords, unk_prob};
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
  {
    // KenLM's own binary format (what `build_binary` writes; upstream's LoadVirtual, ctc_decoder.cpp:64, takes it too)
    // is not read here: say so instead of failing to find ARPA sections
    char magic[64] = {0};
    const int got = gzread(f, magic, sizeof(magic) - 1);
    if (got > 0 && strncmp(magic, "mmap lm http://kheafield.com/code", 33) == 0) {
      gzclose(f);
      set_error("%s is a KenLM binary model; this library reads ARPA (plain or .gz) -- convert it back with KenLM, or "
                "load the ARPA file it was built from", path);
      return E2E_ERR_UNSUPPORTED;
    }
    gzrewind(f);
  }
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
  // (load <= 1/4: most misses end at the first slot.  The table's footprint is not what the kernel's queries cost: 2, 3, 4, 8
  //  slots per entry -- 2 to 8 MB for the bench's model -- measured 28.0 / 26.6 / 26.6 / 26.1 ms per C4 batch.)
  lm->ng.assign(pow2_at_least(entries.size() * 4 + 16), NgSlot{{0, 0, 0, 0, 0, 0}, 0, 0.f, 0.f});
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
  lm->vkeys.assign(pow2_at_least(words.size() * 4 + 16), 0);
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
  // the kernel's 16-byte forms of the two tables (same slots)
  std::vector<NgSig> ngs(lm->ng.size(), NgSig{0, 0.f, 0.f, 0, 0});
  std::vector<UniEntry> uni(words.size(), UniEntry{1.f, 0.f, 0});
  lm->nwords = (uint32_t)words.size();
  bool sig_ok = true;
  {
    std::vector<uint64_t> seen;
    seen.reserve(entries.size());
    for (size_t i = 0; i < lm->ng.size(); i++) {
      const NgSlot& sl = lm->ng[i];
      if (sl.n == 0) continue;
      ngs[i].sig = ngram_hash(sl.ids, sl.n); ngs[i].prob = sl.prob; ngs[i].backoff = sl.backoff;
      seen.push_back(ngs[i].sig);
      if (sl.n == 1 && sl.ids[0] == 0) lm->unk_prob = sl.prob;
    }
    std::sort(seen.begin(), seen.end());
    sig_ok = std::adjacent_find(seen.begin(), seen.end()) == seen.end();
    // continuation bits: (w1 .. wn) sets bit cont_bit(wn) of its context (w1 .. wn-1).  The kernel's scorer
    // (lm_score_parallel) looks an n-gram up only behind a HIT of its context, so a model that lists an n-gram without its
    // context (SRILM-pruned files do; KenLM inserts blank entries for them) cannot use these tables: the id-keyed walk
    // (lm_base_score), which probes every order, takes over.
    bool contexts_listed = true;
    const LmView hv = lm->host_view();
    for (size_t i = 0; i < lm->ng.size(); i++) {
      const NgSlot& sl = lm->ng[i];
      if (sl.n == 1) { uni[sl.ids[0]].prob = sl.prob; uni[sl.ids[0]].backoff = sl.backoff; }
    }
    for (size_t i = 0; i < lm->ng.size() && contexts_listed; i++) {
      const NgSlot& sl = lm->ng[i];
      if (sl.n < 2) continue;
      const NgSlot* c = lm_ngram_find(hv, sl.ids, sl.n - 1);
      if (!c) { contexts_listed = false; break; }
      const uint64_t bit = 1ULL << cont_bit(sl.ids[sl.n - 1]);
      if (sl.n == 2) uni[sl.ids[0]].cont |= bit; else ngs[(size_t)(c - lm->ng.data())].cont |= bit;
    }
    if (!contexts_listed) {
      sig_ok = false;
      if (getenv("E2E_LM_DEBUG")) fprintf(stderr, "e2e_lm: an n-gram's context is not listed; using the id tables (slower)\n");
    }
  }
  std::vector<VEntry> vt(lm->vkeys.size(), VEntry{0, 0u, 1.f});
  for (size_t i = 0; i < lm->vkeys.size() && sig_ok; i++) {      // (one entry per distinct folded spelling already)
    if (lm->vkeys[i] == 0) continue;
    // cuckoo insertion: a free slot of the item's two, else evict the occupant of one and move that on
    VEntry item{lm->vkeys[i], lm->vvals[i], uni[lm->vvals[i]].prob};
    const uint32_t mask = (uint32_t)vt.size() - 1;
    uint32_t i1, i2;
    two_slots(item.key, mask, i1, i2);
    if (vt[i1].key == 0) { vt[i1] = item; continue; }
    if (vt[i2].key == 0) { vt[i2] = item; continue; }
    uint32_t pos = i1;
    bool placed = false;
    for (int kick = 0; kick < 2000 && !placed; kick++) {
      std::swap(item, vt[pos]);
      if (item.key == 0) { placed = true; break; }
      two_slots(item.key, mask, i1, i2);
      pos = pos == i1 ? i2 : i1;
    }
    if (!placed) sig_ok = false;                                  // (never seen at load <= 1/4; the id-keyed walk takes over)
  }
  // Self-check of what the kernel will read, against the id tables it stands for: every spelling is found in one of its two
  // vocabulary slots with the right id and unigram; every listed n-gram is found by its signature with the right numbers and
  // its context lists its last word.  (What is NOT listed can only cost a wasted lookup: the filters have no false negatives.)
  if (sig_ok) {
    const uint32_t vmask2 = (uint32_t)vt.size() - 1;
    for (size_t i = 0; i < lm->vkeys.size() && sig_ok; i++) {
      if (lm->vkeys[i] == 0) continue;
      uint32_t i1, i2;
      two_slots(lm->vkeys[i], vmask2, i1, i2);
      const VEntry* e = vt[i1].key == lm->vkeys[i] ? &vt[i1] : vt[i2].key == lm->vkeys[i] ? &vt[i2] : nullptr;
      sig_ok = e && e->val == lm->vvals[i] && e->prob == uni[e->val].prob;
    }
    const LmView hv = lm->host_view();
    for (size_t i = 0; i < lm->ng.size() && sig_ok; i++) {
      const NgSlot& sl = lm->ng[i];
      if (sl.n == 0) continue;
      const uint64_t sig = ngram_hash(sl.ids, sl.n);
      uint32_t j = (uint32_t)sig & ngmask;
      while (ngs[j].sig != sig && ngs[j].sig != 0) j = (j + 1) & ngmask;
      sig_ok = ngs[j].sig == sig && ngs[j].prob == sl.prob && ngs[j].backoff == sl.backoff;
      if (sig_ok && sl.n == 1) sig_ok = uni[sl.ids[0]].prob == sl.prob && uni[sl.ids[0]].backoff == sl.backoff;
      if (sig_ok && sl.n >= 2) {
        const uint64_t bit = 1ULL << cont_bit(sl.ids[sl.n - 1]);
        const NgSlot* c = lm_ngram_find(hv, sl.ids, sl.n - 1);
        const uint64_t cont = sl.n == 2 ? uni[sl.ids[0]].cont : (c ? ngs[(size_t)(c - lm->ng.data())].cont : 0ULL);
        sig_ok = (sl.n == 2 || c) && (cont & bit) != 0;         // (an unlisted context fails: the kernel would never probe the n-gram)
      }
    }
    if (!sig_ok) fprintf(stderr, "e2e_lm: the kernel's tables failed their self-check; using the id tables (slower)\n");
  }
  if (getenv("E2E_LM_DEBUG")) fprintf(stderr, "e2e_lm: %zu entries, %zu words, signature tables %s\n", entries.size(), words.size(), sig_ok ? "ok" : "NOT usable");
  bool ok = (!sig_ok || up((void**)&lm->d_ngs, ngs.data(), ngs.size() * sizeof(NgSig))) &&
            up((void**)&lm->d_vt, vt.data(), vt.size() * sizeof(VEntry)) &&
            up((void**)&lm->d_uni, uni.data(), uni.size() * sizeof(UniEntry)) &&
            up((void**)&lm->d_vkeys, lm->vkeys.data(), lm->vkeys.size() * sizeof(uint64_t)) &&
            up((void**)&lm->d_vvals, lm->vvals.data(), lm->vvals.size() * sizeof(uint32_t)) &&
            up((void**)&lm->d_ng, lm->ng.data(), lm->ng.size() * sizeof(NgSlot)) &&
            up((void**)&lm->d_label_bytes, lm->label_bytes.data(), lm->label_bytes.size()) &&
            up((void**)&lm->d_label_off, lm->label_off.data(), lm->label_off.size() * sizeof(int));
  if (!ok) {
    // no usable GPU: keep the host tables (e2e_lm_word_index / e2e_lm_score still work); e2e_ctc_beam refuses it
    (void)hipGetLastError();
    (void)hipFree(lm->d_vkeys); (void)hipFree(lm->d_vvals); (void)hipFree(lm->d_ng);
    (void)hipFree(lm->d_label_bytes); (void)hipFree(lm->d_label_off); (void)hipFree(lm->d_ngs); (void)hipFree(lm->d_vt); (void)hipFree(lm->d_uni);
    lm->d_vkeys = nullptr; lm->d_vvals = nullptr; lm->d_ng = nullptr; lm->d_label_bytes = nullptr; lm->d_label_off = nullptr;
    lm->d_ngs = nullptr; lm->d_vt = nullptr; lm->d_uni = nullptr;
  } else if (hipGetDevice(&lm->device) != hipSuccess) {
    lm->device = -1;
  }
  *out = lm;
  return E2E_OK;
}

extern "C" void e2e_lm_free(e2e_lm* lm) {
  if (!lm) return;
  (void)hipFree(lm->d_vkeys); (void)hipFree(lm->d_vvals); (void)hipFree(lm->d_ng);
  (void)hipFree(lm->d_label_bytes); (void)hipFree(lm->d_label_off); (void)hipFree(lm->d_ngs); (void)hipFree(lm->d_vt); (void)hipFree(lm->d_uni);
  delete lm;
}

extern "C" int e2e_lm_order(const e2e_lm* lm) { return lm ? lm->order : 0; }
extern "C" int e2e_lm_device(const e2e_lm* lm) { return lm ? lm->device : -1; }

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

constexpr int kThreadsNoLm = 1024, kThreadsLm = 1024;   // (with a language model: 512 threads measured 30 % slower)
constexpr int kMaxCand = 8192;     // W*V + W candidates per step (LDS key array)
constexpr int kLdsBudget = 158 * 1024;
constexpr int kSelBits = 11, kSelBins = 1 << kSelBits;   // radix-select digit (two alternating histograms in LDS)
constexpr int kStateSlots = 512;                       // LM state table of a step (open addressing, <= half full: checked by the host)
constexpr int kSelSmall = 64;                           // a threshold bin this small is finished exactly by one wave

// LM-related state of a prefix (Prefix::lm_*, num_*, last_word, ctc_decoder.h:79-86)
struct LmFields {
  double lm_score, lm_before;
  int num_words, num_oov, num_oov_before, word_len;
  unsigned long long word_hash;            // hash of the spelled last word (get_idx(vector<int>), :84-88)
  unsigned int st[kCtx], stb[kCtx];        // LM context after / before the last word, most recent first
  int st_n, stb_n;
};

// tree node in HBM: the structure only (16 bytes).  Everything a prefix needs while it is in the beam -- the four
// log-probabilities, the LM state, its child table, its parent's id -- lives in LDS; a pruned-but-alive prefix still
// "receives" probability upstream, but nothing ever reads it: quirk Q7 reduces to "its (parent, char) slot stays
// occupied".  Nodes are never reused: at most W prefixes are created per step, so W*(T+3) nodes cover an utterance
// and allocation is a counter (no free list to initialise, read or write).
struct BeamNode {
  int parent, last_char;
};

struct BeamParams {
  const void* lp; int64_t sB, sT, sV; const int64_t* x_len;
  int B, T, V, blank, W, space_id;
  int has_lm; LmView lm; double lmwt, wip, oov;
  int64_t* out; int64_t max_out; int64_t* out_len;
  BeamNode* nodes;                                    // per-utterance workspace
  int NCAP, CMAX, WP2, HS;
};

__device__ __forceinline__ double ninf() { return -__builtin_huge_val(); }

// src/utils/math_utils.h:8-16
__device__ __forceinline__ double lse2(double a, double b) {
  if (a == ninf()) return b;
  if (b == ninf()) return a;
  if (a > b) return a + log(1.0 + exp(b - a));
  return b + log(1.0 + exp(a - b));
}

// inclusive prefix sum over the 64 lanes, all DPP (row_shr 1/2/4/8 with zero fill, row_bcast 15/31)
__device__ __forceinline__ int wave_scan_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// maximum over the 64 lanes, all DPP; every lane of row 3 (lanes 48..63) ends with it, returned wave-uniform
__device__ __forceinline__ int wave_max_i(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));   // row_mirror
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL
// operations (vmcnt(0)), which would expose a memory round trip at every phase that follows a fire-and-forget store or
// the prefetch of the next row.  Two full barriers per step remain: after the rebuild (node stores and reference-count
// atomics before the release) and at the end of the step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Candidate d >= n of a step is the pair q = d - n (q = c*n + i, the reference's order); a pair that creates no prefix
// holds this marker instead of a score: a negative NaN, which the order-preserving key maps BELOW -inf (scores can
// be -inf), and which every pass over the keys skips.
constexpr unsigned long long kNoCandKey = ~0xFFF8000000000000ULL;
// order-preserving map double -> uint64 (larger double <=> larger key)
// (-0.0 and +0.0 compare equal as doubles -- the oracle's order -- so the key is taken of d + 0.0, which is +0.0 for both)
__device__ __forceinline__ unsigned long long okey(double d) {
  unsigned long long u = (unsigned long long)__double_as_longlong(d + 0.0);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}

// the LM part of get_next_prefix (:258-308) for the child (parent pr, char c)
// What a (prefix, character != space) pair asks of the language model: the id of the word spelled so far and its
// score in the prefix's context.  The answer depends on the prefix's LM state and the character only -- not on the time
// step -- so a beam member's V answers are looked up ONCE, when it enters the beam, by V different threads, and kept
// in LDS while it stays (`lmc`): the pair loop and the rebuild then read them instead of walking the n-gram tables
// (global memory, several dependent probes) for every pair of every step.
struct LmAnswer { float sc; uint32_t wi; };
// (LabelTab: the labels' spellings, staged in LDS by the kernel when they fit -- two dependent global loads less per query)
// (one: LDS, per label its byte -- case already folded -- if it is spelled with exactly one, else -1; null when not staged.
//  The usual alphabet: one LDS read instead of three dependent loads through generic pointers.)
struct LabelTab { const int* off; const unsigned char* bytes; const int* one; };
constexpr int kLabelLdsBytes = 512;
__device__ __forceinline__ unsigned long long spell(const BeamParams& p, const LabelTab& lt, unsigned long long h, int c) {
  if (lt.one) { const int o = lt.one[c]; if (o >= 0) return fnv_step(h, (unsigned char)o); }
  for (int bi = lt.off[c]; bi < lt.off[c + 1]; bi++) {
    unsigned char ch = lt.bytes[bi];
    if (p.lm.fold_case && ch >= 'A' && ch <= 'Z') ch += 32;
    h = fnv_step(h, ch);
  }
  return h;
}
// lm_base_score for contexts of up to kParCtx words in two rounds of table probes.  Round 1 (issued by the caller beside the
// vocabulary probe, which it does not depend on): the context k-grams -- their back-off weights and continuation bits.
// Round 2, once the word is known: the unigram, and the (k+1)-grams (ctx[k-1..0], word) whose context lists the word among
// its continuations.  Within a round every lookup that is still open requests its next slot before any answer is consumed.
// The answers are combined in the chain's order (same float additions as lm_base_score).
constexpr int kParCtx = 2;
struct LmContexts { float backoff[kParCtx + 1]; uint64_t cont[kParCtx + 1]; unsigned hit; uint64_t hp[kParCtx + 1]; };
__device__ __forceinline__ void lm_contexts(const LmView& lm, const uint32_t (&ctx)[kCtx], int n, LmContexts& cx) {
  const uint4* tab = reinterpret_cast<const uint4*>(lm.ngs);
  uint64_t h[kParCtx + 1]; uint32_t idx[kParCtx + 1]; uint4 e[kParCtx + 1];
  unsigned open = 0;
  cx.hit = 0; cx.hp[0] = kFnvInit;
#pragma unroll
  for (int k = 1; k <= kParCtx; k++) {
    uint64_t hp = kFnvInit;
#pragma unroll
    for (int i = 0; i < k; i++) hp = ng_mix(hp, ctx[k - 1 - i]);
    cx.hp[k] = hp;
    h[k] = ng_finish(hp, k);
    idx[k] = (uint32_t)h[k] & lm.ngmask;
    cx.backoff[k] = 0.f; cx.cont[k] = 0;
    if (k >= 2 && k <= n) { open |= 1u << k; e[k] = tab[2 * (size_t)idx[k]]; }
  }
  if (n >= 1) {                                   // the one-word context: the unigram entry itself
    const uint4 u = reinterpret_cast<const uint4*>(lm.uni)[ctx[0] < lm.nwords ? ctx[0] : 0];
    if (__uint_as_float(u.x) <= 0.f && ctx[0] < lm.nwords) {
      cx.hit |= 2u; cx.backoff[1] = __uint_as_float(u.y); cx.cont[1] = ((uint64_t)u.w << 32) | u.z;
    }
  }
  while (open) {
#pragma unroll
    for (int k = 2; k <= kParCtx; k++) {
      if (open >> k & 1u) {
        const uint64_t sig = ((uint64_t)e[k].y << 32) | e[k].x;
        if (sig == h[k]) {
          cx.hit |= 1u << k; open &= ~(1u << k);
          cx.backoff[k] = __uint_as_float(e[k].w);
          const uint4 c = tab[2 * (size_t)idx[k] + 1];                  // (same cache line as the slot's first half)
          cx.cont[k] = ((uint64_t)c.y << 32) | c.x;
        } else if (sig == 0) open &= ~(1u << k);
        else { idx[k] = (idx[k] + 1) & lm.ngmask; e[k] = tab[2 * (size_t)idx[k]]; }
      }
    }
  }
}
__device__ __forceinline__ float lm_score_parallel(const LmView& lm, const LmContexts& cx, int n, uint32_t word, float uni_prob) {
  const uint4* tab = reinterpret_cast<const uint4*>(lm.ngs);
  uint64_t h[kParCtx + 1]; uint32_t idx[kParCtx + 1]; uint4 e[kParCtx + 1];
  float prob[kParCtx + 1];
  unsigned open = 0, hit = 0;
  const int bit = cont_bit(word);
#pragma unroll
  for (int k = 0; k <= kParCtx; k++) {
    h[k] = ng_finish(ng_mix(cx.hp[k], word), k + 1);
    idx[k] = (uint32_t)h[k] & lm.ngmask;
    prob[k] = 0.f;
    // (only if the context is listed and lists the word among its continuations)
    if (k >= 1 && k <= n && (cx.hit >> k & 1u) && (cx.cont[k] >> bit & 1ULL)) { open |= 1u << k; e[k] = tab[2 * (size_t)idx[k]]; }
  }
  if (uni_prob <= 0.f) { hit |= 1u; prob[0] = uni_prob; }          // (the unigram came with the vocabulary entry)
  while (open) {
#pragma unroll
    for (int k = 1; k <= kParCtx; k++) {
      if (open >> k & 1u) {
        const uint64_t sig = ((uint64_t)e[k].y << 32) | e[k].x;
        if (sig == h[k]) { hit |= 1u << k; open &= ~(1u << k); prob[k] = __uint_as_float(e[k].z); }
        else if (sig == 0) open &= ~(1u << k);
        else { idx[k] = (idx[k] + 1) & lm.ngmask; e[k] = tab[2 * (size_t)idx[k]]; }
      }
    }
  }
  // longest listed n-gram, then the back-off weights of the longer contexts, shortest first (KenLM's float order)
  int found_k = 0;
  float result = lm.unk_prob;
#pragma unroll
  for (int k = 0; k <= kParCtx; k++)
    if (k <= n && (hit >> k & 1u)) { result = prob[k]; found_k = k; }
#pragma unroll
  for (int k = 1; k <= kParCtx; k++)
    if (k > found_k && k <= n && (cx.hit >> k & 1u)) result += cx.backoff[k];
  return result;
}
// FAST: the model has signature tables and at most kParCtx words of context (checked by the host): only the round-probed
// walk is compiled in.  Otherwise: the general walk over the id tables (any order up to 6).
template <bool FAST>
__device__ __forceinline__ LmAnswer lm_query(const BeamParams& p, const LabelTab& lt, const LmFields& pr, int parent_last, int c) {
  const bool new_word = pr.num_words == 0 || parent_last == p.space_id;                           // :258-259 (c != space)
  LmAnswer a;
  uint64_t h = spell(p, lt, new_word ? kFnvInit : pr.word_hash, c);

  int cn = new_word ? pr.st_n : pr.stb_n;
  if (FAST) {
    uint32_t ctx[kCtx];
#pragma unroll
    for (int s2 = 0; s2 < kCtx; s2++) ctx[s2] = new_word ? pr.st[s2] : pr.stb[s2];
    if (h == 0) h = 1;
    if (cn > p.lm.order - 1) cn = p.lm.order - 1;
    uint32_t v1, v2;
    two_slots(h, p.lm.vmask, v1, v2);
    const uint4 e1 = reinterpret_cast<const uint4*>(p.lm.vt)[v1], e2 = reinterpret_cast<const uint4*>(p.lm.vt)[v2];   // (in flight beside the context lookups)
    LmContexts cx;
    lm_contexts(p.lm, ctx, cn, cx);
    const bool in1 = (((uint64_t)e1.y << 32) | e1.x) == h, in2 = (((uint64_t)e2.y << 32) | e2.x) == h;
    a.wi = in1 ? e1.z : in2 ? e2.z : 0u;                             // NotFound() == <unk> == 0
    const float uni_prob = in1 ? __uint_as_float(e1.w) : in2 ? __uint_as_float(e2.w) : p.lm.unk_prob;
    a.sc = lm_score_parallel(p.lm, cx, cn, a.wi, uni_prob);
  } else {
    a.wi = lm_word_lookup(p.lm, h);
    a.sc = lm_base_score(p.lm, new_word ? pr.st : pr.stb, cn, a.wi, nullptr, nullptr);
  }
  return a;
}

// Quirk Q8's division, (double)score / ln 10, without the division: for every float whose magnitude lies in [2^-60, 2^20)
// -- all 671 088 640 of them checked against the IEEE quotient on the host (both signs are symmetric) -- the product with the
// rounded reciprocal, corrected once through the exact remainder, IS the correctly rounded quotient.  Three operations
// instead of the ~35 of a double division, once per (prefix, character) pair of every step.
__device__ __forceinline__ double div_ln10(float sc) {
  const double kLogE10 = 2.302585092994045684, kInv = 1.0 / 2.302585092994045684;
  const double x = (double)sc;
  const float a = fabsf(sc);
  if (a >= 0x1p-60f && a < 0x1p20f) {
    const double q0 = x * kInv;
    return fma(fma(-q0, kLogE10, x), kInv, q0);
  }
  return x / kLogE10;
}

// the LM part of get_next_prefix (:258-308) for the child (parent pr, char c), given the LM's answer for the pair
// (LM = false: the kernel instantiated for decoding without a language model touches num_words only -- the word
// insertion penalty needs it -- and none of the LM state, which otherwise costs the pair loop a third of its
// instructions and the kernel its scratch memory)
template <bool LM>
__device__ __forceinline__ void child_lm(const BeamParams& p, const LabelTab& lt, const LmFields& pr, int parent_last, int c, LmAnswer ans, LmFields& nn) {
  const bool new_word = c != p.space_id && (pr.num_words == 0 || parent_last == p.space_id);     // :258-259
  nn.num_words = pr.num_words + (new_word ? 1 : 0);
  nn.lm_score = 0.0; nn.num_oov = 0;
  if (!LM) return;
  nn.lm_before = 0.0; nn.num_oov_before = 0;
  nn.word_len = 0; nn.word_hash = kFnvInit; nn.st_n = 0; nn.stb_n = 0;
  if (c != p.space_id) {
    nn.word_hash = spell(p, lt, new_word ? kFnvInit : pr.word_hash, c);
    nn.word_len = (new_word ? 0 : pr.word_len) + 1;
    if (new_word) {                                                       // :265-281
      for (int s = 0; s < pr.st_n; s++) nn.stb[s] = pr.st[s];
      nn.stb_n = pr.st_n; nn.lm_before = pr.lm_score; nn.num_oov_before = pr.num_oov;
    } else {                                                              // :282-297
      for (int s = 0; s < pr.stb_n; s++) nn.stb[s] = pr.stb[s];
      nn.stb_n = pr.stb_n; nn.lm_before = pr.lm_before; nn.num_oov_before = pr.num_oov_before;
    }
    // the state after the word: the word, then the context it was scored in (lm_base_score's out_ctx)
    int m = nn.stb_n; if (m > p.lm.order - 1) m = p.lm.order - 1;
    m = m + 1; if (m > p.lm.order - 1) m = p.lm.order - 1;
    if (m > 0) nn.st[0] = ans.wi;
    for (int s = 1; s < m; s++) nn.st[s] = nn.stb[s - 1];
    nn.st_n = m;
    nn.lm_score = nn.lm_before + div_ln10(ans.sc);                 // quirk Q8: divides by ln 10
    nn.num_oov = nn.num_oov_before + (ans.wi == 0 ? 1 : 0);
  } else {                                                                // :299-307 copy
    nn.word_hash = pr.word_hash; nn.word_len = pr.word_len;
    nn.lm_score = pr.lm_score; nn.lm_before = pr.lm_before;
    nn.num_oov = pr.num_oov; nn.num_oov_before = pr.num_oov_before;
    for (int s = 0; s < pr.st_n; s++) nn.st[s] = pr.st[s];
    for (int s = 0; s < pr.stb_n; s++) nn.stb[s] = pr.stb[s];
    nn.st_n = pr.st_n; nn.stb_n = pr.stb_n;
  }
}
// ... and only what the score of the would-be prefix needs (the pair loop)
template <bool LM>
__device__ __forceinline__ void child_score_fields(const BeamParams& p, const LmFields& pr, int parent_last, int c, LmAnswer ans, LmFields& nn) {
  const bool new_word = c != p.space_id && (pr.num_words == 0 || parent_last == p.space_id);
  nn.num_words = pr.num_words + (new_word ? 1 : 0);
  nn.lm_score = 0.0; nn.num_oov = 0;
  if (!LM) return;
  if (c != p.space_id) {
    const double before = new_word ? pr.lm_score : pr.lm_before;
    const int oov_before = new_word ? pr.num_oov : pr.num_oov_before;
    nn.lm_score = before + div_ln10(ans.sc);
    nn.num_oov = oov_before + (ans.wi == 0 ? 1 : 0);
  } else {
    nn.lm_score = pr.lm_score; nn.num_oov = pr.num_oov;
  }
}

// get_prev_full_prob_with_lmwt, :314-318, from the already "next_step"-ed probabilities
template <bool LM>
__device__ __forceinline__ double beam_score_full(const BeamParams& p, double full, const LmFields& lm) {
  const double lm_score = LM ? lm.lm_score : 0.0;
  const int num_oov = LM ? lm.num_oov : 0;
  return full + lm_score * p.lmwt - lm.num_words * p.wip + num_oov * p.oov;
}
template <bool LM>
__device__ __forceinline__ double beam_score(const BeamParams& p, double ppnb, double ppb, const LmFields& lm) {
  // (without a language model lm_score and num_oov are zero for every prefix; the expression keeps its shape so that
  // the result has the reference's bits for any lmwt / oov the caller passes)
  const double lm_score = LM ? lm.lm_score : 0.0;
  const int num_oov = LM ? lm.num_oov : 0;
  return lse2(ppnb, ppb) + lm_score * p.lmwt - lm.num_words * p.wip + num_oov * p.oov;
}
template <bool LM>
__device__ __forceinline__ void copy_lm(LmFields& dst, const LmFields& src) {
  if (LM) dst = src; else dst.num_words = src.num_words;
}

// members of the beam, structure of arrays in LDS (two copies: the beam is rebuilt into the other one every step)
struct Members {
  double* ppb; double* ppnb;   // prev_prob_blank / prev_prob_not_blank
  double* npb; double* npnb;   // this step's prob_blank / prob_not_blank (become prev at next_step)
  double* inc;                 // contribution to prob_not_blank arriving from the parent (if it is in the beam)
  double* full;                // log_sum_exp(prev_pnb, prev_pb): carried from the previous step's score (nfull) / a new member's val
  double* nfull;               // log_sum_exp(npnb, npb), computed for this step's score
  int* node; int* last; int* kept;
  int* newpos;                 // position in the beam this step selects (valid where kept)
  int* gown; int* gchar; int* gnode;   // the member's guard (see the file header): owner's position (-1: none), character, node
  int* from;                   // position in the previous beam (-1: the member is new)
  LmFields* lm;
  __device__ unsigned char* carve(unsigned char* q, int W) {
    ppb = (double*)q; q += sizeof(double) * W; ppnb = (double*)q; q += sizeof(double) * W;
    npb = (double*)q; q += sizeof(double) * W; npnb = (double*)q; q += sizeof(double) * W;
    inc = (double*)q; q += sizeof(double) * W; full = (double*)q; q += sizeof(double) * W;
    nfull = (double*)q; q += sizeof(double) * W;
    lm = (LmFields*)q; q += sizeof(LmFields) * W;
    node = (int*)q; q += sizeof(int) * W; last = (int*)q; q += sizeof(int) * W;
    kept = (int*)q; q += sizeof(int) * W; newpos = (int*)q; q += sizeof(int) * W;
    gown = (int*)q; q += sizeof(int) * W; gchar = (int*)q; q += sizeof(int) * W;
    gnode = (int*)q; q += sizeof(int) * W;
    from = (int*)q; q += sizeof(int) * W;        // (an even number of int arrays: the next set starts 8-byte aligned)
    return q;
  }
  __host__ __device__ static size_t bytes(int W) { return (size_t)W * (7 * sizeof(double) + sizeof(LmFields) + 8 * sizeof(int)); }
};

struct BeamLds {
  static size_t bytes(int W, int V, int CMAX, int WP2, int HS, bool lm) {
    return sizeof(double) * ((size_t)CMAX + 2 * V + WP2 + kSelSmall) +
           sizeof(int) * ((size_t)WP2 + kSelSmall + 2 * (size_t)W * V + 2 * kSelBins + 64 + 4 * (size_t)HS) +
           2 * Members::bytes(W) + (lm ? 2 * sizeof(LmAnswer) * (size_t)W * V + sizeof(int) * (size_t)(2 * V + 2) + kLabelLdsBytes : 0) + 64;
  }
};

// node id -> position in the beam, for the <= W members: open addressing in LDS, one table per member set (the
// table of the set that is being built is cleared a phase earlier).  Replaces a `slot` field in the HBM nodes that
// cost a global round trip per step to read.
struct SlotMap {
  int* key; int* val; int mask;
  __device__ static unsigned hash(int k) { return (unsigned)k * 2654435761u; }
  __device__ void insert(int k, int j) const {
    unsigned h = (hash(k) >> 8) & mask;
    while (atomicCAS(&key[h], -1, k) != -1) h = (h + 1) & mask;
    val[h] = j;
  }
  __device__ int find(int k) const {
    unsigned h = (hash(k) >> 8) & mask;
    for (;;) {
      const int kk = key[h];
      if (kk == k) return val[h];
      if (kk == -1) return -1;
      h = (h + 1) & mask;
    }
  }
};

#ifdef E2E_BEAM_PROFILE
} }  // leave the namespaces for the device symbol
__device__ unsigned long long g_beam_prof[16];
__device__ unsigned long long g_beam_sigs[1 << 17];
__device__ int g_beam_nsig;
namespace e2e { namespace {
#define BPROF(slot) do { if (b == 0 && tid == 0) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); g_beam_prof[slot] += _n - _tprev; _tprev = _n; } } while (0)
#else
#define BPROF(slot) do {} while (0)
#endif

// NT threads per workgroup
// LMK: 0 no language model, 1 the general LM walk, 2 the fast one (see lm_query)
template <typename IO, int LMK, int NT>
__global__ __launch_bounds__(NT) void ctc_beam_kernel(BeamParams p) {
  constexpr int kThreads = NT;
  constexpr bool LM = LMK != 0;
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V, W = p.W, blank = p.blank;
  // ---- LDS carve-up ----
  unsigned char* q8 = smem;
  double* key = (double*)q8; q8 += sizeof(double) * p.CMAX;          // score of candidate d: old members, then the n*V pairs
  double* srow2 = (double*)q8; q8 += sizeof(double) * 2 * V;         // this step's and the next step's log-probabilities
  double* skey = (double*)q8; q8 += sizeof(double) * (p.WP2 + kSelSmall);   // the gathered candidates, for the final ordering
  // (the two member sets and the two slot maps are addressed as "base + set * size", never through an array of
  // pointer structs: a pointer that went through memory loses its LDS address space and every access through it
  // becomes a FLAT instruction -- slower, and not ordered by an LDS-only barrier)
  unsigned char* const mem0 = q8;
  const size_t mbytes = Members::bytes(W);
  q8 += 2 * mbytes;
  Members M0; M0.carve(mem0, W);
  int* sidx = (int*)q8; q8 += sizeof(int) * (p.WP2 + kSelSmall);
  int* const ctab0 = (int*)q8; q8 += sizeof(int) * 2 * (size_t)W * V; // [set][member][V] child tables (weak next_data)
  int* hist = (int*)q8; q8 += sizeof(int) * 2 * kSelBins;
  int* s_part = (int*)q8; q8 += sizeof(int) * 64;
  int* const sm0 = (int*)q8; q8 += sizeof(int) * 4 * p.HS;            // [set][key | val][HS]
  LmAnswer* const lmc0 = (LmAnswer*)q8; if (LM) q8 += 2 * sizeof(LmAnswer) * (size_t)W * V;   // [set][member][V] the LM's answers
  int* const lab_off = (int*)q8; if (LM) q8 += sizeof(int) * (size_t)(V + 2);
  unsigned char* const lab_bytes = q8; if (LM) q8 += kLabelLdsBytes;
  int* const lab_one = (int*)q8; if (LM) q8 += sizeof(int) * (size_t)V;
  LabelTab lt; lt.off = p.lm.label_off; lt.bytes = p.lm.label_bytes; lt.one = nullptr;
  auto slot_map = [&](int set) { SlotMap m; m.key = sm0 + set * 2 * p.HS; m.val = m.key + p.HS; m.mask = p.HS - 1; return m; };
  __shared__ int s_next_node, s_err, s_krem, s_done, s_bin;
  // per-step accumulators, double-buffered by step parity: a step resets the NEXT step's set while nobody uses it, so
  // that no barrier is needed between the end of one step and the pair loop of the next
  __shared__ int s_total_new2[2], s_nnew2[2];
  __shared__ unsigned s_hi2[2], s_lo2[2];
  __shared__ unsigned long long s_prefix;

  BeamNode* nodes = p.nodes + (size_t)b * p.NCAP;
  const IO* lp = reinterpret_cast<const IO*>(p.lp) + (int64_t)b * p.sB;
  int64_t Tq = p.x_len[b];
  const int T = Tq < 0 ? 0 : (Tq > p.T ? p.T : (int)Tq);

  // ---- pools, root prefix (get_initial_prefix, :222-230) ----
  for (int c = tid; c < V; c += kThreads) ctab0[c] = -1;                             // set 0, member 0 = the root
  for (int h = tid; h < p.HS; h += kThreads) { sm0[h] = -1; sm0[2 * p.HS + h] = -1; }
  if (T > 0) for (int c = tid; c < V; c += kThreads) srow2[c] = (double)lp[(int64_t)c * p.sV];
  if (LM) {
    const int nbytes = p.lm.label_off[V];
    if (nbytes <= kLabelLdsBytes) {                       // (uniform) the spellings fit: read them from LDS from now on
      for (int c = tid; c <= V; c += kThreads) lab_off[c] = p.lm.label_off[c];
      for (int i = tid; i < nbytes; i += kThreads) lab_bytes[i] = p.lm.label_bytes[i];
      lt.off = lab_off; lt.bytes = lab_bytes;
    }
    for (int c = tid; c < V; c += kThreads) {
      const int o0 = p.lm.label_off[c];
      int ch = -1;
      if (p.lm.label_off[c + 1] == o0 + 1) {
        ch = p.lm.label_bytes[o0];
        if (p.lm.fold_case && ch >= 'A' && ch <= 'Z') ch += 32;
      }
      lab_one[c] = ch;
    }
    lt.one = lab_one;
  }
  if (tid == 0) {
    s_next_node = 1; s_err = 0;                                                     // node 0 is taken
    s_hi2[0] = 0u; s_lo2[0] = 0xffffffffu; s_total_new2[0] = 0; s_nnew2[0] = 0;
    BeamNode& r = nodes[0];
    r.parent = -1; r.last_char = -1;
    LmFields l;
    l.lm_score = 0.0; l.lm_before = 0.0; l.num_words = 0; l.num_oov = 0; l.num_oov_before = 0; l.word_len = 0;
    l.word_hash = kFnvInit; l.st_n = 0; l.stb_n = 0;
    if (p.has_lm) { l.st[0] = p.lm.bos; l.st_n = 1; l.stb[0] = p.lm.bos; l.stb_n = 1; }
    M0.ppb[0] = 0.0; M0.ppnb[0] = ninf(); M0.full[0] = lse2(ninf(), 0.0); M0.inc[0] = ninf(); M0.kept[0] = 0; M0.node[0] = 0; M0.last[0] = -1; M0.gown[0] = -1; M0.gchar[0] = 0; M0.gnode[0] = 0;
    M0.lm[0] = l;
  }
  __syncthreads();
  if (tid == 0) slot_map(0).insert(0, 0);
  if (LM) for (int c = tid; c < V; c += kThreads) if (c != blank && c != p.space_id) lmc0[c] = lm_query<LMK == 2>(p, lt, M0.lm[0], -1, c);
  __syncthreads();
  int n = 1, cur = 0;
  // the pair loop's thread layout depends on n alone, and n is W for all but an utterance's first steps: worked out when n
  // changes (three integer divisions, ~100 instructions per thread of a phase that is bound by the instructions it issues)
  int lay_n = -1, lay_P = 1, lay_mpp = kThreads, lay_ii0 = 0, lay_part = 0;
  unsigned lay_nmagic = 0u;          // ceil(2^32 / n): q / n for q < W * V by one multiplication (n > 1)
  // e / V for row indices e < W * V <= kMaxCand by multiplication: ceil(2^32 / V), exact while e * V < 2^32
  const unsigned v_magic = V > 1 ? (unsigned)((0x100000000ULL + (unsigned)V - 1u) / (unsigned)V) : 0u;
  auto div_v = [&](int e) -> int { return V > 1 ? (int)__umulhi((unsigned)e, v_magic) : e; };
#ifdef E2E_BEAM_PROFILE
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
  if (b == 0 && tid == 0) for (int i = 0; i < 16; i++) g_beam_prof[i] = 0;
#endif

  for (int t = 0; t < T; t++) {
    Members A, Bm;
    A.carve(mem0 + (size_t)cur * mbytes, W);
    Bm.carve(mem0 + (size_t)(cur ^ 1) * mbytes, W);
    const SlotMap mapA = slot_map(cur);              // node -> position among the current members
    const SlotMap mapB = slot_map(cur ^ 1);          // ... among the members this step selects (filled in the rebuild)
    unsigned long long* const ukey = reinterpret_cast<unsigned long long*>(key);   // inside the step key[] holds okey(score)
    unsigned long long* const uskey = reinterpret_cast<unsigned long long*>(skey);
    const int* const ctab = ctab0 + (size_t)cur * W * V;          // child tables of the current members: [member][V]
    int* const ctabB = ctab0 + (size_t)(cur ^ 1) * W * V;         // ... of the members this step selects
    const LmAnswer* const lmcA = lmc0 + (size_t)cur * W * V;
    LmAnswer* const lmcB = lmc0 + (size_t)(cur ^ 1) * W * V;
    double* const srow = srow2 + (t & 1) * V;
    // the next step's row is requested now and parked in LDS at the end of the step: no global round trip at a
    // step's start
    double next_lp = 0.0;
    if (tid < V && t + 1 < T) next_lp = (double)lp[(int64_t)(t + 1) * p.sT + (int64_t)tid * p.sV];
    // (no barrier here: a member's inc / kept / full were set when it was placed, this step's accumulators were reset
    // during the previous step, and the next beam's slot map, child tables and the selection's first histogram are
    // cleared during the members phase below)
    const int ps = t & 1;
    unsigned& s_hi = s_hi2[ps]; unsigned& s_lo = s_lo2[ps];
    int& s_total_new = s_total_new2[ps]; int& s_nnew = s_nnew2[ps];
    // pairs: candidate q = c*n + i is the reference's order (character outer, prefix inner, :370-395).  The threads are
    // laid out member-major -- P threads per member, each taking every P-th character -- so that what a pair needs of
    // its member (probabilities, last character, word count, LM state) is read once per thread, not once per pair.
    const int npairs = n * V;
    if (n != lay_n) {                                           // (uniform)
      lay_n = n; lay_P = n < kThreads ? kThreads / n : 1; lay_mpp = kThreads / lay_P; lay_ii0 = tid / lay_P; lay_part = tid - lay_ii0 * lay_P;
      lay_nmagic = n > 1 ? (unsigned)((0x100000000ULL + (unsigned)n - 1u) / (unsigned)n) : 0u;
    }
    const int P = lay_P;                                        // threads per member
    const int members_per_pass = lay_mpp;
    const int part = lay_part;
    int my_new = 0;
    // blank shares, child shares, scores of the would-be prefixes (weak child lookup, :250-252)
    // (the high words of the largest key of all and of the smallest key of the old members bracket the selection
    // threshold; they are collected while the keys are produced)
    unsigned key_hi = 0u, key_lo = 0xffffffffu;
    for (int ii = lay_ii0; ii < n; ii += members_per_pass) {
      const double full = A.full[ii], ppb = A.ppb[ii];
      const int last = A.last[ii];
      LmFields pr;                                              // (only the fields the scores need are ever loaded)
      pr.num_words = A.lm[ii].num_words;
      if (LM) { pr.lm_score = A.lm[ii].lm_score; pr.lm_before = A.lm[ii].lm_before; pr.num_oov = A.lm[ii].num_oov; pr.num_oov_before = A.lm[ii].num_oov_before; }
      for (int ci = part; ci < V; ci += P) {
        const double curp = srow[ci];
        unsigned long long* const slot = ukey + n + ci * n + ii;
        if (ci == blank) { *slot = kNoCandKey; A.npb[ii] = curp + full; continue; }   // :374-376 (prob_blank was -inf)
        const double val = curp + (ci == last ? ppb : full);                     // :383-385 / :389-391
        const int k = ctab[ii * V + ci];
        unsigned long long uk = kNoCandKey;
        if (k >= 0) {
          const int j = mapA.find(k);
          if (j >= 0) A.inc[j] = val;            // the child is a beam member: its share from this parent
          // else: alive but pruned (Q7) -- the probability is lost and the slot stays taken
        } else {
          LmFields nl;
          LmAnswer ans; ans.sc = 0.f; ans.wi = 0u;
          if (LM) ans = lmcA[ii * V + ci];
          child_score_fields<LM>(p, pr, last, ci, ans, nl);
          const double sc = beam_score<LM>(p, val, ninf(), nl);                  // after next_step: prev_pnb = val, prev_pb = -inf
          uk = okey(sc);
          key_hi = max(key_hi, (unsigned)(uk >> 32));
          my_new++;
        }
        *slot = uk;
      }
    }
    {
      const int incl = wave_scan_i(my_new);
      if (lane == 63 && incl) atomicAdd(&s_total_new, incl);
    }
    key_hi = (unsigned)wave_max_i((int)(key_hi ^ 0x80000000u)) ^ 0x80000000u;     // (signed max on biased values)
    if (lane == 0) atomicMax(&s_hi, key_hi);
    lds_barrier();
    BPROF(1);
    // members: repeated-character share (:386-387), next_step (:337-342), score
    for (int i = tid; i < n; i += kThreads) {
      const int lc = A.last[i];
      double pnb = A.inc[i];
      if (lc >= 0 && lc != blank) pnb = lse2(pnb, srow[lc] + A.ppnb[i]);
      A.npnb[i] = pnb;
      const double nf = lse2(pnb, A.npb[i]);            // the score's log-sum-exp is next step's `full` if the member stays
      A.nfull[i] = nf;
      const double sc = beam_score_full<LM>(p, nf, A.lm[i]);
      const unsigned long long uk = okey(sc);
      ukey[i] = uk;
      const unsigned h32 = (unsigned)(uk >> 32);
      key_hi = max(key_hi, h32); key_lo = min(key_lo, h32);
    }
    if (wid < (n + 63) / 64) {
      key_hi = (unsigned)wave_max_i((int)(key_hi ^ 0x80000000u)) ^ 0x80000000u;
      key_lo = ~((unsigned)wave_max_i((int)((~key_lo) ^ 0x80000000u)) ^ 0x80000000u);
      if (lane == 0) { atomicMax(&s_hi, key_hi); atomicMin(&s_lo, key_lo); }
    }
    {
      // housekeeping for the phases that follow, by the waves without members (all waves if every wave has some): the
      // member update above is two dependent f64 log-sum-exps on two waves, the other fourteen would only wait
      const int mw = (n + 63) >> 6, nw = kThreads / 64;
      const int ctid = mw < nw ? tid - 64 * mw : tid, cstride = 64 * (mw < nw ? nw - mw : nw);
      if (ctid == 0) { s_hi2[ps ^ 1] = 0u; s_lo2[ps ^ 1] = 0xffffffffu; s_total_new2[ps ^ 1] = 0; s_nnew2[ps ^ 1] = 0; }
      if (ctid >= 0) {
        for (int h = ctid; h < p.HS; h += cstride) mapB.key[h] = -1;
        for (int e = ctid; e < W * V; e += cstride) ctabB[e] = -1;
        for (int h = ctid; h < kSelBins; h += cstride) hist[h] = 0;      // (the second histogram is cleared by the pass before it)
      }
    }
    lds_barrier();
    BPROF(2);
    const int total_new = s_total_new;
    const int nreal = n + total_new;                // candidates that exist
    const int ntot = n + npairs;                    // entries of key[]
    const int nsel = nreal > W ? W : nreal;
    int* const sel = hist + kSelBins;               // (no pruning only) the candidates in beam order
    // (LM: scratch of the answer phase, in regions that are dead by then -- the first histogram, the gathered candidates,
    // the candidate keys)
    int* const newlist = sidx;                                                        // new members that have to ask
    int* const ldr = reinterpret_cast<int*>(skey);                                    // leader of member j
    unsigned long long* const tsig = reinterpret_cast<unsigned long long*>(hist);     // open addressing: signature -> smallest rank
    // candidate d takes place j of the new beam (the other member set): a member that stays is copied, a pair becomes a
    // prefix -- node, LM state, its own guard.  Called by the thread that has just worked out the candidate's rank.
    // ... in two halves that touch different fields of the new member: what the lattice needs (probabilities, node, guard,
    // slot map) and the LM state -- with a language model they run on different waves (below).
    auto place_core = [&](int j, int d) {
      if (d < n) {
        const int i = d;
        A.kept[i] = 1; A.newpos[i] = j; Bm.from[j] = i;
        Bm.ppb[j] = A.npb[i]; Bm.ppnb[j] = A.npnb[i]; Bm.full[j] = A.nfull[i];
        Bm.node[j] = A.node[i]; Bm.last[j] = A.last[i];
        Bm.gown[j] = A.gown[i]; Bm.gchar[j] = A.gchar[i]; Bm.gnode[j] = A.gnode[i];     // (owner: position in A, for now)
        mapB.insert(A.node[i], j);
      } else {
        const int q = d - n;
        const int c = n > 1 ? (int)__umulhi((unsigned)q, lay_nmagic) : q, i = q - c * n;      // (q / n: exact for q * n < 2^32)
        const double val = srow[c] + (c == A.last[i] ? A.ppb[i] : A.full[i]);
        int k = atomicAdd(&s_next_node, 1);                                       // make_shared<Prefix>, :254
        if (k >= p.NCAP) { s_err = 1; k = 0; }
        else {
          BeamNode nn;
          nn.parent = A.node[i]; nn.last_char = c;
          nodes[k] = nn;                                                          // (fire and forget)
          mapB.insert(k, j);
        }
        Bm.ppb[j] = ninf(); Bm.ppnb[j] = val; Bm.full[j] = lse2(val, ninf()); Bm.node[j] = k; Bm.last[j] = c;
        Bm.gown[j] = i; Bm.gchar[j] = c; Bm.gnode[j] = k;                         // its own guard, if its parent stays
        Bm.from[j] = -1;
      }
    };
    auto place_lm = [&](int j, int d) {
      if (d < n) {
        copy_lm<LM>(Bm.lm[j], A.lm[d]);
      } else {
        const int q = d - n;
        const int c = n > 1 ? (int)__umulhi((unsigned)q, lay_nmagic) : q, i = q - c * n;
        LmAnswer ans; ans.sc = 0.f; ans.wi = 0u;
        if (LM) ans = lmcA[i * V + c];
        child_lm<LM>(p, lt, A.lm[i], A.last[i], c, ans, Bm.lm[j]);                 // (straight into LDS: a local LmFields lives in scratch)
      }
    };
    auto place = [&](int j, int d) { place_core(j, d); place_lm(j, d); };
    if (nreal > W) {                                                             // :405-415
      // ---- radix select of the W-th largest score on the order-preserving 64-bit key, 11 bits per pass ----
      // Where to start: the threshold lies between the smallest score of a full beam's old members (W candidates are
      // at least that good) and the largest score of all, so it shares their common leading bits (s_hi / s_lo).
      // Starting below them, the first digit already spreads the candidates that matter over the bins (a pass over
      // the sign/exponent bits would put all of them into one): 1.01 passes per step measured.  A pass stops the
      // search as soon as the bin of the chosen digit holds exactly the remaining k (taken whole) or at most 64
      // candidates (ranked exactly below).  Two histograms alternate; both start the step cleared.
      const unsigned H32 = s_hi, L32 = n == W ? s_lo : 0u;
      const unsigned xdiff = H32 ^ L32;
      const int hb = xdiff ? 63 - __builtin_clz(xdiff) : 31;                    // highest bit that may differ
      unsigned long long mask = hb == 63 ? 0ULL : ~0ULL << (hb + 1);
      unsigned long long prefix = ((unsigned long long)H32 << 32) & mask;       // (decided bits: the same in every thread)
      int krem = W, bin = 0;
      bool done = false;
      int shift = hb + 1 - kSelBits;                                            // >= 21
      for (int pass = 0;; pass++) {
        int* hcur = hist + (pass & 1) * kSelBins;
        int* hnext = hist + ((pass & 1) ^ 1) * kSelBins;
        const int nbits = 64 - __builtin_popcountll(mask) - shift;                    // (the last digit may be short)
        const int width = nbits < kSelBits ? nbits : kSelBits;
        const unsigned long long dmask = (1ULL << width) - 1ULL;
        for (int d = tid; d < ntot; d += kThreads) {
          const unsigned long long u = ukey[d];
          if ((u & mask) == prefix && u != kNoCandKey) atomicAdd(&hcur[(int)((u >> shift) & dmask)], 1);
        }
        lds_barrier();
        // the digit where the running count (from the top) reaches krem: every thread owns kPer adjacent bins
        // (one wave walking 32 bins per lane paid a 64-way bank conflict on every read)
        constexpr int kPer = kSelBins / kThreads;
        static_assert(kPer * kThreads == kSelBins && kPer >= 1, "bins per thread");
        const int top = kSelBins - 1 - kPer * tid;
        int cnt[kPer], mine = 0;
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) { cnt[jj] = hcur[top - jj]; mine += cnt[jj]; }
        const int inc = wave_scan_i(mine);
        if (lane == 63) s_part[wid] = inc;
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) hnext[top - jj] = 0;    // (at pass 0 this is the region that held the previous step's sel)
        lds_barrier();
        int above = inc - mine;                       // candidates with a larger digit than this thread's first
        {
          const int wtot = wave_scan_i(lane < kThreads / 64 ? s_part[lane] : 0);
          if (wid > 0) above += __builtin_amdgcn_readlane(wtot, wid - 1);
        }
        const unsigned long long digit_mask = dmask << shift;
        if (above < krem && above + mine >= krem) {   // the threshold digit is one of this thread's
#pragma unroll
          for (int jj = 0; jj < kPer; jj++) {
            if (above + cnt[jj] >= krem) {
              s_krem = krem - above;
              s_prefix = (prefix & ~digit_mask) | ((unsigned long long)(top - jj) << shift);
              s_bin = cnt[jj];
              s_done = cnt[jj] == krem - above ? 1 : 0;     // the whole bin survives: no finer threshold needed
              break;
            }
            above += cnt[jj];
          }
        }
        mask |= digit_mask;
        lds_barrier();
        krem = s_krem; prefix = s_prefix; bin = s_bin; done = s_done != 0;
#ifdef E2E_BEAM_PROFILE
        if (b == 0 && tid == 0) g_beam_prof[10] += 1;
#endif
        if (done || shift == 0 || bin <= kSelSmall) break;
        shift = shift - kSelBits > 0 ? shift - kSelBits : 0;
        lds_barrier();                                // (s_* are rewritten by the next pass)
      }
      BPROF(7);
      // Gathered for the final ranking: keys whose decided bits are above the threshold prefix, plus those equal to
      // it -- all of them if they are at most 64 (the ranking below keeps the best krem of them), else exactly krem:
      // the whole bin, or (every bit decided: exact ties) the first by position.
      const unsigned long long Tk = prefix;
      const bool small_bin = !done && shift != 0;           // (then bin <= kSelSmall and bin > krem)
#define OKEY_CMP(u) ((u) & mask)
      // ---- compaction: larger keys first, then the equal ones ----
      const int per = (ntot + kThreads - 1) / kThreads;
      const int d0 = min(tid * per, ntot), d1 = min(d0 + per, ntot);
      int ngt = 0, neq = 0;
      unsigned cls = 0;                              // two bits per candidate of this thread: 1 above, 2 in the threshold bin
      static_assert(kMaxCand <= 16 * kThreads, "classification bits per thread");
      for (int d = d0, sh2 = 0; d < d1; d++, sh2 += 2) {
        const unsigned long long uf = ukey[d], u = OKEY_CMP(uf);
        const bool gt = u > Tk, eq = u == Tk && uf != kNoCandKey;
        ngt += gt; neq += eq;
        cls |= (gt ? 1u : eq ? 2u : 0u) << sh2;
      }
      // (both counts ride in one integer -- at most 8192 candidates, 16 bits each -- through one wave scan, and the 16
      // waves' totals through one 16-lane scan instead of a 16-step loop in every wave)
      const int iboth = wave_scan_i(ngt | (neq << 16));
      if (lane == 63) s_part[wid] = iboth;
      lds_barrier();
      const int wtot = wave_scan_i(lane < kThreads / 64 ? s_part[lane] : 0);
      const int wbase = wid > 0 ? __builtin_amdgcn_readlane(wtot, wid - 1) : 0;
      const int tg = __builtin_amdgcn_readlane(wtot, kThreads / 64 - 1) & 0xffff;
      const int mine_excl = wbase + iboth - (ngt | (neq << 16));
      int og = mine_excl & 0xffff, oe = mine_excl >> 16;
      const int take = small_bin ? bin : krem;
      for (int d = d0; d < d1 && cls; d++, cls >>= 2) {
        if (cls & 1u) { uskey[og] = ukey[d]; sidx[og] = d; og++; }
        else if (cls & 2u) {
          if (oe < take) { uskey[tg + oe] = ukey[d]; sidx[tg + oe] = d; }
          oe++;
        }
      }
#undef OKEY_CMP
      const int M = tg + take;                      // W <= M <= W - 1 + kSelSmall gathered candidates
      lds_barrier();
      BPROF(8);
      // ---- rank the gathered candidates by (score desc, position asc): eight lanes count for one candidate; rank < W
      //      is the candidate's place in the new beam (the surplus of a small threshold bin falls off the end) ----
      if (LM) for (int h = tid; h < kStateSlots; h += kThreads) tsig[h] = 0ULL;
      for (int e0 = 0; e0 < M; e0 += kThreads / 8) {
        const int e = e0 + (tid >> 3), part = tid & 7;
        int cnt = 0;
        if (e < M) {
          // (the gathered list is in position order within its two parts, and the keys of the first part are all
          // larger than those of the second: among equal keys the earlier list index is the earlier position)
          const unsigned long long ke = uskey[e];
#pragma unroll 4
          for (int jj = part; jj < M; jj += 8) {
            const unsigned long long kj = uskey[jj];
            cnt += (kj > ke || (kj == ke && jj < e)) ? 1 : 0;
          }
        }
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x141, 0xf, 0xf, true);     // row_half_mirror: the other quad of the 8
        if (e < M && part == 0 && cnt < W) sel[cnt] = sidx[e];
      }
      // The members are built by W threads on two waves, not by the ranking threads (one lane in eight of all sixteen waves:
      // every wave walked the whole of place() -- with a language model some 400 instructions -- for eight useful lanes,
      // and the phase was bound by the instructions the four SIMDs had to issue).
      lds_barrier();
      if (LM) {
        // blocks of 128 threads alternate between the two halves: waves 0-1 the lattice half of members 0..127, waves 2-3
        // their LM half, ...
        const int blk = tid >> 7, j0 = (tid & 127) + 128 * (blk >> 1);
        if (blk & 1) { for (int j = j0; j < nsel; j += kThreads / 2) place_lm(j, sel[j]); }
        else { for (int j = j0; j < nsel; j += kThreads / 2) place_core(j, sel[j]); }
      } else {
        for (int j = tid; j < nsel; j += kThreads) place(j, sel[j]);
      }
    } else {
      // nothing is pruned (the first steps of an utterance): old members, then the pairs that exist, in order
      for (int j = tid; j < n; j += kThreads) sel[j] = j;
      const int chunk = (npairs + kThreads - 1) / kThreads;
      const int q0 = min(tid * chunk, npairs), q1 = min(q0 + chunk, npairs);
      int mine = 0;
      for (int q = q0; q < q1; q++) mine += ukey[n + q] != kNoCandKey;
      const int incl = wave_scan_i(mine);
      if (lane == 63) s_part[wid] = incl;
      lds_barrier();
      int pos = n + incl - mine;
      for (int w = 0; w < wid; w++) pos += s_part[w];
      for (int q = q0; q < q1; q++)
        if (ukey[n + q] != kNoCandKey) sel[pos++] = n + q;
      if (LM) for (int h = tid; h < kStateSlots; h += kThreads) tsig[h] = 0ULL;
      lds_barrier();
      if (LM) {
        // blocks of 128 threads alternate between the two halves: waves 0-1 the lattice half of members 0..127, waves 2-3
        // their LM half, ...
        const int blk = tid >> 7, j0 = (tid & 127) + 128 * (blk >> 1);
        if (blk & 1) { for (int j = j0; j < nsel; j += kThreads / 2) place_lm(j, sel[j]); }
        else { for (int j = j0; j < nsel; j += kThreads / 2) place_core(j, sel[j]); }
      } else {
        for (int j = tid; j < nsel; j += kThreads) place(j, sel[j]);
      }
    }
    lds_barrier();
    BPROF(3);
    // ---- guards and child tables of the new beam ----
    // A guard whose owner left the beam is inherited from the owner's guard (the next alive prefix up the path whose
    // parent was a member), until an owner that stays is found or the path runs out.  Every alive child of a member is
    // some member's guard: writing the guards into the cleared tables reproduces exactly the entries whose weak_ptr has
    // not expired upstream.
    for (int j = tid; j < nsel; j += kThreads) {
      Bm.inc[j] = ninf(); Bm.kept[j] = 0;                      // (what the next step's pair loop expects to find)
      int go = Bm.gown[j], gc = Bm.gchar[j], gn = Bm.gnode[j];
      while (go >= 0 && !A.kept[go]) { const int o = go; go = A.gown[o]; gc = A.gchar[o]; gn = A.gnode[o]; }
      if (go >= 0) {
        const int o = A.newpos[go];
        Bm.gown[j] = o; Bm.gchar[j] = gc; Bm.gnode[j] = gn;
        ctabB[o * V + gc] = gn;
      } else {
        Bm.gown[j] = -1;
      }
    }
    if (LM) {
      // The LM's answers for the new beam: carried over with a member that stays, asked for a member that is new -- but
      // only once per LM STATE.  What is asked depends on the member's state only (the word begun, the context, whether
      // a word starts), and most new members share theirs with another member: prefixes that differ further back than
      // the model looks (measured: 70 new members per step, 22 distinct states).  A wave walks the whole query code as
      // soon as one of its lanes has a query, so the queries are packed: one (asking member, character) per thread.
      //   1. every member looks its state's signature up in a small hash table: a member that stays enters itself (its
      //      answers are already there), a new member follows the holder if there is one and it really has the same state
      //      (the signature is a filter), else it enters itself, asks, and joins the packed list;
      //   2. rows are copied (staying members) / asked (list);  3. followers copy their leader's row.
      BPROF(11);
      auto state_of = [&](int j2, bool& nw, int& cn, unsigned long long& wh) {
        const LmFields& m = Bm.lm[j2];
        nw = m.num_words == 0 || Bm.last[j2] == p.space_id;
        cn = nw ? m.st_n : m.stb_n;
        wh = nw ? 0ULL : m.word_hash;
      };
      // One pass, on the threads next to the ones that walk the guards above (tid ^ 128: the two loops are independent).  A
      // table word is the state's signature with the holder's position in its low 12 bits.  A member that stays makes
      // itself the holder (smallest position wins); a new member takes a free word -- it is then the one that asks -- or
      // finds a holder and, if that really has the same state, follows it.  Whoever holds a state when a new member comes
      // by will have its row in time (copied below, or asked in the next phase); which of several equal-state members
      // that is changes nothing but who asks.
      for (int j2 = tid ^ 128; j2 < nsel; j2 += kThreads) {
        bool nw; int cn; unsigned long long wh;
        state_of(j2, nw, cn, wh);
        unsigned long long h = wh;
        for (int q2 = 0; q2 < cn; q2++) h = ng_mix(h, nw ? Bm.lm[j2].st[q2] : Bm.lm[j2].stb[q2]);
        h = ng_mix(h, (uint32_t)cn + (nw ? 100u : 0u));
        unsigned long long hk = h & ~0xFFFULL;
        if (hk == 0) hk = 0x1000ULL;
        const bool stays = Bm.from[j2] >= 0;
        int leader = j2;
        for (unsigned sl = (unsigned)(h >> 20) & (kStateSlots - 1);; sl = (sl + 1) & (kStateSlots - 1)) {
          const unsigned long long old = atomicCAS(&tsig[sl], 0ULL, hk | (unsigned long long)j2);
          if (old == 0ULL) break;                              // the word is this member's
          if ((old & ~0xFFFULL) != hk) continue;
          if (stays) { atomicMin(&tsig[sl], hk | (unsigned long long)j2); break; }
          const int o = (int)(old & 0xFFFULL);
          bool nw2; int cn2; unsigned long long wh2;
          state_of(o, nw2, cn2, wh2);
          bool same = nw2 == nw && cn2 == cn && wh2 == wh;
          for (int q2 = 0; q2 < cn && same; q2++)
            same = (nw ? Bm.lm[o].st[q2] : Bm.lm[o].stb[q2]) == (nw ? Bm.lm[j2].st[q2] : Bm.lm[j2].stb[q2]);
          if (same) { leader = o; break; }                    // (else: another state behind the same signature bits -- next word)
        }
        if (!stays && leader == j2) {
          newlist[atomicAdd(&s_nnew, 1)] = j2;
#ifdef E2E_BEAM_PROFILE
          if (b == 0) { const int q9 = atomicAdd(&g_beam_nsig, 1); if (q9 < (1 << 17)) g_beam_sigs[q9] = h; }
#endif
        }
        ldr[j2] = leader;
      }
      // (the rows of the members that stay: the waves behind those two copy them meanwhile)
      for (int e = tid - 256; e >= 0 && e < nsel * V; e += kThreads - 256) {
        const int j2 = div_v(e);
        const int f = Bm.from[j2];
        if (f >= 0) lmcB[e] = lmcA[f * V + (e - j2 * V)];
      }
      lds_barrier();
      BPROF(13);
      const int nnew = s_nnew;
      for (int t2 = tid; t2 < nnew * V; t2 += kThreads) {
        const int r = div_v(t2), c = t2 - r * V, j2 = newlist[r];
        if (c == blank || c == p.space_id) continue;
        lmcB[j2 * V + c] = lm_query<LMK == 2>(p, lt, Bm.lm[j2], Bm.last[j2], c);
      }
      lds_barrier();
      BPROF(14);
      for (int e = tid; e < nsel * V; e += kThreads) {
        const int j2 = div_v(e);
        const int o = ldr[j2];
        if (o != j2) lmcB[e] = lmcB[o * V + (e - j2 * V)];
      }
    }
    if (tid < V) srow2[((t + 1) & 1) * V + tid] = next_lp;          // (the other half of the double buffer: nobody reads it this step)
    lds_barrier();
    BPROF(5);
    n = nsel; cur ^= 1;
    if (s_err) break;
  }

  // ---- final sort (:418-424) reduces to the best prefix; its sentence (:232-245) ----
  {
    Members A;
    A.carve(mem0 + (size_t)cur * mbytes, W);
    for (int i = tid; i < n; i += kThreads) key[i] = beam_score<LM>(p, A.ppnb[i], A.ppb[i], A.lm[i]);
    __syncthreads();
  }
  int64_t* out = p.out + (int64_t)b * p.max_out;
  for (int64_t i = tid; i < p.max_out; i += kThreads) out[i] = 0;
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    int bi = 0;
    for (int i = 1; i < n; i++) if (key[i] > key[bi]) bi = i;            // first maximum = (score desc, position asc)
    Members A;
    A.carve(mem0 + (size_t)cur * mbytes, W);
    const int best = A.node[bi];
    int64_t m = 0;
    for (int k = best; k >= 0; k = nodes[k].parent) if (k == best || nodes[k].parent >= 0) m++;
    int64_t at = m;
    for (int k = best; k >= 0; k = nodes[k].parent)
      if (k == best || nodes[k].parent >= 0) { at--; if (at < p.max_out) out[at] = nodes[k].last_char; }
    // in-band status (include/e2e_ctc.h): a length above max_out says "truncated, m were needed"; -1 = node pool
    // exhausted (the sentence is then that of the last completed step and must not be used)
    p.out_len[b] = s_err ? (int64_t)-1 : m;
  }
}

// ------------------------------------------------------------------------------------------------------
// The general form: any alphabet width
// ------------------------------------------------------------------------------------------------------
// The kernel above keeps everything that scales with beam_width * alphabet in LDS -- candidate keys, the members' child
// tables, the LM's answers -- which bounds the width by the alphabet (81 at V = 80, 7 at V = 1000).  This one is the same
// algorithm, phase by phase, with those three in HBM (L2-resident: 16-24 bytes per (member, character)):
//   * candidate keys: an array in the workspace, streamed by the pair loop (written) and by the radix select and the
//     gather (read); the select starts below the leading bits that the best score and the worst old member share, like
//     above, and runs on to the last bit (no small-bin finish);
//   * child tables: at most one alive child per member is ever recorded (its guard), so a step's table is a hash map
//     of <= W entries keyed by (member position, character) in LDS instead of W*V words;
//   * the LM's answers: rows in the workspace, copied with a member that stays, asked for a member that is new (every
//     new member asks for itself: no sharing between members of equal LM state);
//   * the frame's log-probabilities are read from the input where they are needed (no staging).
// Barriers are full (__syncthreads after a workgroup fence): data crosses threads through global memory here.
// Not tuned: it exists so that an alphabet the fast kernel cannot hold is decoded at all, on the device, with
// the same result (tests run the whole beam suite through it: E2E_BEAM_GENERAL=1).  beam_width <= kGenMaxW and what the
// maps and the selection's arrays leave of one workgroup's LDS (the member sets move to the workspace beyond ~256): 512.
constexpr int kGenMaxW = 1024;
constexpr int kGenThreads = 1024;

// Character pre-selection of the general kernel (no language model): capacity of the per-step character list and words of
// the character bitmap in LDS (alphabets beyond 2^16 columns take every character, as before).
__host__ __device__ inline int gen_list_cap(int W) { return 4 * W + 64; }
__host__ __device__ inline int gen_bitmap_words(int V) { return V <= 65536 ? (V + 31) / 32 : 1; }
// order-preserving 32-bit key of a float (larger value -> larger key; -0 < +0 does not matter here)
__device__ __forceinline__ unsigned okey32(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct GenParams {
  unsigned long long* gkey;      // [B][W + W*V]
  LmAnswer* lmc;                 // [B][2][W][V]   (null without a language model)
  unsigned char* gmem;           // [B][2][Members::bytes(W)]: both member sets, when they no longer fit LDS (else null)
  int CH;                        // child map slots (power of two >= 4W)
};

struct ChildMap {                // (member position * V + character) -> node id of the alive child
  int* key; int* val; int mask;
  __device__ static unsigned hash(int k) { return (unsigned)k * 2654435761u; }
  __device__ void insert(int k, int v) const {
    unsigned h = (hash(k) >> 7) & mask;
    while (atomicCAS(&key[h], -1, k) != -1) h = (h + 1) & mask;
    val[h] = v;
  }
  __device__ int find(int k) const {
    unsigned h = (hash(k) >> 7) & mask;
    for (;;) {
      const int kk = key[h];
      if (kk == k) return val[h];
      if (kk == -1) return -1;
      h = (h + 1) & mask;
    }
  }
};

__device__ __forceinline__ void gsync() { __threadfence_block(); __syncthreads(); }

template <typename IO, int LMK>
__global__ __launch_bounds__(kGenThreads) void ctc_beam_general_kernel(BeamParams p, GenParams g) {
  constexpr int kThreads = kGenThreads;
  constexpr bool LM = LMK != 0;
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int V = p.V, W = p.W, blank = p.blank;
  // ---- LDS carve-up ----
  unsigned char* q8 = smem;
  // the two member sets: LDS while they fit beside the rest, else the workspace (beams of several hundred hypotheses:
  // every access to a member then goes to L2 -- slower per step, but upstream has no bound on the width at all)
  const size_t mbytes = Members::bytes(W);
  unsigned char* const mem0 = g.gmem ? g.gmem + (size_t)blockIdx.x * 2 * mbytes : q8;
  if (!g.gmem) q8 += 2 * mbytes;
  unsigned long long* uskey = (unsigned long long*)q8; q8 += sizeof(unsigned long long) * (p.WP2 + 8);
  double* fkey = (double*)q8; q8 += sizeof(double) * (p.WP2 + 8);          // final scores
  int* sidx = (int*)q8; q8 += sizeof(int) * (p.WP2 + 8);
  int* hist = (int*)q8; q8 += sizeof(int) * kSelBins;
  int* s_part = (int*)q8; q8 += sizeof(int) * 64;
  int* const sm0 = (int*)q8; q8 += sizeof(int) * 4 * p.HS;                  // [set][key | val][HS]  node -> position
  int* const cm0 = (int*)q8; q8 += sizeof(int) * 4 * g.CH;                  // [set][key | val][CH]  (position, char) -> child node
  int* const s_lst = (int*)q8; q8 += sizeof(int) * gen_list_cap(W);         // this step's characters, ascending
  unsigned* const s_bits = (unsigned*)q8; q8 += sizeof(unsigned) * gen_bitmap_words(V);   // one bit per character
  auto slot_map = [&](int set) { SlotMap m; m.key = sm0 + set * 2 * p.HS; m.val = m.key + p.HS; m.mask = p.HS - 1; return m; };
  auto child_map = [&](int set) { ChildMap m; m.key = cm0 + set * 2 * g.CH; m.val = m.key + g.CH; m.mask = g.CH - 1; return m; };
  __shared__ int s_next_node, s_err, s_krem, s_done, s_total_new, s_nl;
  __shared__ unsigned s_fmax, s_ckey;
  __shared__ unsigned s_hi, s_lo;
  __shared__ unsigned long long s_prefix;
  LabelTab lt; lt.off = p.lm.label_off; lt.bytes = p.lm.label_bytes; lt.one = nullptr;

  BeamNode* nodes = p.nodes + (size_t)b * p.NCAP;
  const IO* lp = reinterpret_cast<const IO*>(p.lp) + (int64_t)b * p.sB;
  unsigned long long* const gkey = g.gkey + (size_t)b * ((size_t)W + (size_t)W * V);
  LmAnswer* const lmc0 = LM ? g.lmc + (size_t)b * 2 * (size_t)W * V : nullptr;
  int64_t Tq = p.x_len[b];
  const int T = Tq < 0 ? 0 : (Tq > p.T ? p.T : (int)Tq);
  Members M0; M0.carve(mem0, W);

  for (int h = tid; h < 2 * p.HS; h += kThreads) { sm0[h] = -1; sm0[2 * p.HS + h] = -1; }
  for (int h = tid; h < 2 * g.CH; h += kThreads) { cm0[h] = -1; cm0[2 * g.CH + h] = -1; }
  if (tid == 0) {
    s_next_node = 1; s_err = 0;                                                     // node 0 is taken
    BeamNode& r = nodes[0];
    r.parent = -1; r.last_char = -1;
    LmFields l;
    l.lm_score = 0.0; l.lm_before = 0.0; l.num_words = 0; l.num_oov = 0; l.num_oov_before = 0; l.word_len = 0;
    l.word_hash = kFnvInit; l.st_n = 0; l.stb_n = 0;
    if (p.has_lm) { l.st[0] = p.lm.bos; l.st_n = 1; l.stb[0] = p.lm.bos; l.stb_n = 1; }
    M0.ppb[0] = 0.0; M0.ppnb[0] = ninf(); M0.full[0] = lse2(ninf(), 0.0); M0.inc[0] = ninf(); M0.kept[0] = 0; M0.node[0] = 0; M0.last[0] = -1; M0.gown[0] = -1; M0.gchar[0] = 0; M0.gnode[0] = 0;
    M0.lm[0] = l;
  }
  gsync();
  if (tid == 0) slot_map(0).insert(0, 0);
  if (LM) for (int c = tid; c < V; c += kThreads) if (c != blank && c != p.space_id) lmc0[c] = lm_query<LMK == 2>(p, lt, M0.lm[0], -1, c);
  gsync();
  int n = 1, cur = 0;
  // q / n for the candidate numbers of a step (q < n * V, q * n < 2^32) by one multiplication: ceil(2^32 / n), worked out when
  // the beam's size changes (a division per candidate was a third of the pair loop's instructions at ~30 candidates per thread)
  // (exact while q * n < 2^32; a beam of 512 over an alphabet of 32 000 without the pre-selection is beyond that: plain division)
  int magic_n = -1; unsigned magic = 0u; bool magic_ok = false;
  auto div_n = [&](int q) -> int { return magic_ok ? (int)__umulhi((unsigned)q, magic) : q / n; };

  for (int t = 0; t < T; t++) {
    if (n != magic_n) {
      magic_n = n; magic = n > 1 ? (unsigned)((0x100000000ULL + (unsigned)n - 1u) / (unsigned)n) : 0u;
      magic_ok = n > 1 && (unsigned long long)n * (unsigned long long)V * (unsigned long long)n < (1ULL << 32);
    }
    Members A, Bm;
    A.carve(mem0 + (size_t)cur * mbytes, W);
    Bm.carve(mem0 + (size_t)(cur ^ 1) * mbytes, W);
    const SlotMap mapA = slot_map(cur), mapB = slot_map(cur ^ 1);
    const ChildMap cmA = child_map(cur), cmB = child_map(cur ^ 1);
    const LmAnswer* const lmcA = LM ? lmc0 + (size_t)cur * W * V : nullptr;
    LmAnswer* const lmcB = LM ? lmc0 + (size_t)(cur ^ 1) * W * V : nullptr;
    const IO* const row = lp + (int64_t)t * p.sT;
    auto LP = [&](int c) -> double { return (double)row[(int64_t)c * p.sV]; };
    if (tid == 0) { s_hi = 0u; s_lo = 0xffffffffu; s_total_new = 0; }
    for (int h = tid; h < kSelBins; h += kThreads) hist[h] = 0;
    __syncthreads();
    // ---- which characters can matter this step (no language model) ----
    // Without a language model the score of a would-be prefix (i, c) is lp[c] + full_i - wip * words, monotone in lp[c] for a
    // given prefix except for c = the prefix's last character (it extends from ppb_i) and c = space (the word count).  A
    // prefix has at most m_i <= W characters whose child already lives (no candidate), so its candidate of rank > 2W in
    // lp order has >= W candidates of the same prefix before it in the reference's order (score desc, position asc: equal
    // lp, lower c first) and cannot be among the W survivors.  The step therefore scores, for every prefix, only
    // L = {the 2W largest lp[c], every c tied with the last of them} + {space} + {last_i}, in ascending c so that the order
    // of the keys stays the reference's: |L| * n <= (3W + 1) * W keys instead of V * n (10 k instead of 800 k at V = 8000,
    // W = 100).  Exact unless two DIFFERENT lp values could round to the same score, i.e. unless |full| is within 2^24 of
    // their spacing: f32 inputs and |full| < 2^20 only; otherwise, with a language model, for tiny alphabets or if the ties
    // overflow the list, every character is taken as before.
    bool pre = LMK == 0 && sizeof(IO) <= 4 && V <= 65536 && V - 1 > 3 * W + 2;      // (16-bit inputs are f32 numbers too)
    if (pre) {
      if (tid == 0) s_fmax = 0u;
      __syncthreads();
      unsigned fm = 0u;
      for (int i = tid; i < n; i += kThreads) fm = max(fm, __float_as_uint(fabsf((float)A.full[i])));
      fm = (unsigned)wave_max_i((int)fm);
      if (lane == 0 && fm) atomicMax(&s_fmax, fm);
      __syncthreads();
      pre = __uint_as_float(s_fmax) < 1048576.f;
    }
    int NL = V;
    if (pre) {
      // threshold: the key of the (2W)-th largest lp[c], c != blank -- radix select over V 32-bit keys, 11 bits per pass
      const int want = 2 * W;
      unsigned prefix = 0u, maskc = 0u;
      int krem = want;
      for (int pass = 0; pass < 3; pass++) {
        const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0, width = pass == 2 ? 10 : 11;
        for (int c = tid; c < V; c += kThreads) {
          if (c == blank) continue;
          const unsigned u = okey32((float)LP(c));
          if ((u & maskc) == prefix) atomicAdd(&hist[(int)((u >> shift) & ((1u << width) - 1u))], 1);
        }
        __syncthreads();
        constexpr int kPer = kSelBins / kThreads;
        const int top = kSelBins - 1 - kPer * tid;
        int cnt[kPer], mine = 0;
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) { cnt[jj] = hist[top - jj]; mine += cnt[jj]; }
        const int inc = wave_scan_i(mine);
        if (lane == 63) s_part[wid] = inc;
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) hist[top - jj] = 0;
        int above = inc - mine;
        {
          const int wtot = wave_scan_i(lane < kThreads / 64 ? s_part[lane] : 0);
          if (wid > 0) above += __builtin_amdgcn_readlane(wtot, wid - 1);
        }
        if (above < krem && above + mine >= krem) {
#pragma unroll
          for (int jj = 0; jj < kPer; jj++) {
            if (above + cnt[jj] >= krem) { s_krem = krem - above; s_ckey = prefix | ((unsigned)(top - jj) << shift); break; }
            above += cnt[jj];
          }
        }
        __syncthreads();
        krem = s_krem; prefix = s_ckey; maskc |= ((1u << width) - 1u) << shift;
        __syncthreads();
      }
      const unsigned Tk = prefix;                        // every c with key >= Tk is taken (ties with the 2W-th included)
      const int words = (V + 31) / 32;
      for (int wd = tid; wd < words; wd += kThreads) s_bits[wd] = 0u;
      __syncthreads();
      for (int c = tid; c < V; c += kThreads)
        if (c != blank && okey32((float)LP(c)) >= Tk) atomicOr(&s_bits[c >> 5], 1u << (c & 31));
      if (tid == 0 && p.space_id >= 0 && p.space_id < V && p.space_id != blank) atomicOr(&s_bits[p.space_id >> 5], 1u << (p.space_id & 31));
      for (int i = tid; i < n; i += kThreads) { const int lc = A.last[i]; if (lc >= 0 && lc != blank) atomicOr(&s_bits[lc >> 5], 1u << (lc & 31)); }
      __syncthreads();
      // the list, ascending: word wd's characters start at the number of bits set before it
      int run = 0;                                       // (bits set in the words of earlier rounds)
      for (int w0 = 0; w0 < words; w0 += kThreads) {
        const int wd = w0 + tid;
        const unsigned bits = wd < words ? s_bits[wd] : 0u;
        const int pc = __builtin_popcount(bits);
        const int inc = wave_scan_i(pc);
        if (lane == 63) s_part[wid] = inc;
        __syncthreads();
        int off = run + inc - pc, tot = 0;
        for (int w2 = 0; w2 < kThreads / 64; w2++) { if (w2 < wid) off += s_part[w2]; tot += s_part[w2]; }
        unsigned bb = bits;
        while (bb) { const int bit = __builtin_ctz(bb); bb &= bb - 1; if (off < gen_list_cap(W)) s_lst[off] = wd * 32 + bit; off++; }
        run += tot;
        __syncthreads();
      }
      if (tid == 0) s_nl = run;
      __syncthreads();
      NL = s_nl;
      if (NL > gen_list_cap(W)) { pre = false; NL = V; }        // (more ties than the list holds: every character)
    }
    auto CH = [&](int sidx) -> int { return pre ? s_lst[sidx] : sidx; };
    // ---- pairs: candidate q = c*n + i is the reference's order (character outer, prefix inner, :370-395) ----
    // (the blank creates no candidate: its share of every member is taken first)
    for (int i = tid; i < n; i += kThreads) A.npb[i] = LP(blank) + A.full[i];      // :374-376 (prob_blank was -inf)
    const int npairs = n * NL;
    int my_new = 0;
    unsigned key_hi = 0u, key_lo = 0xffffffffu;
    for (int e = tid; e < npairs; e += kThreads) {
      const int si = div_n(e), ii = e - si * n;        // (slot si of the list outer, prefix inner: the reference's order)
      const int ci = CH(si);
      const double full = A.full[ii], ppb = A.ppb[ii];
      const int last = A.last[ii];
      const double curp = LP(ci);
      unsigned long long* const slot = gkey + n + e;
      if (ci == blank) { *slot = kNoCandKey; continue; }
      const double val = curp + (ci == last ? ppb : full);                     // :383-385 / :389-391
      const int k = cmA.find(ii * V + ci);
      unsigned long long uk = kNoCandKey;
      if (k >= 0) {
        const int j = mapA.find(k);
        if (j >= 0) A.inc[j] = val;            // the child is a beam member: its share from this parent
        // else: alive but pruned (Q7) -- the probability is lost and the slot stays taken
      } else {
        LmFields pr, nl;
        pr.num_words = A.lm[ii].num_words;
        if (LM) { pr.lm_score = A.lm[ii].lm_score; pr.lm_before = A.lm[ii].lm_before; pr.num_oov = A.lm[ii].num_oov; pr.num_oov_before = A.lm[ii].num_oov_before; }
        LmAnswer ans; ans.sc = 0.f; ans.wi = 0u;
        if (LM) ans = lmcA[(size_t)ii * V + ci];
        child_score_fields<LM>(p, pr, last, ci, ans, nl);
        const double sc = beam_score<LM>(p, val, ninf(), nl);                  // after next_step: prev_pnb = val, prev_pb = -inf
        uk = okey(sc);
        key_hi = max(key_hi, (unsigned)(uk >> 32));
        my_new++;
      }
      *slot = uk;
    }
    {
      const int incl = wave_scan_i(my_new);
      if (lane == 63 && incl) atomicAdd(&s_total_new, incl);
    }
    gsync();
    // ---- members: repeated-character share (:386-387), next_step (:337-342), score ----
    for (int i = tid; i < n; i += kThreads) {
      const int lc = A.last[i];
      double pnb = A.inc[i];
      if (lc >= 0 && lc != blank) pnb = lse2(pnb, LP(lc) + A.ppnb[i]);
      A.npnb[i] = pnb;
      const double nf = lse2(pnb, A.npb[i]);
      A.nfull[i] = nf;
      const double sc = beam_score_full<LM>(p, nf, A.lm[i]);
      const unsigned long long uk = okey(sc);
      gkey[i] = uk;
      const unsigned h32 = (unsigned)(uk >> 32);
      key_hi = max(key_hi, h32); key_lo = min(key_lo, h32);
    }
    key_hi = (unsigned)wave_max_i((int)(key_hi ^ 0x80000000u)) ^ 0x80000000u;
    key_lo = ~((unsigned)wave_max_i((int)((~key_lo) ^ 0x80000000u)) ^ 0x80000000u);
    if (lane == 0) { atomicMax(&s_hi, key_hi); atomicMin(&s_lo, key_lo); }
    for (int h = tid; h < p.HS; h += kThreads) mapB.key[h] = -1;
    for (int h = tid; h < g.CH; h += kThreads) cmB.key[h] = -1;
    gsync();
    const int total_new = s_total_new;
    const int nreal = n + total_new;                // candidates that exist
    const size_t ntot = (size_t)n + (size_t)npairs; // entries of gkey[]
    const int nsel = nreal > W ? W : nreal;
    // candidate d takes place j of the new beam (see the fast kernel's `place`)
    auto place = [&](int j, int d) {
      if (d < n) {
        const int i = d;
        A.kept[i] = 1; A.newpos[i] = j; Bm.from[j] = i;
        Bm.ppb[j] = A.npb[i]; Bm.ppnb[j] = A.npnb[i]; Bm.full[j] = A.nfull[i];
        Bm.node[j] = A.node[i]; Bm.last[j] = A.last[i]; copy_lm<LM>(Bm.lm[j], A.lm[i]);
        Bm.gown[j] = A.gown[i]; Bm.gchar[j] = A.gchar[i]; Bm.gnode[j] = A.gnode[i];
        mapB.insert(A.node[i], j);
      } else {
        const int q = d - n;
        const int si = div_n(q), i = q - si * n;
        const int c = CH(si);
        const double val = LP(c) + (c == A.last[i] ? A.ppb[i] : A.full[i]);
        LmAnswer ans; ans.sc = 0.f; ans.wi = 0u;
        if (LM) ans = lmcA[(size_t)i * V + c];
        child_lm<LM>(p, lt, A.lm[i], A.last[i], c, ans, Bm.lm[j]);
        int k = atomicAdd(&s_next_node, 1);                                       // make_shared<Prefix>, :254
        if (k >= p.NCAP) { s_err = 1; k = 0; }
        else {
          BeamNode nn;
          nn.parent = A.node[i]; nn.last_char = c;
          nodes[k] = nn;
          mapB.insert(k, j);
        }
        Bm.ppb[j] = ninf(); Bm.ppnb[j] = val; Bm.full[j] = lse2(val, ninf()); Bm.node[j] = k; Bm.last[j] = c;
        Bm.gown[j] = i; Bm.gchar[j] = c; Bm.gnode[j] = k;
        Bm.from[j] = -1;
      }
    };
    // this thread's contiguous chunk of candidate positions (ordered passes below)
    const size_t per = (ntot + kThreads - 1) / kThreads;
    const size_t d0 = min((size_t)tid * per, ntot), d1 = min(d0 + per, ntot);
    if (nreal > W) {                                                             // :405-415
      // ---- radix select of the W-th largest key, 11 bits per pass, from the first bit that can differ ----
      const unsigned H32 = s_hi, L32 = n == W ? s_lo : 0u;
      const unsigned xdiff = H32 ^ L32;
      const int hb = xdiff ? 63 - __builtin_clz(xdiff) : 31;
      unsigned long long mask = hb == 63 ? 0ULL : ~0ULL << (hb + 1);
      unsigned long long prefix = ((unsigned long long)H32 << 32) & mask;
      int krem = W;
      bool done = false;
      int shift = hb + 1 - kSelBits;
      for (;;) {
        const int nbits = 64 - __builtin_popcountll(mask) - shift;
        const int width = nbits < kSelBits ? nbits : kSelBits;
        const unsigned long long dmask = (1ULL << width) - 1ULL;
        for (size_t d = tid; d < ntot; d += kThreads) {
          const unsigned long long u = gkey[d];
          if ((u & mask) == prefix && u != kNoCandKey) atomicAdd(&hist[(int)((u >> shift) & dmask)], 1);
        }
        __syncthreads();
        constexpr int kPer = kSelBins / kThreads;
        const int top = kSelBins - 1 - kPer * tid;
        int cnt[kPer], mine = 0;
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) { cnt[jj] = hist[top - jj]; mine += cnt[jj]; }
        const int inc = wave_scan_i(mine);
        if (lane == 63) s_part[wid] = inc;
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) hist[top - jj] = 0;
        int above = inc - mine;
        {
          const int wtot = wave_scan_i(lane < kThreads / 64 ? s_part[lane] : 0);
          if (wid > 0) above += __builtin_amdgcn_readlane(wtot, wid - 1);
        }
        const unsigned long long digit_mask = dmask << shift;
        if (above < krem && above + mine >= krem) {
#pragma unroll
          for (int jj = 0; jj < kPer; jj++) {
            if (above + cnt[jj] >= krem) {
              s_krem = krem - above;
              s_prefix = (prefix & ~digit_mask) | ((unsigned long long)(top - jj) << shift);
              s_done = cnt[jj] == krem - above ? 1 : 0;     // the whole bin survives: no finer threshold needed
              break;
            }
            above += cnt[jj];
          }
        }
        mask |= digit_mask;
        __syncthreads();
        krem = s_krem; prefix = s_prefix; done = s_done != 0;
        if (done || shift == 0) break;
        shift = shift - kSelBits > 0 ? shift - kSelBits : 0;
        __syncthreads();
      }
      // ---- gather: keys above the threshold, then the first krem of those equal to it, in position order ----
      const unsigned long long Tk = prefix;
      int ngt = 0, neq = 0;
      for (size_t d = d0; d < d1; d++) {
        const unsigned long long uf = gkey[d], u = uf & mask;
        ngt += u > Tk; neq += (u == Tk && uf != kNoCandKey);
      }
      const int igt = wave_scan_i(ngt), ieq = wave_scan_i(neq);
      if (lane == 63) { s_part[wid] = igt; s_part[16 + wid] = ieq; }
      __syncthreads();
      int og = igt - ngt, oe = ieq - neq, tg = 0;
      for (int w2 = 0; w2 < kThreads / 64; w2++) { if (w2 < wid) { og += s_part[w2]; oe += s_part[16 + w2]; } tg += s_part[w2]; }
      for (size_t d = d0; d < d1; d++) {
        const unsigned long long uf = gkey[d], u = uf & mask;
        if (u > Tk) { uskey[og] = uf; sidx[og] = (int)d; og++; }
        else if (u == Tk && uf != kNoCandKey) { if (oe < krem) { uskey[tg + oe] = uf; sidx[tg + oe] = (int)d; } oe++; }
      }
      const int M = tg + krem;                      // == W
      __syncthreads();
      // ---- rank by (score desc, position asc): eight lanes count for one candidate ----
      for (int e0 = 0; e0 < M; e0 += kThreads / 8) {
        const int e = e0 + (tid >> 3), part = tid & 7;
        int cnt = 0;
        if (e < M) {
          const unsigned long long ke = uskey[e];
          for (int jj = part; jj < M; jj += 8) {
            const unsigned long long kj = uskey[jj];
            cnt += (kj > ke || (kj == ke && jj < e)) ? 1 : 0;
          }
        }
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, true);
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, true);
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x141, 0xf, 0xf, true);
        if (e < M && part == 0 && cnt < W) place(cnt, sidx[e]);
      }
    } else {
      // nothing is pruned: old members, then the pairs that exist, in order
      int mine = 0;
      for (size_t d = max(d0, (size_t)n); d < d1; d++) mine += gkey[d] != kNoCandKey;
      const int incl = wave_scan_i(mine);
      if (lane == 63) s_part[wid] = incl;
      __syncthreads();
      int pos = n + incl - mine;
      for (int w2 = 0; w2 < wid; w2++) pos += s_part[w2];
      for (size_t d = max(d0, (size_t)n); d < d1; d++)
        if (gkey[d] != kNoCandKey) sidx[pos++] = (int)d;
      for (int j = tid; j < n; j += kThreads) sidx[j] = j;
      __syncthreads();
      for (int j = tid; j < nsel; j += kThreads) place(j, sidx[j]);
    }
    gsync();
    // ---- guards and child map of the new beam (see the fast kernel) ----
    for (int j = tid; j < nsel; j += kThreads) {
      Bm.inc[j] = ninf(); Bm.kept[j] = 0;
      int go = Bm.gown[j], gc = Bm.gchar[j], gn = Bm.gnode[j];
      while (go >= 0 && !A.kept[go]) { const int o = go; go = A.gown[o]; gc = A.gchar[o]; gn = A.gnode[o]; }
      if (go >= 0) {
        const int o = A.newpos[go];
        Bm.gown[j] = o; Bm.gchar[j] = gc; Bm.gnode[j] = gn;
        cmB.insert(o * V + gc, gn);
      } else {
        Bm.gown[j] = -1;
      }
    }
    if (LM) {
      // the LM's answers for the new beam: rows copied with members that stay, asked for members that are new
      for (size_t e = tid; e < (size_t)nsel * V; e += kThreads) {
        const int j2 = (int)(e / V), c = (int)(e - (size_t)j2 * V);
        const int f = Bm.from[j2];
        if (f >= 0) lmcB[e] = lmcA[(size_t)f * V + c];
        else if (c != blank && c != p.space_id) lmcB[e] = lm_query<LMK == 2>(p, lt, Bm.lm[j2], Bm.last[j2], c);
      }
    }
    gsync();
    n = nsel; cur ^= 1;
    if (s_err) break;
  }

  // ---- final sort (:418-424) reduces to the best prefix; its sentence (:232-245) ----
  {
    Members A;
    A.carve(mem0 + (size_t)cur * mbytes, W);
    for (int i = tid; i < n; i += kThreads) fkey[i] = beam_score<LM>(p, A.ppnb[i], A.ppb[i], A.lm[i]);
    __syncthreads();
  }
  int64_t* out = p.out + (int64_t)b * p.max_out;
  for (int64_t i = tid; i < p.max_out; i += kThreads) out[i] = 0;
  gsync();
  if (tid == 0) {
    int bi = 0;
    for (int i = 1; i < n; i++) if (fkey[i] > fkey[bi]) bi = i;            // first maximum = (score desc, position asc)
    Members A;
    A.carve(mem0 + (size_t)cur * mbytes, W);
    const int best = A.node[bi];
    int64_t m = 0;
    for (int k = best; k >= 0; k = nodes[k].parent) if (k == best || nodes[k].parent >= 0) m++;
    int64_t at = m;
    for (int k = best; k >= 0; k = nodes[k].parent)
      if (k == best || nodes[k].parent >= 0) { at--; if (at < p.max_out) out[at] = nodes[k].last_char; }
    p.out_len[b] = s_err ? (int64_t)-1 : m;
  }
}

struct GenLayout { size_t gkey, lmc, gmem, total, lds; int CH; bool members_in_ws; };
GenLayout gen_layout(int B, int V, int W, int WP2, int HS, bool lm) {
  GenLayout l;
  l.CH = 256; while (l.CH < 4 * W) l.CH <<= 1;
  size_t o = 0;
  l.gkey = o; o += align_up((size_t)B * ((size_t)W + (size_t)W * V) * sizeof(unsigned long long), 256);
  l.lmc = o; if (lm) o += align_up((size_t)B * 2 * (size_t)W * V * sizeof(LmAnswer), 256);
  l.gmem = o; o += align_up((size_t)B * 2 * Members::bytes(W), 256);       // (used only when the member sets leave LDS)
  l.total = o;
  const size_t rest = (sizeof(unsigned long long) + sizeof(double) + sizeof(int)) * (size_t)(WP2 + 8) +
          sizeof(int) * (kSelBins + 64 + 4 * (size_t)HS + 4 * (size_t)l.CH + (size_t)gen_list_cap(W) + (size_t)gen_bitmap_words(V)) + 64;
  l.members_in_ws = rest + 2 * Members::bytes(W) > (size_t)kLdsBudget;
  l.lds = rest + (l.members_in_ws ? 0 : 2 * Members::bytes(W));
  return l;
}

struct BeamLayout { size_t nodes, total, lds; int NCAP, CMAX, WP2, HS; };

BeamLayout beam_layout(int B, int T, int V, int W, bool lm = false) {
  BeamLayout l;
  l.CMAX = W * V + W + 8;
  // live nodes: the beam and its ancestors (at most one root path of length <= T per member); at most W are created per step
  l.NCAP = W * (T + 3) + 8;
  l.WP2 = 64; while (l.WP2 < W) l.WP2 <<= 1;
  l.HS = 256; while (l.HS < 4 * W) l.HS <<= 1;
  l.lds = BeamLds::bytes(W, V, l.CMAX, l.WP2, l.HS, lm);
  size_t o = 0;
  l.nodes = o; o += align_up((size_t)B * l.NCAP * sizeof(BeamNode), 256);
  l.total = o;
  return l;
}

}  // namespace
}  // namespace e2e

// does one workgroup's LDS hold the beam of this width over this alphabet? (the fast kernel)
static bool beam_fits(int V, int W, bool lm) {
  if (V < 1 || W < 1) return false;
  if ((long long)W * V + W + 8 > kMaxCand || W > kSelBins) return false;
  if (lm && 2 * W > kStateSlots) return false;
  const BeamLayout l = beam_layout(1, 1, V, W, lm);
  return l.lds <= (size_t)kLdsBudget;
}
// ... the general kernel (everything that scales with the alphabet is in the workspace)
static bool gen_fits(int V, int W, bool lm) {
  if (V < 1 || W < 1 || W > kGenMaxW || (long long)W * V > 0x7fffffffLL / 2) return false;
  const BeamLayout l = beam_layout(1, 1, V, W, lm);
  return gen_layout(1, V, W, l.WP2, l.HS, lm).lds <= (size_t)kLdsBudget;
}

extern "C" int e2e_ctc_beam_max_width(int V, int with_lm) {
  if (V < 1) return 0;
  int lo = 0, hi = kSelBins;                    // the largest supported width (the tests are monotone in the width)
  while (lo < hi) {
    const int mid = (lo + hi + 1) / 2;
    if (beam_fits(V, mid, with_lm != 0) || gen_fits(V, mid, with_lm != 0)) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// which kernel a call takes (same decision in the workspace query and in the call)
static bool beam_takes_general(int V, int W, bool lm) {
  static const bool force_general = getenv("E2E_BEAM_GENERAL") != nullptr;      // (tests: the whole suite through the general kernel)
  const bool fast_ok = beam_fits(V, W, lm), gen_ok = gen_fits(V, W, lm);
  return gen_ok && (!fast_ok || force_general);
}

extern "C" size_t e2e_ctc_beam_workspace_bytes_lm(int B, int T, int V, int beam_width, int with_lm) {
  if (B < 0 || T < 1 || V < 1 || beam_width < 1) return 0;
  const BeamLayout l = beam_layout(B, T, V, beam_width, with_lm != 0);
  // the general kernel's share (candidate keys, the LM's answer rows, member sets of very wide beams) only when that kernel
  // will run, its LM rows only with a language model: 6.4 MB per utterance instead of 19 at V = 8000, W = 100 without one
  const size_t gen = beam_takes_general(V, beam_width, with_lm != 0)
                         ? gen_layout(B, V, beam_width, l.WP2, l.HS, with_lm != 0).total : 0;
  return l.total + gen + 256;
}
// (sized for a call with a language model, which every call without one fits as well)
extern "C" size_t e2e_ctc_beam_workspace_bytes(int B, int T, int V, int beam_width) {
  const size_t a = e2e_ctc_beam_workspace_bytes_lm(B, T, V, beam_width, 1), b = e2e_ctc_beam_workspace_bytes_lm(B, T, V, beam_width, 0);
  return a > b ? a : b;
}

extern "C" int e2e_ctc_beam(const void* lp, int dtype, int64_t sB, int64_t sT, int64_t sV,
                            const int64_t* x_len, int B, int T, int V, int blank,
                            int beam_width, int space_id, const e2e_lm* lm,
                            double lmwt, double wip, double oov_penalty,
                            int64_t* out, int64_t max_out, int64_t* out_len,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != E2E_F32 && dtype != E2E_F64 && !dtype_is_16bit(dtype)) { set_error("dtype must be E2E_F32, E2E_F64, E2E_F16 or E2E_BF16"); return E2E_ERR_ARG; }
  if (B < 0 || T < 1 || V < 1 || beam_width < 1 || max_out < 1) { set_error("bad sizes"); return E2E_ERR_ARG; }
  if (blank < 0 || blank >= V) { set_error("blank=%d outside [0,%d)", blank, V); return E2E_ERR_ARG; }
  if (B > 0 && (!lp || !x_len || !out || !out_len)) { set_error("null pointer argument"); return E2E_ERR_ARG; }
  if (lm && !lm->d_ng) { set_error("the language model has no device tables (it was loaded without a GPU)"); return E2E_ERR_HIP; }
  if (lm) {
    int cur = -1;
    E2E_HIP_CHECK(hipGetDevice(&cur), "hipGetDevice");
    if (cur != lm->device) {
      set_error("the language model's tables are on device %d but the call runs on device %d: load it once per device",
                lm->device, cur);
      return E2E_ERR_ARG;
    }
    // the kernel spells words with label_off[0..V]: the model must have been loaded with exactly this alphabet
    if ((int)lm->label_off.size() - 1 != V) {
      set_error("the language model was loaded with %d labels but the log-probabilities have %d columns",
                (int)lm->label_off.size() - 1, V);
      return E2E_ERR_ARG;
    }
  }
  const BeamLayout l = beam_layout(B, T, V, beam_width, lm != nullptr);
  const bool fast_ok = beam_fits(V, beam_width, lm != nullptr), gen_ok = gen_fits(V, beam_width, lm != nullptr);
  const bool general = beam_takes_general(V, beam_width, lm != nullptr);
  if (!fast_ok && !gen_ok) {
    set_error("beam_width = %d over an alphabet of %d%s: at most %d (one workgroup's LDS holds the beam)", beam_width, V,
              lm ? " with a language model" : "", e2e_ctc_beam_max_width(V, lm != nullptr));
    return E2E_ERR_UNSUPPORTED;
  }
  const GenLayout gl = gen_layout(B, V, beam_width, l.WP2, l.HS, lm != nullptr);
  uintptr_t base = reinterpret_cast<uintptr_t>(workspace);
  const uintptr_t aligned = (base + 255) & ~(uintptr_t)255;
  const size_t need = l.total + (general ? gl.total : 0);
  if (!workspace || workspace_bytes < need + (aligned - base)) { set_error("workspace too small: need %zu", need + 256); return E2E_ERR_WORKSPACE; }
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
  p.nodes = reinterpret_cast<BeamNode*>(ws + l.nodes);
  p.NCAP = l.NCAP; p.CMAX = l.CMAX; p.WP2 = l.WP2; p.HS = l.HS;
  hipStream_t s = (hipStream_t)stream;
  const bool fast_lm = lm && lm->d_ngs && lm->order - 1 <= kParCtx;
  if (general) {
    GenParams g;
    g.gkey = reinterpret_cast<unsigned long long*>(ws + l.total + gl.gkey);
    g.lmc = lm ? reinterpret_cast<LmAnswer*>(ws + l.total + gl.lmc) : nullptr;
    g.gmem = gl.members_in_ws ? reinterpret_cast<unsigned char*>(ws + l.total + gl.gmem) : nullptr;
    g.CH = gl.CH;
    // (16-bit log-probabilities are read as they are -- every one of them is an f32 number, so the search is the f32 one's, bit for bit)
#define E2E_GEN_OF(IO) (!lm ? (const void*)&ctc_beam_general_kernel<IO, 0> : fast_lm ? (const void*)&ctc_beam_general_kernel<IO, 2> : (const void*)&ctc_beam_general_kernel<IO, 1>)
    const void* gfn = dtype == E2E_F32 ? E2E_GEN_OF(float) : dtype == E2E_F64 ? E2E_GEN_OF(double) : dtype == E2E_F16 ? E2E_GEN_OF(f16_t) : E2E_GEN_OF(bf16_t);
#undef E2E_GEN_OF
    E2E_HIP_CHECK(allow_dynamic_lds(gfn, (int)gl.lds), "hipFuncSetAttribute");
    void* gargs[] = { &p, &g };
    E2E_HIP_CHECK(hipLaunchKernel(gfn, dim3(B), dim3(kGenThreads), gargs, gl.lds, s), "ctc_beam_general_kernel launch");
    E2E_HIP_CHECK(hipGetLastError(), "ctc_beam_general_kernel launch");
    return E2E_OK;
  }
#define E2E_BEAM_OF(IO) (!lm ? (const void*)&ctc_beam_kernel<IO, 0, kThreadsNoLm> : fast_lm ? (const void*)&ctc_beam_kernel<IO, 2, kThreadsLm> : (const void*)&ctc_beam_kernel<IO, 1, kThreadsLm>)
  const void* fn = dtype == E2E_F32 ? E2E_BEAM_OF(float) : dtype == E2E_F64 ? E2E_BEAM_OF(double) : dtype == E2E_F16 ? E2E_BEAM_OF(f16_t) : E2E_BEAM_OF(bf16_t);
#undef E2E_BEAM_OF
  const int nthreads = lm ? kThreadsLm : kThreadsNoLm;
  E2E_HIP_CHECK(allow_dynamic_lds(fn, (int)l.lds), "hipFuncSetAttribute");
  void* args[] = { &p };
  E2E_HIP_CHECK(hipLaunchKernel(fn, dim3(B), dim3(nthreads), args, l.lds, s), "ctc_beam_kernel launch");
  E2E_HIP_CHECK(hipGetLastError(), "ctc_beam_kernel launch");
  return E2E_OK;
}

#ifdef E2E_BEAM_PROFILE
extern "C" int e2e_debug_beam_sigs(unsigned long long* host, int cap) {
  int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_beam_nsig), sizeof(int)) != hipSuccess) return -1;
  if (n > cap) n = cap;
  if (n > (1 << 17)) n = 1 << 17;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_beam_sigs), sizeof(unsigned long long) * n) != hipSuccess) return -1;
  const int zero = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_beam_nsig), &zero, sizeof(int));
  return n;
}
extern "C" int e2e_debug_beam_profile(unsigned long long* host) {
  if (hipDeviceSynchronize() != hipSuccess) return E2E_ERR_HIP;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_beam_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : E2E_ERR_HIP;
}
#endif

// Beam search / LM entry points -- not implemented yet (return E2E_ERR_UNSUPPORTED).
#include "common.h"
using namespace e2e;
extern "C" {
int e2e_lm_load_arpa(const char*, const char* const*, int, int, e2e_lm** out) { if (out) *out = nullptr; set_error("LM not built"); return E2E_ERR_UNSUPPORTED; }
void e2e_lm_free(e2e_lm*) {}
int e2e_lm_order(const e2e_lm*) { return 0; }
uint32_t e2e_lm_word_index(const e2e_lm*, const char*) { return 0; }
double e2e_lm_score(const e2e_lm*, const uint32_t*, int, uint32_t) { return 0.0; }
size_t e2e_ctc_beam_workspace_bytes(int, int, int, int) { return 0; }
int e2e_ctc_beam(const void*, int, int64_t, int64_t, int64_t, const int64_t*, int, int, int, int, int, int, const e2e_lm*,
                 double, double, double, int64_t*, int64_t, int64_t*, void*, size_t, void*) {
  set_error("beam search not built"); return E2E_ERR_UNSUPPORTED;
}
}

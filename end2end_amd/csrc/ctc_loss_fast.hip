// Fast scaled CTC path -- placeholder until the scaled-lattice kernel lands:
// reports "unsupported" so that E2E_ALGO_AUTO resolves to the exact kernel.
#include "common.h"

namespace e2e {
bool fast_supported(int, int, int, int) { return false; }
size_t fast_workspace_bytes(int, int, int, int) { return 0; }
int launch_fast(const LossArgs&, bool) { set_error("fast CTC path not built"); return E2E_ERR_UNSUPPORTED; }
}  // namespace e2e

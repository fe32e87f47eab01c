"""How far are the packed-f32 chains (e2e_ctc_loss_opts.chains = E2E_CHAINS_F32) from the exact kernel, per regime?
Prints, per sweep mode, the largest |grad - exact| and the largest excess over the DEFAULT tolerance (2e-6 + 1e-4 |g|)
among utterances the fast path kept, for long-target shapes (the option only changes utterances with > 127 labels)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for mode, sharps in (("smooth", [0.1, 1.0]), ("normal", [1.0, 3.0]), ("sharp", [5.0, 8.0]), ("consistent", [4.0, 8.0])):
    worst = 0.0; worst_excess = -1.0; kept = 0; over = 0; worst_loss = 0.0
    for case in range(12):
        B = 8; T = int(rng.choice([400, 1000, 2000])); V = int(rng.choice([29, 48])); S = int(rng.integers(128, min(255, T // 2) + 1))
        sharp = float(rng.choice(sharps))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(B, T, V, generator=g) * sharp
        tg = torch.tensor(rng.integers(1, V, size=(B, S)), dtype=torch.long)
        tl = torch.tensor(rng.integers(128, S + 1, size=B)); xl = torch.full((B,), T)
        if mode == "consistent":            # emissions that agree with the targets (a trained model): blank / label peaks along an alignment
            for b in range(B):
                pos = np.sort(rng.choice(T, size=int(tl[b]), replace=False))
                x[b, :, 0] += sharp * 2
                for i, t in enumerate(pos): x[b, t, int(tg[b, i])] += sharp * 4
        le, ge = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_EXACT)
        lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST, chains=_lib.CHAINS_F32)
        for b in range(B):
            if np.isnan(lf[b]) or not np.isfinite(le[b]): continue
            kept += 1
            d = np.abs(gf[b].astype(np.float64) - ge[b]); tol = 2e-6 + 1e-4 * np.abs(ge[b])
            worst = max(worst, float(d.max())); ex = float((d - tol).max()); worst_excess = max(worst_excess, ex); over += ex > 0
            worst_loss = max(worst_loss, abs(float(lf[b]) - float(le[b])) / max(1.0, abs(float(le[b]))))
    print("%-10s kept %3d utterances: max |dgrad| %.2e, max excess over the default tolerance %.2e (%d utterances over), max rel dloss %.1e" % (mode, kept, worst, worst_excess, over, worst_loss))

"""How often does the fast path give up (flag -> exact fallback) on PEAKY but CONSISTENT emissions, i.e. what a trained
acoustic model produces: logits that favour a valid alignment of the utterance's own targets by `boost` over unit noise?
(Random targets against sharp random logits -- tools/diag/fuzz_fast_vs_exact.py -- are a different, unrealistic regime.)"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
rng = np.random.default_rng(0)
def aligned_logits(B, T, V, S, boost, blank=0):
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(max(S // 2, 1), S + 1, size=B); xl = np.full(B, T)
    for b in range(B):
        L = int(tl[b])
        # a random monotone alignment: each label gets >= 1 frame, blanks fill the rest (forced between repeats)
        need = L + sum(1 for i in range(1, L) if tg[b, i] == tg[b, i - 1])
        assert need <= T
        slots = np.sort(rng.choice(T, size=L, replace=False)) if need == L else None
        path = np.full(T, blank)
        if slots is None:
            slots = np.arange(0, 2 * L, 2) + (T - 2 * L) // 2 if 2 * L <= T else None
        if slots is None: continue
        # avoid adjacent equal labels without a blank between them
        for i, s_ in enumerate(slots):
            path[s_] = tg[b, i]
        for i in range(1, L):
            if tg[b, i] == tg[b, i - 1] and slots[i] == slots[i - 1] + 1: path[slots[i]] = blank   # (drops a label: still informative)
        x[b, np.arange(T), path] += boost
    return torch.from_numpy(x), torch.from_numpy(tg), torch.from_numpy(xl), torch.from_numpy(tl)
for (B, T, V, S) in [(64, 1000, 29, 200), (64, 500, 29, 60), (32, 300, 64, 100)]:
    for boost in [0.0, 2.0, 5.0, 10.0, 20.0]:
        x, tg, xl, tl = aligned_logits(B, T, V, S, boost)
        lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST)
        le, ge = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_EXACT)
        ok = ~np.isnan(lf)
        err = np.abs(gf[ok].astype(np.float64) - ge[ok].astype(np.float64)).max() if ok.any() else float("nan")
        print("B=%d T=%d V=%d S<=%d boost %4.1f: mean loss %8.2f, flagged %3d of %d, max grad err of the rest %.1e" % (B, T, V, S, boost, float(np.nanmean(le)), int((~ok).sum()), B, err))

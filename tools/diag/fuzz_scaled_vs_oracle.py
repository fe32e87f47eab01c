"""Randomised parity sweep of the exact kernel's SCALED form (what E2E_ALGO_AUTO runs for f32 utterances the fast path hands
over -- here: a target equal to the blank id -- or cannot take -- targets beyond 447 labels) against the oracle: sharp
emissions up to scale 12, -inf log-probs, infeasible alignments, T from 1, ragged lengths, any blank id.
  python tools/diag/fuzz_scaled_vs_oracle.py [cases] [seed]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U, oracle_lib as O
from end2end_amd import _lib
WIDE = os.environ.get("FUZZ_WIDE") == "1"      # also alphabets that take the wide path
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    long_targets = case % 12 == 11
    B = int(rng.integers(1, 5))
    if long_targets:
        T = int(rng.integers(450, 700)); V = int(rng.choice([3, 29, 64])); Smax = int(rng.integers(448, min(T + 2, 600) + 1))
    else:
        T = int(rng.integers(1, 200)); V = int(rng.choice([2, 3, 5, 29, 64, 96, 200, 1500] if WIDE else [2, 3, 5, 29, 64, 96])); Smax = int(rng.integers(1, min(120, T + 3) + 1))
    fused = bool(rng.integers(0, 2)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * float(rng.choice([0.3, 1.0, 4.0, 8.0, 12.0]))
    lp = torch.log_softmax(x, -1)
    if not fused and rng.integers(0, 3) == 0: lp[:, ::4, int(rng.integers(0, V))] = float("-inf")
    labs = [v for v in range(V) if v != blank]
    tg = torch.tensor(rng.choice(labs, size=(B, max(Smax, 1))), dtype=torch.long)
    xl = torch.tensor(rng.integers(1, T + 1, size=B)); xl[0] = T
    tl = torch.tensor(rng.integers(0, Smax + 1, size=B)); tl[0] = Smax
    if not long_targets:
        for b in range(B):
            if tl[b] > 0: tg[b, int(rng.integers(0, int(tl[b])))] = blank
    inp = (x if fused else lp).float()
    ref_lp = torch.log_softmax(inp.double(), -1) if fused else inp.double()
    l_o, g_o = O.ctc_loss(ref_lp.numpy(), tg.numpy(), xl.numpy(), tl.numpy(), blank)
    if fused:
        for b in range(B):
            if np.isfinite(l_o[b]): g_o[b, int(xl[b]):] = 0.0          # (an infeasible utterance stays NaN everywhere, quirk Q2)
    lg, gg = U.c_abi_loss(inp, tg, xl, tl, blank, not fused, _lib.ALGO_AUTO)
    try:
        U.assert_same(lg, l_o, 1e-4, 2e-6, "losses"); U.assert_same(gg, g_o, 1e-4, 2e-6, "grads")
    except AssertionError as e:
        bad += 1
        print("MISMATCH case %d (B=%d T=%d V=%d S=%d fused=%d blank=%d long=%d): %s" % (case, B, T, V, Smax, fused, blank, long_targets, str(e).splitlines()[0][:120]))
print("%d cases, %d mismatches" % (n_cases, bad))

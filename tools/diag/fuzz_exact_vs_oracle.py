"""Randomised parity sweep: the exact GPU kernel (E2E_ALGO_EXACT, f32 and f64 inputs, log-probs or fused logits) against the
oracle on small random shapes, including -inf log-probs, infeasible alignments, blank anywhere, ragged lengths."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U, oracle_lib as O
from end2end_amd import _lib
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0; worst = 0.0
for case in range(n_cases):
    B = int(rng.integers(1, 6)); T = int(rng.integers(1, 90)); V = int(rng.choice([2, 3, 5, 29, 64, 97, 300]))
    Smax = int(rng.integers(0, min(60, T + 3) + 1))
    f64 = bool(rng.integers(0, 2)); fused = bool(rng.integers(0, 2)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * float(rng.choice([0.3, 1.0, 4.0]))
    lp = torch.log_softmax(x, -1)
    if not fused and rng.integers(0, 3) == 0: lp[:, ::4, int(rng.integers(0, V))] = float("-inf")
    labs = [v for v in range(V) if v != blank]
    tg = torch.tensor(rng.choice(labs, size=(B, max(Smax, 1))), dtype=torch.long)
    xl = torch.tensor(rng.integers(1, T + 1, size=B)); xl[0] = T
    tl = torch.tensor(rng.integers(0, Smax + 1, size=B)); tl[0] = Smax
    inp = (x if fused else lp).to(torch.float64 if f64 else torch.float32)
    ref_lp = torch.log_softmax(inp.double(), -1) if fused else inp.double()
    l_o, g_o = O.ctc_loss(ref_lp.numpy(), tg.numpy(), xl.numpy(), tl.numpy(), blank)
    if fused:
        for b in range(B):
            if np.isfinite(l_o[b]): g_o[b, int(xl[b]):] = 0.0          # (an infeasible utterance stays NaN everywhere, quirk Q2)
    lg, gg = U.c_abi_loss(inp, tg, xl, tl, blank, not fused, _lib.ALGO_EXACT)
    rt, at = (1e-9, 1e-12) if f64 else (1e-4, 2e-6)
    try:
        U.assert_same(lg, l_o, rt, at, "losses"); U.assert_same(gg, g_o, rt, at, "grads")
    except AssertionError as e:
        bad += 1
        print("MISMATCH case %d (B=%d T=%d V=%d S=%d f64=%d fused=%d blank=%d): %s" % (case, B, T, V, Smax, f64, fused, blank, str(e).splitlines()[0][:120]))
print("%d cases, %d mismatches" % (n_cases, bad))

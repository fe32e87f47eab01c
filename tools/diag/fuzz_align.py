"""Randomised parity sweep of the GPU forced alignment (e2e_ctc_align) against the oracle: random shapes, ragged lengths,
repeats, exact ties (rounded / constant emissions), -inf holes, too few frames, CTC and ASG, f32 / f64, blank positions."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import oracle_lib as O
from test_gpu_align import c_abi_align
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    B = int(rng.integers(1, 6)); T = int(rng.integers(1, 300)); V = int(rng.integers(2, 40))
    S = int(rng.integers(0, min(T, 120) + 1))
    is_ctc = bool(rng.integers(0, 2))
    blank = int(rng.choice([0, V - 1, rng.integers(0, V)])) if is_ctc else 0
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * float(rng.choice([0.0, 0.3, 1.0, 4.0]))
    style = int(rng.integers(0, 3))
    if style == 1: x = x.round()
    lp = torch.log_softmax(x, -1)
    if style == 2 and V > 2: lp[:, ::3, int(rng.integers(0, V))] = float("-inf")
    pool = [v for v in range(V) if v != blank] if is_ctc else list(range(V))
    tg = torch.tensor(rng.choice(pool, size=(B, max(S, 1))))
    if rng.integers(0, 2) and S > 1: tg[:, 1::2] = tg[:, 0::2][:, : tg[:, 1::2].shape[1]]
    tl = rng.integers(0 if is_ctc else 1, S + 1, size=B) if S > 0 else np.zeros(B, dtype=np.int64)
    if not is_ctc: tl = np.maximum(tl, 1)
    xl = rng.integers(1, T + 1, size=B); xl[0] = T
    if not is_ctc: xl = np.maximum(xl, np.minimum(tl, T))
    if rng.integers(0, 2): lp = lp.float()
    got = c_abi_align(lp, tg, xl, tl, blank, is_ctc)
    want = O.ctc_align(lp.double().numpy(), tg.numpy(), xl, tl, blank, is_ctc)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH case", case, dict(B=B, T=T, V=V, S=S, is_ctc=is_ctc, blank=blank, style=style, xl=xl.tolist(), tl=list(map(int, tl))))
print("cases", n_cases, "mismatches", bad)

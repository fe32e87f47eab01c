"""Randomised parity sweep of the GPU prefix beam search against the oracle (CPU restatement of the reference): random
shapes, beam widths, blank / space positions, word-insertion penalties, tie-heavy (rounded) and -inf-holed emissions,
ragged lengths, with and without the tiny 3-gram LM.  Every utterance must decode to exactly the oracle's label sequence."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U, oracle_lib as O
from end2end_amd.engines import LanguageModel
ARPA = os.path.join(root, "tests", "golden", "tiny_3gram.arpa")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    with_lm = bool(rng.integers(0, 4) == 0)
    B = int(rng.integers(1, 5)); T = int(rng.integers(1, 70))
    if with_lm:
        labels = ["_", "a", "b", " "]; V = 4; blank = 0
        cs = bool(rng.integers(0, 2))
        lm = LanguageModel(ARPA, labels, cs); olm = O.OracleLM(ARPA)
        kw = dict(lmwt=float(rng.choice([0.5, 1.0, 2.0])), wip=float(rng.choice([0.0, 1.0])), oov_penalty=float(rng.choice([-1000.0, -3.0])), case_sensitive=cs)
    else:
        V = int(rng.integers(2, 14)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
        alphabet = list("abcdefghijklm")
        labels = [alphabet[i] for i in range(V)]
        labels[blank] = "_"
        if V > 2 and rng.integers(0, 2): labels[int(rng.choice([i for i in range(V) if i != blank]))] = " "
        lm = olm = None
        kw = dict(wip=float(rng.choice([0.0, 1.0, 2.5])))
    W = int(rng.choice([1, 2, 3, 5, 16, 40, 64, 65, 100, 128, 200]))
    if W * V + W + 8 > 8192: W = 100
    sharp = float(rng.choice([0.3, 1.0, 3.0]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * sharp
    style = int(rng.integers(0, 4))
    if style == 1: x = x.round()                       # many exact ties
    if style == 2: x = torch.zeros_like(x)             # everything ties
    lp = torch.log_softmax(x, -1)
    if style == 3 and V > 2: lp[:, ::3, int(rng.integers(0, V))] = float("-inf")
    xl = rng.integers(1, T + 1, size=B).tolist(); xl[0] = T
    gpu_kw = {k: v for k, v in kw.items() if k != "case_sensitive"}
    if rng.integers(0, 2): lp_in = lp.float(); lp_ref = lp_in.double()
    else: lp_in = lp; lp_ref = lp
    only = os.environ.get("FUZZ_ONLY")
    if only is not None and case != int(only): continue          # (the random stream above is consumed all the same)
    ids, lens = U.c_abi_beam(lp_in, xl, blank, W, labels, lm, **gpu_kw)
    o_ids, o_lens, _ = O.ctc_beam(lp_ref.numpy(), xl, blank, W, labels, olm, **kw)
    if lens.tolist() != o_lens.tolist() or ids.tolist() != o_ids.tolist():
        # Uniform emissions make DIFFERENT prefixes exactly equiprobable (e.g. "a" and "aa" after 5 frames of p = 1/2:
        # 15 alignments each); which one wins then hangs on the last bit of exp/log, where the device's math library and
        # the host's differ.  Reported, but not counted as a parity failure.
        if style == 2: print("(mathematical tie between prefixes, decided by the last bit of libm: case %d)" % case); continue
        # ... and rounded logits on a tiny alphabet can do the same (all frames 1/2 : 1/2): without an LM, check whether
        # the two answers' exact CTC log-probabilities (full forward pass, same word-penalty terms) coincide
        if not with_lm:
            tie = True
            for b_ in range(B):
                if ids[b_, :lens[b_]].tolist() == o_ids[b_, :o_lens[b_]].tolist(): continue
                sa, sb = ids[b_, :lens[b_]], o_ids[b_, :o_lens[b_]]
                if (sa < 0).any() or (sb < 0).any() or " " in labels: tie = False; break
                n_ = int(xl[b_])
                la, _ = O.ctc_loss(lp_ref.numpy()[b_:b_ + 1, :n_], sa[None, :], [n_], [len(sa)], blank)
                lb, _ = O.ctc_loss(lp_ref.numpy()[b_:b_ + 1, :n_], sb[None, :], [n_], [len(sb)], blank)
                if len(sa) == 0 or len(sb) == 0 or abs(la[0] - lb[0]) > 1e-12 * max(1.0, abs(la[0])): tie = False; break
            if tie: print("(mathematical tie between prefixes, decided by the last bit of libm: case %d)" % case); continue
        # ... or cascade: with rounded emissions scores that are mathematically equal differ in the last bit between the
        # device's and the host's exp/log, a different candidate is pruned mid-way and the final answers differ without
        # being ties themselves.  Probe: emissions perturbed by ~1e-11 break the exact ties and change nothing else -- a
        # logic error survives that, a tie does not.
        if style == 1 and not with_lm:
            prng = np.random.default_rng(case)
            survived = 0
            for _ in range(8):
                qn = lp_ref.numpy() + prng.normal(size=tuple(lp_ref.shape)) * 1e-11
                i2, l2 = U.c_abi_beam(torch.from_numpy(qn), xl, blank, W, labels, lm, **gpu_kw)
                o2, ol2, _ = O.ctc_beam(qn, xl, blank, W, labels, olm, **kw)
                survived += (l2.tolist() != ol2.tolist() or i2.tolist() != o2.tolist())
            if survived == 0: print("(cascade of mathematical ties, gone under a 1e-11 perturbation: case %d)" % case); continue
        bad += 1
        if only is not None:
            for b_ in range(B):
                print("  utt %d (len %d): gpu %s | oracle %s" % (b_, xl[b_], ids[b_, :lens[b_]].tolist(), o_ids[b_, :o_lens[b_]].tolist()))
            np.save(os.path.join(root, "gpurun_out", "fuzz_case_lp.npy"), lp_ref.numpy())
        print("MISMATCH case %d: B=%d T=%d V=%d W=%d blank=%d style=%d lm=%d kw=%s xl=%s" % (case, B, T, V, W, blank, style, with_lm, kw, xl))
print("%d cases, %d mismatches" % (n_cases, bad))

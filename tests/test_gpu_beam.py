"""-m gpu: HIP prefix beam search through the C ABI against the reference tests' known answers and the oracle."""
import os

import numpy as np
import pytest
import torch

import golden_util as G
import oracle_lib as O
import gpu_util as U
from end2end_amd.engines import LanguageModel

pytestmark = pytest.mark.gpu
ARPA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_3gram.arpa")


def rand_lp(seed, B, T, V, sharp=2.0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    return torch.log_softmax(torch.randn(B, T, V, generator=g, dtype=torch.float64) * sharp, -1).to(dtype)


def same_as_oracle(lp, x_len, blank, W, labels, lm=None, olm=None, **kw):
    gpu_kw = {k: v for k, v in kw.items() if k != "case_sensitive"}     # the LM handle already carries it
    ids, lens = U.c_abi_beam(lp, x_len, blank, W, labels, lm, **gpu_kw)
    o_ids, o_lens, _ = O.ctc_beam(lp.double().numpy(), x_len, blank, W, labels, olm, **kw)
    assert lens.tolist() == o_lens.tolist()
    assert ids.tolist() == o_ids.tolist()
    return ids, lens


@pytest.mark.parametrize("case", [c for c in G.known_answers()["decode"] if "beam" in c], ids=lambda c: c["name"])
def test_known_answers(case):
    lp = torch.log(torch.tensor(case["x"], dtype=torch.float32))
    ids, lens = U.c_abi_beam(lp, None, case["blank"], case["beam_width"], case["labels"], wip=case["wip"])
    sent = ["".join(case["labels"][i] for i in ids[b, : lens[b]]) for b in range(len(lens))]
    assert sent == case["beam"]


@pytest.mark.parametrize("W", [2, 3, 10, 20, 100])
@pytest.mark.parametrize("wip", [0.0, 1.0])
def test_random_no_lm_matches_oracle(W, wip):
    labels = ["_", "a", "b", "c", " ", "d", "'"]
    lp = rand_lp(100 + W, 5, 40, 7)
    same_as_oracle(lp, [40, 33, 17, 40, 1], 0, W, labels, wip=wip)
    same_as_oracle(lp, [40, 33, 17, 40, 1], 6, W, labels[::-1], wip=wip)      # blank last, space elsewhere
    same_as_oracle(lp.float(), None, 0, W, None, wip=wip)                      # f32 input, no labels


@pytest.mark.parametrize("W", [4, 70, 100])
def test_massive_ties_are_broken_by_position_like_the_oracle(W):
    # constant / two-valued emissions: most candidates of a step have EXACTLY equal scores, so the selection's
    # tie handling (threshold bin taken whole, first-by-position among exact ties, the one-wave finish of a small bin)
    # decides which prefixes survive; the oracle keeps (score desc, position asc)
    labels = ["_", "a", "b", "c", "d", "e", "f", "g", "h", " "]
    V = len(labels)
    flat = torch.full((2, 9, V), float(np.log(1.0 / V)), dtype=torch.float64)
    same_as_oracle(flat, [9, 6], 0, W, labels, wip=0.0)
    two = torch.log(torch.tensor([0.3] + [0.7 / (V - 1)] * (V - 1), dtype=torch.float64)).repeat(2, 9, 1)
    same_as_oracle(two, [9, 7], 0, W, labels, wip=1.0)
    # ... and with impossible symbols (-inf) in the mix
    holes = flat.clone(); holes[:, :, 3] = float("-inf"); holes[:, ::2, 5] = float("-inf")
    same_as_oracle(holes, [9, 9], 0, W, labels, wip=0.0)


def test_speech_shape_beam100_matches_oracle():
    labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
    lp = rand_lp(7, 3, 120, 29, sharp=3.0)
    same_as_oracle(lp, [120, 90, 61], 0, 100, labels, wip=1.0)


def test_empty_prefix_winner_and_time_major_view():
    lp = torch.log(torch.tensor([[[0.98, 0.01, 0.01]] * 4], dtype=torch.float64))
    ids, lens = U.c_abi_beam(lp, None, 0, 10, None)
    assert ids.tolist() == [[-1]] and lens.tolist() == [1]                    # quirk Q6
    x = rand_lp(11, 4, 30, 6)
    tm = x.permute(1, 0, 2).contiguous().permute(1, 0, 2)
    a = U.c_abi_beam(x, None, 0, 8, None)
    b = U.c_abi_beam(tm, None, 0, 8, None)
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


@pytest.mark.parametrize("case_sensitive", [True, False])
@pytest.mark.parametrize("W,lmwt,wip,oov", [(10, 1.0, 0.0, -10.0), (30, 0.5, 1.0, -3.0), (100, 2.0, 0.0, -1000.0)])
def test_lm_scored_beam_matches_oracle(W, lmwt, wip, oov, case_sensitive):
    labels = ["_", "a", "b", " "] if case_sensitive else ["_", "A", "b", " "]
    lm = LanguageModel(ARPA, labels, case_sensitive)
    olm = O.OracleLM(ARPA)
    lp = rand_lp(31 + W, 4, 25, 4, sharp=1.5)
    same_as_oracle(lp, [25, 25, 18, 9], 0, W, labels, lm, olm, lmwt=lmwt, wip=wip, oov_penalty=oov,
                   case_sensitive=case_sensitive)


def test_model_with_an_unlisted_context_matches_oracle(tmp_path):
    """An ARPA file that lists the trigram 'a b a' but not its context 'a b' (SRILM-pruned models do; KenLM inserts blank
    entries): the kernel's signature tables gate an n-gram behind a hit of its context, so the loader must hand such a
    model to the id-keyed walk -- the decoded sentences and the host scorer agree with the oracle's back-off scorer."""
    src = open(ARPA).read()
    pruned = src.replace("-0.6\ta b\t-0.2\n", "").replace("ngram 2=5", "ngram 2=4")
    assert pruned != src
    path = str(tmp_path / "pruned.arpa")
    open(path, "w").write(pruned)
    labels = ["_", "a", "b", " "]
    lm = LanguageModel(path, labels, True)
    olm = O.OracleLM(path)
    a, b = lm.word_index("a"), lm.word_index("b")
    assert abs(lm.score([b, a], a) - (-0.25)) < 1e-6 and abs(olm.base_score([olm.word_index("b"), olm.word_index("a")], olm.word_index("a"))[0] + 0.25) < 1e-6
    for seed, W in ((5, 10), (6, 40), (7, 100)):
        lp = rand_lp(seed, 4, 30, 4, sharp=1.5)
        same_as_oracle(lp, [30, 30, 21, 12], 0, W, labels, lm, olm, lmwt=1.5, wip=0.5, oov_penalty=-5.0, case_sensitive=True)


def test_lm_changes_the_answer():
    # acoustically "b a" and "a b" are close; the LM (which has "<s> a b" but no "<s> b") decides
    labels = ["_", "a", "b", " "]
    p = np.full((1, 5, 4), 0.02)
    p[0, 0, [1, 2]] = [0.46, 0.50]
    p[0, 1, 3] = 0.94
    p[0, 2, [1, 2]] = [0.50, 0.46]
    p[0, 3, 0] = 0.94
    p[0, 4, 0] = 0.94
    p /= p.sum(-1, keepdims=True)
    lp = torch.log(torch.tensor(p))
    no_lm, n0 = U.c_abi_beam(lp, None, 0, 20, labels, wip=0.0)
    lm = LanguageModel(ARPA, labels, True)
    with_lm, n1 = U.c_abi_beam(lp, None, 0, 20, labels, lm, lmwt=3.0, wip=0.0, oov_penalty=-10.0)
    s0 = "".join(labels[i] for i in no_lm[0, : n0[0]])
    s1 = "".join(labels[i] for i in with_lm[0, : n1[0]])
    assert s0 == "b a" and s1 == "a b"


def test_too_wide_a_beam_is_reported():
    from end2end_amd._lib import E2EError
    with pytest.raises(E2EError, match="at most 512"):
        U.c_abi_beam(rand_lp(1, 1, 4, 100), None, 0, 600, None)


@pytest.mark.parametrize("V,W,T", [(100, 100, 40), (200, 100, 60), (1000, 50, 30), (8000, 20, 12), (300, 256, 25),
                                   (40, 300, 30), (700, 512, 20)])
def test_wide_alphabets_take_the_general_kernel_and_match_the_oracle(V, W, T):
    """Alphabets the one-workgroup-LDS kernel cannot hold at this width (the reference has no limit,
    ctc_decoder.cpp:353-441): the general kernel -- keys, child tables and LM answers in the workspace -- must give the
    oracle's result.  Includes the reference's default beam_width = 100 at V >= 82, and beams of 300 / 512 hypotheses, whose
    member sets live in the workspace (round 3; 256 was the limit before)."""
    labels = ["_"] + ["w%d" % i for i in range(V - 2)] + [" "]
    lp = rand_lp(500 + V, 3, T, V, sharp=3.0)
    same_as_oracle(lp, [T, T - 3, max(T // 2, 1)], 0, W, labels, wip=0.5)
    same_as_oracle(lp.float(), None, V - 1, W, None, wip=0.0)          # blank last, no labels, f32


def test_general_kernel_with_a_language_model_matches_the_oracle(tmp_path):
    # 120 single-letter-pair labels: beyond the fast kernel's range with an LM at width 60; words are spelled from them
    letters = "abcde"
    labels = ["_", " "] + [a for a in letters] + [a + b2 for a in letters for b2 in letters] + ["x%d" % i for i in range(88)]
    V = len(labels)
    path = str(tmp_path / "lm.arpa")
    _write_arpa(path, letters, 120, 3, seed=9)
    lm = LanguageModel(path, labels, True)
    olm = O.OracleLM(path)
    g = torch.Generator().manual_seed(91)
    x = torch.randn(3, 40, V, generator=g, dtype=torch.float64) * 2.0
    x[:, :, 1:32] += 3.0
    lp = torch.log_softmax(x, -1)
    same_as_oracle(lp, [40, 31, 17], 0, 60, labels, lm, olm, lmwt=1.2, wip=0.3, oov_penalty=-3.0, case_sensitive=True)


def test_full_c4_shape_properties():
    # BASELINE configs[3] without the LM fixture: B=64, T=1500, V=29, beam 100 -- the oracle needs minutes here, so
    # size-independent properties: the beam result is at least as probable as the greedy one, contains no blanks or
    # immediate artefacts, and is reproducible
    labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
    g = torch.Generator().manual_seed(2)
    lp = torch.log_softmax(torch.randn(64, 1500, 29, generator=g) * 3, -1)
    xl = torch.randint(750, 1501, (64,), generator=g)
    ids, lens = U.c_abi_beam(lp, xl, 0, 100, labels, wip=0.0)
    ids2, lens2 = U.c_abi_beam(lp, xl, 0, 100, labels, wip=0.0)
    assert np.array_equal(ids, ids2) and np.array_equal(lens, lens2)
    gi, gl = U.c_abi_greedy(lp, xl, 0)
    for b in (0, 31, 63):
        n = int(xl[b])
        seq = ids[b, : lens[b]]
        assert (seq != 0).all() and lens[b] <= n
        nll_beam, _ = O.ctc_loss(lp[b:b + 1, :n].double().numpy(), seq[None, :], [n], [len(seq)], 0)
        gseq = gi[b, : gl[b]]
        nll_greedy, _ = O.ctc_loss(lp[b:b + 1, :n].double().numpy(), gseq[None, :], [n], [len(gseq)], 0)
        assert nll_beam[0] <= nll_greedy[0] + 1e-6


def _write_arpa(path, letters, n_words, order, seed):
    """A seeded synthetic ARPA of the given order over short words of `letters` (the reference's LibriSpeech model is not
    available offline): enough entries that table probes collide, enough gaps that the back-off chain is walked."""
    import random
    rng = random.Random(seed)
    words = set()
    while len(words) < n_words:
        words.add("".join(rng.choice(letters) for _ in range(rng.randint(1, 4))))
    words = sorted(words)
    grams = {1: [(w,) for w in words]}
    for n in range(2, order + 1):
        prev = grams[n - 1] if n > 2 else [(w,) for w in words] + [("<s>",)]
        grams[n] = sorted({rng.choice(prev) + (rng.choice(words),) for _ in range(2 * n_words)})
    with open(path, "w") as f:
        f.write("\\data\\\n")
        for n in range(1, order + 1):
            f.write("ngram %d=%d\n" % (n, len(grams[n]) + (3 if n == 1 else 0)))
        f.write("\n\\1-grams:\n-2.0\t<unk>\n-99\t<s>\t-0.3\n-1.5\t</s>\n")
        for n in range(1, order + 1):
            if n > 1:
                f.write("\n\\%d-grams:\n" % n)
            for g in grams[n]:
                bo = "" if n == order else "\t%.4f" % -rng.uniform(0.05, 0.6)
                f.write("%.4f\t%s%s\n" % (-rng.uniform(0.2, 4.0), " ".join(g), bo))
        f.write("\n\\end\\\n")


@pytest.mark.parametrize("order,W", [(3, 40), (4, 25), (2, 100)])
def test_speech_alphabet_lm_beam_matches_oracle(tmp_path, order, W):
    # 29 labels, a few hundred words: most spelled prefixes are out of vocabulary, n-grams are mostly missing (the
    # back-off chain is walked to the unigram), the tables hold enough entries for probe sequences to collide, and the
    # per-member answers are carried over many steps.  Order 4 takes the general device walk for contexts of 3 words.
    labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
    path = str(tmp_path / ("synthetic_%dgram.arpa" % order))
    _write_arpa(path, "abcde", 150, order, seed=order)
    lm = LanguageModel(path, labels, True)
    olm = O.OracleLM(path)
    g = torch.Generator().manual_seed(40 + order)
    x = torch.randn(3, 60, 29, generator=g, dtype=torch.float64) * 2.0
    x[:, :, [1, 2, 3, 4, 5, 27]] += 3.0                    # the model's letters and the space are likely
    lp = torch.log_softmax(x, -1)
    same_as_oracle(lp, [60, 47, 23], 0, W, labels, lm, olm, lmwt=1.3, wip=0.7, oov_penalty=-4.0, case_sensitive=True)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_sweep_matches_oracle(seed):
    """A fixed-seed slice of tools/diag/fuzz_beam.py: random shapes, beam widths, blank / space positions, penalties,
    tie-heavy and -inf-holed emissions, ragged lengths, f32 / f64 input, with and without the tiny 3-gram LM."""
    rng = np.random.default_rng(seed)
    for case in range(25):
        with_lm = bool(rng.integers(0, 3) == 0)
        B = int(rng.integers(1, 4)); T = int(rng.integers(1, 50))
        if with_lm:
            labels = ["_", "a", "b", " "]; V = 4; blank = 0
            cs = bool(rng.integers(0, 2))
            lm = LanguageModel(ARPA, labels, cs); olm = O.OracleLM(ARPA)
            kw = dict(lmwt=float(rng.choice([0.5, 1.0, 2.0])), wip=float(rng.choice([0.0, 1.0])),
                      oov_penalty=float(rng.choice([-1000.0, -3.0])), case_sensitive=cs)
        else:
            V = int(rng.integers(2, 14)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
            labels = list("abcdefghijklm")[:V]
            labels[blank] = "_"
            if V > 2 and rng.integers(0, 2):
                labels[int(rng.choice([i for i in range(V) if i != blank]))] = " "
            lm = olm = None
            kw = dict(wip=float(rng.choice([0.0, 1.0, 2.5])))
        W = int(rng.choice([1, 2, 3, 5, 16, 40, 64, 65, 100, 128, 200]))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * float(rng.choice([0.3, 1.0, 3.0]))
        style = int(rng.integers(0, 3))
        if style == 1:
            x = x.round()                                   # many exact ties
        lp = torch.log_softmax(x, -1)
        if style == 2 and V > 2:
            lp[:, ::3, int(rng.integers(0, V))] = float("-inf")
        xl = rng.integers(1, T + 1, size=B).tolist(); xl[0] = T
        if rng.integers(0, 2):
            lp = lp.float()
        same_as_oracle(lp, xl, blank, W, labels, lm, olm, **kw)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("case", G.beam_q7_cases(), ids=lambda c: c["name"])
def test_pruned_but_living_child_is_found_not_ranked(case, dtype):
    """Quirk Q7 decides these answers (src/decoders/ctc_decoder.cpp:247-252, 397-415; vectors and derivation:
    tests/golden/make_beam_q7_golden.py): the kernel's guards must keep a dropped prefix whose descendant survives findable
    and unranked, exactly as the reference's weak pointers do -- not re-create it."""
    lp = torch.tensor(case["log_probs"], dtype=torch.float64)[None].to(dtype)
    ids, lens = U.c_abi_beam(lp, None, case["blank"], case["beam_width"], case["labels"], None)
    got = ids[0, : lens[0]].tolist()
    assert got == case["expected"] and got != case["without_q7"]


def _fuzz_case(rng):
    """One case of tools/diag/fuzz_beam.py's generator (same distribution, same order of draws)."""
    with_lm = bool(rng.integers(0, 4) == 0)
    B = int(rng.integers(1, 5)); T = int(rng.integers(1, 70))
    if with_lm:
        labels = ["_", "a", "b", " "]; V = 4; blank = 0
        cs = bool(rng.integers(0, 2))
        lm = LanguageModel(ARPA, labels, cs); olm = O.OracleLM(ARPA)
        kw = dict(lmwt=float(rng.choice([0.5, 1.0, 2.0])), wip=float(rng.choice([0.0, 1.0])),
                  oov_penalty=float(rng.choice([-1000.0, -3.0])), case_sensitive=cs)
    else:
        V = int(rng.integers(2, 14)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
        labels = list("abcdefghijklm")[:V]
        labels[blank] = "_"
        if V > 2 and rng.integers(0, 2):
            labels[int(rng.choice([i for i in range(V) if i != blank]))] = " "
        lm = olm = None
        kw = dict(wip=float(rng.choice([0.0, 1.0, 2.5])))
    W = int(rng.choice([1, 2, 3, 5, 16, 40, 64, 65, 100, 128, 200]))
    sharp = float(rng.choice([0.3, 1.0, 3.0]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * sharp
    style = int(rng.integers(0, 4))
    if style == 1:
        x = x.round()                                   # many exact ties
    if style == 2:
        x = torch.zeros_like(x)                         # everything ties
    lp = torch.log_softmax(x, -1)
    if style == 3 and V > 2:
        lp[:, ::3, int(rng.integers(0, V))] = float("-inf")
    xl = rng.integers(1, T + 1, size=B).tolist(); xl[0] = T
    f32 = bool(rng.integers(0, 2))
    return dict(lp=lp.float() if f32 else lp, xl=xl, blank=blank, W=W, labels=labels, lm=lm, olm=olm, kw=kw, style=style)


def test_fuzz_slice_equals_the_oracle_or_is_a_proven_tie():
    """55 cases for each of four seeds of the fuzz tool's generator (220 in all; the whole file runs a second time through the general
    kernel), with and without the language model, tie-heavy emissions included.  Index work is held to bit-exact: every case
    must decode to the oracle's label sequences.  The one tolerated exception is a MATHEMATICAL tie between prefixes -- equal
    scores that the device's and the host's exp / log round differently in the last bit --, and it has to be proven: emissions
    perturbed by ~1e-11 (eight draws) break exact ties and nothing else, so under them device and oracle must agree; a logic
    error survives the perturbation.  Proven ties are counted and must stay under 1 % of the cases."""
    n_cases, ties = 55, []
    for seed, case in ((sd, cs) for sd in (101, 102, 103, 104) for cs in range(n_cases)):
        if case == 0:
            rng = np.random.default_rng(seed)
        c = _fuzz_case(rng)
        gpu_kw = {k: v for k, v in c["kw"].items() if k != "case_sensitive"}
        ref = c["lp"].double().numpy()
        ids, lens = U.c_abi_beam(c["lp"], c["xl"], c["blank"], c["W"], c["labels"], c["lm"], **gpu_kw)
        o_ids, o_lens, _ = O.ctc_beam(ref, c["xl"], c["blank"], c["W"], c["labels"], c["olm"], **c["kw"])
        if lens.tolist() == o_lens.tolist() and ids.tolist() == o_ids.tolist():
            continue
        prng = np.random.default_rng(1000 * seed + case)
        for _ in range(8):
            qn = ref + prng.normal(size=ref.shape) * 1e-11
            qn[~np.isfinite(ref)] = -np.inf
            i2, l2 = U.c_abi_beam(torch.from_numpy(qn), c["xl"], c["blank"], c["W"], c["labels"], c["lm"], **gpu_kw)
            o2, ol2, _ = O.ctc_beam(qn, c["xl"], c["blank"], c["W"], c["labels"], c["olm"], **c["kw"])
            assert l2.tolist() == ol2.tolist() and i2.tolist() == o2.tolist(), \
                "seed %d case %d (style %d, W %d): differs from the oracle and is not a tie" % (seed, case, c["style"], c["W"])
        ties.append((seed, case, c["style"]))
    assert len(ties) <= 0.01 * 4 * n_cases, "proven ties (seed, case, style): %s" % ties


@pytest.mark.parametrize("case", G.beam_bruteforce_cases(), ids=lambda c: c["name"])
def test_against_exhaustive_enumeration(case):
    """Independent pin (tests/golden/make_beam_golden.py: every alignment enumerated in pure Python): a beam that never
    prunes returns argmax over all labellings of log P - wip * num_words.  f64 and f32 inputs.  Cases whose prefix count
    exceeds what one workgroup holds run at beam 200 and must then agree with the oracle at that width (which the CPU
    suite holds to the enumeration at full width)."""
    lp = torch.tensor(case["log_probs"], dtype=torch.float64)[None]
    full = case["beam_width"] <= 255
    W = case["beam_width"] if full else 200
    want = case["want_ids"]
    if not full:
        o_ids, o_lens, _ = O.ctc_beam(lp.numpy(), None, case["blank"], W, case["labels"], wip=case["wip"])
        want = o_ids[0, : o_lens[0]].tolist()
    for x in (lp, lp.float()):
        ids, lens = U.c_abi_beam(x, None, case["blank"], W, case["labels"], wip=case["wip"])
        assert ids[0, : lens[0]].tolist() == want


def _c4_inputs():
    """BASELINE configs[3] exactly as bench.py builds it: B=64, T=1500, V=29, beam 100, seeded 10k-word 3-gram."""
    import bench
    labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1024, 1500, 29, generator=g) * 3
    return labels, torch.log_softmax(x[:64], -1), bench.synthetic_arpa


def test_full_c4_with_the_3gram_lm_matches_oracle(tmp_path):
    """BASELINE configs[3] at FULL size with the LM (the bench's inputs and model): every one of the 64 utterances x 1500
    frames must come out identical to the oracle, twice (reproducible)."""
    labels, lp, synthetic_arpa = _c4_inputs()
    path = str(tmp_path / "synthetic_3gram.arpa")
    synthetic_arpa(path, labels)
    lm = LanguageModel(path, labels, True)
    olm = O.OracleLM(path)
    kw = dict(lmwt=1.0, wip=1.0, oov_penalty=-10.0)
    ids, lens = U.c_abi_beam(lp, None, 0, 100, labels, lm, **kw)
    ids2, lens2 = U.c_abi_beam(lp, None, 0, 100, labels, lm, **kw)
    assert np.array_equal(ids, ids2) and np.array_equal(lens, lens2)
    o_ids, o_lens, _ = O.ctc_beam(lp.double().numpy(), None, 0, 100, labels, olm, case_sensitive=True,
                                  n_threads=min(64, os.cpu_count() or 8), **kw)
    assert lens.tolist() == o_lens.tolist()
    assert np.array_equal(ids, o_ids)
    # the LM took part: decoding the same emissions without it gives other sentences
    ids0, lens0 = U.c_abi_beam(lp[:4], None, 0, 100, labels, wip=1.0)
    assert any(ids0[b, : lens0[b]].tolist() != ids[b, : lens[b]].tolist() for b in range(4))


def test_c4_shape_in_vocabulary_speech_with_the_3gram_lm_matches_oracle(tmp_path):
    """Same shape and model, emissions that favour sentences of the model's own words (listed bigrams and trigrams are
    hit, not only <unk> / back-off to the unigram), ragged lengths, a second LM weight."""
    import random
    labels, _, synthetic_arpa = _c4_inputs()
    path = str(tmp_path / "synthetic_3gram.arpa")
    synthetic_arpa(path, labels)
    tri = [ln.split("\t")[1].split() for ln in open(path).read().split("\\3-grams:")[1].splitlines() if "\t" in ln]
    tri = [t3 for t3 in tri if all(w.isalpha() for w in t3)]          # (not the ones that start a sentence: "<s>")
    rng = random.Random(5)
    B, T = 16, 1500
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, 29, generator=g) * 1.5
    xl = []
    for b in range(B):
        text, t = [], 0
        while t < T - 40:
            for w in rng.choice(tri):
                for ch in w + " ":
                    c = labels.index(ch)
                    d = rng.randint(1, 3)
                    x[b, t:t + d, c] += 6.0
                    t += d
                    x[b, t:t + 1, 0] += 6.0                  # a blank between characters (also separates doubles)
                    t += 1
                    if t >= T - 40:
                        break
                if t >= T - 40:
                    break
        xl.append(rng.randint(T // 2, T))
    xl[0] = T
    lp = torch.log_softmax(x, -1)
    lm = LanguageModel(path, labels, True)
    olm = O.OracleLM(path)
    for lmwt in (1.0, 2.5):
        kw = dict(lmwt=lmwt, wip=0.5, oov_penalty=-10.0)
        ids, lens = U.c_abi_beam(lp, xl, 0, 100, labels, lm, **kw)
        o_ids, o_lens, sents = O.ctc_beam(lp.double().numpy(), xl, 0, 100, labels, olm, case_sensitive=True,
                                          n_threads=min(B, os.cpu_count() or 8), **kw)
        assert lens.tolist() == o_lens.tolist() and np.array_equal(ids, o_ids)
    words = set(w for t3 in tri for w in t3)
    hit = sum(w in words for w in sents[0].split())
    assert hit >= 0.8 * max(1, len(sents[0].split()))         # the decoded text is made of the model's words


@pytest.mark.parametrize("name", ["lm_order4", "lm_order3_nounk"])
def test_device_lm_tables_with_the_definition_pinned_models(name):
    """The golden ARPA models whose host-side scores are pinned to the ARPA definition (tests/test_lm_cpu.py) drive
    the DEVICE tables here: order 4 takes the general walk, order 3 the round-probed one; a model without <unk>."""
    labels = ["_", "a", "b", "c", "d", "e", "'", " "]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".arpa")
    lm = LanguageModel(path, labels, True)
    olm = O.OracleLM(path)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(4, 90, 8, generator=g, dtype=torch.float64) * 2.0
    x[:, :, 7] += 1.0
    lp = torch.log_softmax(x, -1)
    for W, kw in ((30, dict(lmwt=1.0, wip=0.0, oov_penalty=-5.0)), (100, dict(lmwt=2.0, wip=1.0, oov_penalty=-1000.0))):
        same_as_oracle(lp, [90, 71, 33, 5], 0, W, labels, lm, olm, case_sensitive=True, **kw)


def test_language_model_follows_the_logits_device():
    """the LM's tables live on one GPU; a decoder built while cuda:0 is current must still decode tensors that
    live on cuda:1 (a per-device copy is loaded on first use), and the raw C ABI refuses a model of another device."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from end2end_amd import CTCDecoder
    from end2end_amd._runtime import E2EError
    labels = ["_", "a", "b", " "]
    dec = CTCDecoder(beam_width=8, labels=labels, lm_path=ARPA, after_logsoftmax=True)
    lp = rand_lp(3, 2, 20, 4).float()
    r0 = dec.decode(lp.to("cuda:0"))
    r1 = dec.decode(lp.to("cuda:1"))
    assert r0.decoded_sentences == r1.decoded_sentences
    lm = LanguageModel(ARPA, labels, True)
    with torch.cuda.device(1):
        with pytest.raises(E2EError, match="device"):
            from end2end_amd import _C, _runtime as R
            x = lp.to("cuda:1")
            out = torch.empty((2, 21), dtype=torch.long, device="cuda:1")
            ol = torch.empty(2, dtype=torch.long, device="cuda:1")
            xl = torch.full((2,), 20, dtype=torch.long, device="cuda:1")
            ws = torch.empty(_C.ctc_beam_workspace_bytes(2, 20, 4, 8), dtype=torch.uint8, device="cuda:1")
            _C.ctc_beam(x.data_ptr(), R.F32, *x.stride(), xl.data_ptr(), 2, 20, 4, 0, 8, 3, lm.on(torch.device("cuda", 0)).handle,
                        1.0, 0.0, -10.0, out.data_ptr(), 21, ol.data_ptr(), ws.data_ptr(), ws.numel(), 0)


def test_whole_suite_through_the_general_kernel():
    """Every test of this file again with E2E_BEAM_GENERAL=1 (read once per process: a child pytest), so that the general
    kernel is held to the same known answers, brute-force goldens, tie cases, LM cases and sweeps as the fast one."""
    import subprocess
    import sys
    if os.environ.get("E2E_BEAM_GENERAL"):
        pytest.skip("already inside the child run")
    env = dict(os.environ, E2E_BEAM_GENERAL="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x",
                          "-k", "not full_c4 and not c4_shape and not whole_suite"],
                         env=env, capture_output=True, text=True, timeout=1500,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("V,W", [(29, 100), (29, 20), (600, 40)], ids=["V29_W100", "V29_W20", "V600_W40_general"])
def test_sixteen_bit_log_probabilities_are_searched_as_their_f32_images(dtype, V, W):
    """The beam search reads f16 / bf16 log-probabilities as they are (no up-cast pass; upstream converts whatever arrives once,
    ctc_decoder.cpp:157-160): every such value is an f32 number, so the result must be the f32 call's on the same values,
    bit for bit."""
    g = torch.Generator().manual_seed(V + W)
    lp16 = torch.log_softmax(torch.randn(3, 70, V, generator=g) * 3, -1).to(dtype)
    xl = [70, 51, 64]
    labels = (["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]) if V == 29 else None
    ids16, len16 = U.c_abi_beam(lp16, xl, 0, W, labels, wip=1.0 if labels else 0.0)
    ids32, len32 = U.c_abi_beam(lp16.float(), xl, 0, W, labels, wip=1.0 if labels else 0.0)
    assert len16.tolist() == len32.tolist() and ids16.tolist() == ids32.tolist()
    assert (len16 > 0).all()

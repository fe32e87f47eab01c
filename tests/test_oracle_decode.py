"""CPU: the oracle's greedy / beam search against the reference tests' known answers."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


def _x(case):
    x = np.array(case["x"], dtype=np.float32)
    if case["input_kind"] == "log_of_probs":
        x = np.log(x)
    return x.astype(np.float64)


@pytest.mark.parametrize("case", G.known_answers()["decode"], ids=lambda c: c["name"])
def test_greedy_known_answers(case):
    x = _x(case)
    out, lens = O.ctc_greedy(x, case.get("x_len"), case["blank"])
    sent = ["".join(case["labels"][i] for i in out[b, : lens[b]]) for b in range(len(lens))]
    assert sent == case["greedy"]
    if "greedy_targets" in case:
        assert out.tolist() == case["greedy_targets"]   # width Tmax, zero padded (Q5)
        assert lens.tolist() == case["greedy_lengths"]


@pytest.mark.parametrize("case", [c for c in G.known_answers()["decode"] if "beam" in c], ids=lambda c: c["name"])
def test_beam_known_answers(case):
    lp = _x(case)
    _, _, sent = O.ctc_beam(lp, None, case["blank"], case["beam_width"], case["labels"], wip=case["wip"])
    assert sent == case["beam"]


def test_greedy_ties_first_max_and_padding():
    x = np.zeros((1, 4, 3))
    x[0, 1, 1] = x[0, 1, 2] = 5.0          # tie -> index 1
    x[0, 2, 2] = 1.0
    out, lens = O.ctc_greedy(x, [3], blank=0)
    assert out.tolist() == [[1, 2, 0, 0]] and lens.tolist() == [2]


def test_empty_prefix_winner_is_minus_one():
    # Q6: all-blank input, no labels -> [[-1]], length 1 (ctc_decoder.cpp:232-245)
    lp = np.log(np.array([[[0.98, 0.01, 0.01]] * 4]))
    out, lens, sent = O.ctc_beam(lp, None, 0, 10, None)
    assert out.tolist() == [[-1]] and lens.tolist() == [1] and sent == [""]


def test_beam_sums_paths_and_width_one_beam_is_not_greedy():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 12, 6)) * 2
    lp = x - np.log(np.exp(x).sum(-1, keepdims=True))
    labels = ["_", "a", "b", "c", " ", "d"]
    o1 = O.ctc_beam(lp, [12, 9, 5], 0, 50, labels, wip=0.0)
    o2 = O.ctc_beam(lp, [12, 9, 5], 0, 50, labels, wip=0.0, n_threads=1)
    assert o1[2] == o2[2] and np.array_equal(o1[0], o2[0])
    # a wide beam can only find an equal or more probable labelling than a narrow one
    from math import isfinite

    def nll(ids, b, n):
        if len(ids) == 1 and ids[0] == -1:
            ids = []
        l, _ = O.ctc_loss(lp[b:b + 1, :n], np.array([list(ids) + [0]]), [n], [len(ids)], 0)
        return l[0]
    narrow = O.ctc_beam(lp, [12, 9, 5], 0, 2, labels, wip=0.0)
    for b, n in enumerate([12, 9, 5]):
        wide_nll = nll(o1[0][b, : o1[1][b]], b, n)
        narrow_nll = nll(narrow[0][b, : narrow[1][b]], b, n)
        assert isfinite(wide_nll) and wide_nll <= narrow_nll + 1e-9


def test_word_insertion_penalty_prefers_fewer_words():
    # two frames strongly "a", then weak choice between " " + "a" and staying
    labels = ["_", "a", " "]
    p = np.array([[[0.05, 0.9, 0.05], [0.3, 0.1, 0.6], [0.05, 0.9, 0.05]]])
    lp = np.log(p)
    s0 = O.ctc_beam(lp, None, 0, 20, labels, wip=0.0)[2][0]
    s5 = O.ctc_beam(lp, None, 0, 20, labels, wip=5.0)[2][0]
    # a lone " " carries no word, so with a heavy penalty it outranks "a" (num_words counts word STARTS,
    # ctc_decoder.cpp:258-262,314-318)
    assert s0 == "a a" and len(s5.split()) < 2


@pytest.mark.parametrize("case", G.beam_bruteforce_cases(), ids=lambda c: c["name"])
def test_beam_against_exhaustive_enumeration(case):
    """Independent pin: with a beam wide enough never to prune, the search must return the labelling that maximises
    log P - wip * num_words over ALL labellings (enumerated alignment by alignment in tests/golden/make_beam_golden.py)."""
    lp = np.array(case["log_probs"], dtype=np.float64)[None]
    for W in (case["beam_width"], case["beam_width"] + 37):
        ids, lens, _ = O.ctc_beam(lp, None, case["blank"], W, case["labels"], wip=case["wip"])
        assert ids[0, : lens[0]].tolist() == case["want_ids"]


@pytest.mark.parametrize("case", G.beam_q7_cases(), ids=lambda c: c["name"])
def test_pruned_but_living_child_is_found_not_ranked(case):
    """Quirk Q7 decides these answers (src/decoders/ctc_decoder.cpp:247-252, 397-415): a prefix dropped from the beam whose
    descendant survives is still found by its parent's weak pointer and therefore never ranked again.  Expected sentences:
    tests/golden/make_beam_q7_golden.py (paper derivation + an independent Python model of the ownership rules); the
    answer of a search WITHOUT the quirk is stored beside them and must not be what the oracle returns."""
    lp = np.array(case["log_probs"])[None]
    ids, lens, _ = O.ctc_beam(lp, [lp.shape[1]], case["blank"], case["beam_width"], case["labels"], None)
    got = ids[0, :lens[0]].tolist()
    assert got == case["expected"] and got != case["without_q7"]


@pytest.mark.parametrize("case", G.align_cases(), ids=lambda c: c["name"])
def test_forced_alignment_against_the_reference_functions(case):
    """oracle_ctc_align vs the outputs of the reference's own _get_alignment_ctc_1d / _asg_1d / get_alignment_3d
    (pytorch_end2end/utils/alignment.py, run by tests/golden/make_align_golden.py)."""
    out = O.ctc_align(case["lp"], case["targets"], case["x_len"], case["t_len"], 0, bool(case["is_ctc"]))
    assert out.tolist() == case["out"].tolist()

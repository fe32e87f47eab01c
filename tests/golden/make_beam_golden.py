#!/usr/bin/env python3
"""Generator of the brute-force decode goldens (tests/golden/beam_bruteforce.json).

INDEPENDENT PIN for prefix beam search without a language model.  Pure Python + numpy, no code shared with the
product or the oracle: for a short utterance EVERY alignment (V^T of them) is enumerated, collapsed
(repeats merged, blanks dropped), and the probabilities of the alignments of each sentence are summed -- the exact CTC
probability of every labelling, the idea of the reference's own `_get_best_result_brute_force`
(tests/test_ctc_decoder.py:16-41, which loops over sentences and asks CTCLoss).  The answer a prefix beam search must
give when its beam is wide enough never to prune (beam_width >= number of distinct prefixes, sum_k (V-1)^k) is

    argmax over sentences of   log P(sentence | x)  -  wip * num_words(sentence)

(src/decoders/ctc_decoder.cpp:314-318 with no LM: lmwt = 0, no OOV term; :418-424 final sort), where num_words counts
the words the reference starts (:258-262: a non-space character after a space or at the beginning).  Cases whose best
and second-best scores are closer than 1e-6 are re-drawn so that the winner does not hang on rounding.

    python tests/golden/make_beam_golden.py     # rewrites beam_bruteforce.json (deterministic)
"""
import itertools
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def num_words(sentence, space_id):
    n, prev_space = 0, True
    for c in sentence:
        if c != space_id and prev_space:
            n += 1
        prev_space = c == space_id
    return n


def brute_force(lp, blank, space_id, wip):
    T, V = lp.shape
    prob = {}
    for ali in itertools.product(range(V), repeat=T):
        s, prev = [], None
        for c in ali:
            if c != prev and c != blank:
                s.append(c)
            prev = c
        p = math.exp(sum(lp[t, c] for t, c in enumerate(ali)))
        key = tuple(s)
        prob[key] = prob.get(key, 0.0) + p
    assert abs(sum(prob.values()) - 1.0) < 1e-9
    scored = sorted(((math.log(p) - wip * num_words(s, space_id), s) for s, p in prob.items()), reverse=True)
    return scored[0][1], scored[0][0], scored[0][0] - scored[1][0], len(prob)


def main():
    cases = []
    specs = [  # (name, T, V, blank, space_id, wip, sharpness); the first group needs a beam of <= 255 prefixes
        ("t4_v4", 4, 4, 0, -1, 0.0, 1.0), ("t4_v4_blank3_space_wip1", 4, 4, 3, 1, 1.0, 0.5),
        ("t4_v4_flat_space_wip05", 4, 4, 0, 2, 0.5, 0.15), ("t5_v3", 5, 3, 0, -1, 0.0, 0.6),
        ("t6_v3_space_wip1", 6, 3, 0, 2, 1.0, 0.5), ("t6_v3_blank2", 6, 3, 2, -1, 0.0, 0.3),
        ("t7_v3", 7, 3, 0, -1, 0.0, 0.9), ("t7_v3_blank1_space_wip1", 7, 3, 1, 2, 1.0, 0.4),
        ("t7_v3_flat", 7, 3, 0, -1, 0.0, 0.1), ("t3_v6", 3, 6, 5, -1, 0.0, 1.0), ("t3_v6_space_wip2", 3, 6, 0, 3, 2.0, 0.3),
        ("t2_v12", 2, 12, 0, -1, 0.0, 0.5),
        # wider than one workgroup's beam today: pinned on the CPU oracle, the GPU runs them at a narrower beam
        ("t5_v4", 5, 4, 0, -1, 0.0, 1.5), ("t6_v4", 6, 4, 0, -1, 0.0, 0.7),
        ("t6_v4_blank2", 6, 4, 2, -1, 0.0, 1.0), ("t6_v4_space_wip1", 6, 4, 0, 3, 1.0, 0.8),
        ("t6_v4_space_wip05", 6, 4, 0, 2, 0.5, 0.5), ("t4_v5_space_wip2", 4, 5, 0, 4, 2.0, 0.6),
        ("t6_v4_flat", 6, 4, 0, -1, 0.0, 0.15), ("t6_v4_space_wip1_flat", 6, 4, 0, 1, 1.0, 0.2),
    ]
    for name, T, V, blank, space_id, wip, sharp in specs:
        seed = 0
        while True:
            rng = np.random.default_rng(sum(map(ord, name)) * 31 + seed)
            x = rng.normal(size=(T, V)) * sharp * 3.0
            lp = np.round(x - np.log(np.exp(x).sum(-1, keepdims=True)), 12)     # the stored values are the inputs
            best, score, margin, n_sent = brute_force(lp, blank, space_id, wip)
            if margin > 1e-6:
                break
            seed += 1
        width = sum((V - 1) ** k for k in range(T + 1))         # every prefix fits: the search never prunes
        cases.append({"name": name, "T": T, "V": V, "blank": blank, "space_id": space_id, "wip": wip,
                      "log_probs": [[float(v) for v in row] for row in lp], "beam_width": width,
                      "best": list(best), "best_score": score, "margin": margin, "n_sentences": n_sent})
        print(name, "best", best, "score %.6f margin %.2e sentences %d width %d" % (score, margin, n_sent, width))
    with open(os.path.join(HERE, "beam_bruteforce.json"), "w") as f:
        json.dump({"doc": "see make_beam_golden.py", "cases": cases}, f, separators=(",", ":"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Hand-built prefix beam search vectors whose answer is DECIDED by quirk Q7 of the reference's decoder
(tests/golden/beam_q7.json).

Q7 (src/decoders/ctc_decoder.cpp:247-252, 397-415).  A prefix owns its parent (`std::shared_ptr<Prefix> parent`,
ctc_decoder.h:86) and knows its children only weakly (`std::map<int, std::weak_ptr<Prefix>> next_data`, :87).  When
`nth_element` + `resize` (:405-415) drop a prefix C from the beam, C is destroyed -- unless one of its descendants is still
in the beam and keeps it alive through the `parent` chain.  If C's parent P is still in the beam as well, the next
`get_next_prefix(P, c)` (:250-252) finds the weak pointer still lockable and returns C with `is_new == false`: C is given
the probability mass of P + c, but it is NOT appended to `new_prefixes` (:381), so it is not in the vector that is ranked --
the extension P + c cannot re-enter the beam for as long as that descendant lives.  A search without the quirk would rank
it again.

Vector "q7_b_ba_bab" (V = 3: blank 0, a 1, b 2; beam_width 2; no language model), derived by hand; probabilities per frame
are (blank, a, b):

  t0  (0.06, 0.04, 0.90)   candidates  "b" .90   "" .06   "a" .04                     -> beam {"b", ""}
  t1  (0.10, 0.60, 0.30)   "ba" = .9*.6 = .54
                           "b"  = .9*.1 (blank) + .9*.3 (repeat) + .06*.3 (from "") = .378
                           "a"  = .06*.6 = .036   "" = .006   "bb" = .3 * prev_blank("b") = 0     -> beam {"ba", "b"}
  t2  (0.45, 1e-9, 0.55)   "b"   = .378*.45 + .288*.55 (repeat: the non-blank part .27 + .018) = .3285
                           "bab" = .54*.55 = .297   "ba" = .54*.45 = .243   "bb" = .09*.55 = .0495
                           -> beam {"b", "bab"}; "ba" is dropped but stays alive: "bab" holds it as its parent, and its
                              own parent "b" is still in the beam
  t3  (0.05, 0.90, 0.05)   "b" + a would be "ba" = .3285*.9 = .29565 -- the best score of the step --, but
                           get_next_prefix("b", a) finds the living "ba" and reports is_new == false: not ranked.
                           "baba" = .297*.9 = .2673   "bab" = .297*.05 + .297*.05 = .0297   "b" = .0243   "bb" = .0085
                           -> the reference's beam is {"baba", "bab"} and it answers "baba" = [2, 1, 2, 1];
                              without the quirk the beam is {"ba", "baba"} and the answer is "ba" = [2, 1].

The file also carries an INDEPENDENT model of those ownership rules -- Python objects with a strong `parent` and
`weakref` children, whose reference counting destroys a dropped prefix exactly when shared_ptr would -- and a plain prefix
beam search without them; both are run on every vector, must differ, and their answers are what the JSON stores.  Nothing is
shared with the product or with oracle/ctc_oracle.c (which restates the quirk with explicit reference counts).

    python tests/golden/make_beam_q7_golden.py     # rewrites beam_q7.json (deterministic)
"""
import json
import math
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NEG = -math.inf


def lse(a, b):
    if a == NEG:
        return b
    if b == NEG:
        return a
    m = max(a, b)
    return m + math.log(math.exp(a - m) + math.exp(b - m))


class Prefix:
    """ctc_decoder.h:69-96 without the language-model members."""
    __slots__ = ("pb", "pnb", "prev_pb", "prev_pnb", "last_char", "parent", "next_data", "__weakref__")

    def __init__(self):
        self.pb = self.pnb = self.prev_pb = self.prev_pnb = NEG
        self.last_char = -1
        self.parent = None          # strong: std::shared_ptr<Prefix>
        self.next_data = {}         # char -> weakref.ref: std::weak_ptr<Prefix>

    def prev_full(self):
        return lse(self.prev_pnb, self.prev_pb)

    def sentence(self):             # get_sentence, ctc_decoder.cpp:232-245
        out, q = [], self
        while q is not None and (q is self or q.parent is not None):
            out.append(q.last_char)
            q = q.parent
        return out[::-1]


def decode_reference_ownership(lp, blank, width):
    """decode_sentence (ctc_decoder.cpp:353-441) with the reference's ownership: children are weak."""
    root = Prefix()
    root.prev_pb = 0.0
    prefixes = [root]
    del root
    for t in range(lp.shape[0]):
        fresh = []
        for c in range(lp.shape[1]):
            cur = lp[t, c]
            for p in prefixes:
                if c == blank:
                    p.pb = lse(p.pb, cur + p.prev_full())
                    continue
                ref = p.next_data.get(c)
                q = ref() if ref is not None else None          # .lock()
                if q is None:
                    q = Prefix()
                    q.last_char = c
                    q.parent = p
                    p.next_data[c] = weakref.ref(q)
                    fresh.append(q)                             # is_new: appended to new_prefixes
                if c == p.last_char:
                    q.pnb = lse(q.pnb, cur + p.prev_pb)
                    p.pnb = lse(p.pnb, cur + p.prev_pnb)
                else:
                    q.pnb = lse(q.pnb, cur + p.prev_full())
                del q
            del p
        prefixes.extend(fresh)
        del fresh
        for p in prefixes:
            p.prev_pb, p.prev_pnb, p.pb, p.pnb = p.pb, p.pnb, NEG, NEG
        del p
        if len(prefixes) > width:
            scores = [p.prev_full() for p in prefixes]
            assert len(set(round(s, 9) for s in scores if s > NEG)) == sum(s > NEG for s in scores), "tie in a hand-built vector"
            order = sorted(range(len(prefixes)), key=lambda i: -scores[i])
            prefixes = [prefixes[i] for i in order[:width]]     # everything else loses its owner here
    best = max(prefixes, key=lambda p: p.prev_full())
    return best.sentence()


def decode_without_the_quirk(lp, blank, width):
    """A prefix beam search that ranks P + c whenever P is in the beam: prefixes are keyed by their label sequence and the
    dropped ones are forgotten entirely."""
    beam = {(): (0.0, NEG)}                                     # sentence -> (prev_pb, prev_pnb)
    for t in range(lp.shape[0]):
        nxt = {}

        def add(s, pb, pnb):
            a = nxt.get(s, (NEG, NEG))
            nxt[s] = (lse(a[0], pb), lse(a[1], pnb))
        for s, (ppb, ppnb) in beam.items():
            full = lse(ppb, ppnb)
            for c in range(lp.shape[1]):
                cur = lp[t, c]
                if c == blank:
                    add(s, cur + full, NEG)
                elif s and c == s[-1]:
                    add(s + (c,), NEG, cur + ppb)
                    add(s, NEG, cur + ppnb)
                else:
                    add(s + (c,), NEG, cur + full)
        ranked = sorted(nxt.items(), key=lambda kv: -lse(*kv[1]))
        beam = dict(ranked[:width])
    return list(max(beam.items(), key=lambda kv: lse(*kv[1]))[0])


VECTORS = [
    # name, blank, beam width, per-frame probabilities in alphabet order
    ("q7_b_ba_bab", 0, 2, [(0.06, 0.04, 0.90), (0.10, 0.60, 0.30), (0.45, 1e-9, 0.55), (0.05, 0.90, 0.05)]),
    # the same situation with the labels permuted (blank in the last column) and one closing frame of blank
    ("q7_blank_last", 2, 2, [(0.04, 0.90, 0.06), (0.60, 0.30, 0.10), (1e-9, 0.55, 0.45), (0.90, 0.05, 0.05), (0.05, 0.05, 0.90)]),
    # width 3, a fourth symbol c: the third place of the beam goes to a bystander ("ca", then "cab") and "ba" (.1269 at t2)
    # is dropped behind "b" .1772, "bab" .1551, "cab" .1485 while "bab" keeps it alive; at t3 "b" + a = .1594 would win
    ("q7_width3_bystander", 0, 3, [(0.05, 0.03, 0.47, 0.45), (0.10, 0.60, 0.30, 1e-9), (0.45, 1e-9, 0.55, 1e-9), (0.05, 0.90, 0.05, 1e-9)]),
]


def main():
    cases = []
    for name, blank, width, probs in VECTORS:
        p = np.array(probs, dtype=np.float64)
        p /= p.sum(-1, keepdims=True)
        lp = np.round(np.log(p), 12)                            # the stored values are the inputs
        ref = decode_reference_ownership(lp, blank, width)
        plain = decode_without_the_quirk(lp, blank, width)
        print(name, "reference ownership ->", ref, "| without the quirk ->", plain)
        assert ref != plain, "the vector does not exercise Q7"
        cases.append({"name": name, "blank": blank, "beam_width": width, "log_probs": [[float(v) for v in row] for row in lp],
                      "expected": ref, "without_q7": plain})
    assert cases[0]["expected"] == [2, 1, 2, 1] and cases[0]["without_q7"] == [2, 1]      # the derivation in the docstring
    with open(os.path.join(HERE, "beam_q7.json"), "w") as f:
        json.dump({"doc": "see make_beam_q7_golden.py", "cases": cases}, f, separators=(",", ":"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generator of the language-model golden vectors (tests/golden/lm_*.arpa, lm_golden.json).

INDEPENDENT PIN.  Nothing here imports the product, the oracle, or any code they share: the expected scores are
derived in pure Python straight from the definition of the ARPA format, as KenLM's BaseScore implements it for the
reference's calls (src/decoders/ctc_decoder.cpp:275-278,291-294):

    p(w | c_1 .. c_k)  =  P[c_1 .. c_k w]                       if that n-gram is listed,
                          B[c_1 .. c_k] + p(w | c_2 .. c_k)     otherwise (B = 0 for a context that is not listed
                                                                 or has no back-off column),
    p(w | <empty>)     =  P[w]; a word outside the vocabulary is <unk>; a model without <unk> scores it -100
                          (KenLM's default unknown_missing_logprob),
    contexts are cut to the model order - 1 most recent words,

plus what KenLM keeps as the state after a word (lm/model.cc ScoreExceptBackoff: the longest matched suffix whose
back-off weight is non-zero, "HasExtension") -- recorded as `kenlm_state_len`; the generator itself asserts that
scoring from that minimised state equals scoring from the full history (so an implementation that keeps the full
history, like the two under test, must give the same numbers).

The models are synthetic (seeded): n-grams counted from random sentences, so every n-gram's prefix and suffix is
listed (what lmplz writes), then a few bigrams that are only suffixes are deleted (what SRILM pruning leaves; KenLM
fills such holes with "blank" entries and still finds the listed trigram -- the definition above says the same).
Special cases on purpose: `-99 <s>`, n-grams without a back-off column, a non-zero back-off on an n-gram nothing
extends, a model without <unk>, words outside the vocabulary inside contexts.

    python tests/golden/make_lm_golden.py      # rewrites the three files next to it (deterministic)
"""
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
WORDS = ["a", "b", "ab", "ba", "abc", "cab", "d", "e", "dead", "bead", "ace", "cad", "dab", "bed", "ebb", "add",
         "be", "ad", "ca", "de", "ec", "bad", "dad", "cede", "deed", "a'd", "e'", "bab", "acc", "baa"]


def build(seed, order, with_unk, n_sent):
    rng = random.Random(seed)
    sents = []
    for _ in range(n_sent):
        n = rng.randint(1, 6)
        sents.append(["<s>"] + [rng.choice(WORDS[: 8 + seed % 5 * 5]) for _ in range(n)] + ["</s>"])
    grams = [set() for _ in range(order + 1)]
    for s in sents:
        for k in range(1, order + 1):
            for i in range(len(s) - k + 1):
                grams[k].add(tuple(s[i:i + k]))
    for w in WORDS:
        grams[1].add((w,))
    grams[1].add(("<s>",)); grams[1].add(("</s>",))
    if with_unk:
        grams[1].add(("<unk>",))
    # holes: delete some bigrams that no longer n-gram starts with (prefix closure stays; only a suffix goes missing)
    prefixes = {g[:-1] for k in range(2, order + 1) for g in grams[k]}
    suffix_only = sorted(g for g in grams[2] if g not in prefixes and any(t[1:] == g for t in grams[3]))
    holes = set(rng.sample(suffix_only, min(6, len(suffix_only))))
    grams[2] -= holes
    P, B = {}, {}
    for k in range(1, order + 1):
        for g in sorted(grams[k]):
            P[g] = round(-rng.uniform(0.1, 3.5), 4)
            if k < order:
                if g in prefixes:
                    B[g] = round(-rng.uniform(0.05, 1.2), 4)
                elif rng.random() < 0.15:
                    B[g] = round(-rng.uniform(0.05, 0.9), 4)      # a back-off weight on an n-gram nothing extends
                # else: no back-off column at all
    P[("<s>",)] = -99.0
    if ("</s>",) in B:
        del B[("</s>",)]
    return P, B, sorted(holes)


def write_arpa(path, P, B, order):
    with open(path, "w") as f:
        f.write("\\data\\\n")
        for k in range(1, order + 1):
            f.write("ngram %d=%d\n" % (k, sum(1 for g in P if len(g) == k)))
        for k in range(1, order + 1):
            f.write("\n\\%d-grams:\n" % k)
            for g in sorted(g for g in P if len(g) == k):
                line = "%g\t%s" % (P[g], " ".join(g))
                if g in B:
                    line += "\t%g" % B[g]
                f.write(line + "\n")
        f.write("\n\\end\\\n")


class Definition:
    """The ARPA definition, recursively, in double precision."""

    def __init__(self, P, B, order):
        self.P, self.B, self.order = P, B, order
        self.vocab = {g[0] for g in P if len(g) == 1}
        self.implied = {g[i:] for g in P for i in range(1, len(g))} - set(P)

    def norm(self, w):
        return w if w in self.vocab else "<unk>"

    def p(self, w, ctx):
        """ctx oldest-first, already normalised; returns (log10 p, sum of |terms|)."""
        g = ctx + (w,)
        if g in self.P:
            return self.P[g], abs(self.P[g])
        if not ctx:
            return -100.0, 100.0                                  # <unk> itself is not listed
        b = self.B.get(ctx, 0.0)
        s, t = self.p(w, ctx[1:])
        return b + s, abs(b) + t

    def score(self, history, w):
        """history: most recent first (any length, any words)."""
        ctx = tuple(reversed([self.norm(x) for x in history[: self.order - 1]]))
        return self.p(self.norm(w), ctx)

    def kenlm_state_len(self, history, w):
        words = [self.norm(w)] + [self.norm(x) for x in history[: self.order - 1]]       # most recent first
        use = 0
        for k in range(1, self.order):
            if k > len(words):
                break
            g = tuple(reversed(words[:k]))
            if g not in self.P:
                if g in self.implied:
                    return None      # a deleted suffix: what KenLM keeps there depends on its "blank" entries
                break
            if self.B.get(g, 0.0) != 0.0:
                use = k
        return use


def main():
    out = {"doc": "see make_lm_golden.py; a query row is [ctx (most recent first, space separated), word, score, "
                  "abs_terms = sum of |summands| (for the float32 tolerance of an implementation that adds float "
                  "probabilities as KenLM does), kenlm_state_len or null, 1 if the n-gram's suffix bigram was deleted]", "models": []}
    for name, seed, order, with_unk, n_sent in (("lm_order4", 7, 4, True, 60), ("lm_order3_nounk", 12, 3, False, 70)):
        P, B, holes = build(seed, order, with_unk, n_sent)
        write_arpa(os.path.join(HERE, name + ".arpa"), P, B, order)
        D = Definition(P, B, order)
        rng = random.Random(seed + 100)
        pool = sorted(D.vocab - {"</s>"}) + ["zz", "Ab"]          # two words outside the vocabulary
        qwords = sorted(D.vocab) + ["zz"]
        queries = []
        # every (context, word) for contexts of up to 1 word; a seeded sample of the longer ones; every listed n-gram;
        # every deleted-suffix case
        ctxs = [()] + [(c,) for c in pool]
        ctxs += [tuple(rng.choice(pool) for _ in range(k)) for k in range(2, order) for _ in range(40)]
        for g in sorted(P):
            if len(g) >= 2:
                ctxs.append(tuple(reversed(g[:-1])))
        for h in holes:
            for t in sorted(P):
                if len(t) == 3 and t[1:] == h:
                    queries.append((tuple(reversed(t[:-1])), t[-1], "suffix_absent"))
        for c in sorted(set(ctxs)):
            ws = qwords if len(c) <= 1 else rng.sample(qwords, 5)
            if len(c) >= 1:                                       # always include the words the model lists after c
                ws = sorted(set(ws) | {g[-1] for g in P if len(g) == len(c) + 1 and g[:-1] == tuple(reversed(c))})
            for w in ws:
                queries.append((c, w, "grid"))
        rows = []
        for c, w, kind in queries:
            s, t = D.score(list(c), w)
            sl = D.kenlm_state_len(list(c), w)
            # state minimisation is score-neutral: from the minimised state every next word scores the same
            full = [w] + list(c)
            for w2 in rng.sample(qwords, 6 if sl is not None else 0):
                a, _ = D.score(full, w2)
                b, _ = D.score(full[:sl], w2)
                assert abs(a - b) < 1e-12, (c, w, w2, a, b)
            rows.append([" ".join(c), w, round(s, 6), round(t, 4), sl, 1 if kind == "suffix_absent" else 0])
        # sentences walked word by word from the begin-of-sentence state, as print_scores_for_sentence does
        # (src/decoders/ctc_decoder.cpp:141-151) and as the beam search does along a prefix
        sentences = []
        for _ in range(25):
            ws = [rng.choice(pool) for _ in range(rng.randint(1, 8))]
            hist, sc = ["<s>"], []
            for w in ws:
                s, t = D.score(hist, w)
                sc.append([round(s, 6), round(t, 4)])
                hist = [w] + hist
            sentences.append({"words": ws, "scores": sc})
        out["models"].append({"name": name, "arpa": name + ".arpa", "order": order, "has_unk": with_unk,
                              "n_entries": len(P), "deleted_suffix_bigrams": [" ".join(h) for h in holes],
                              "queries": rows, "sentences": sentences})
        print(name, "entries", len(P), "queries", len(rows), "holes", len(holes))
    with open(os.path.join(HERE, "lm_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()

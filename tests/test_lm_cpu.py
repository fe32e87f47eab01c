"""CPU: the ARPA reader / back-off scorer of the product library (host tables) and the oracle's C implementation against
(a) hand-computed values, (b) each other, (c) expected scores derived in pure Python straight from the definition of the
ARPA format (tests/golden/make_lm_golden.py -- the independent pin).  KenLM itself cannot be run here (absent from the
reference tree and the image): agreement with a KenLM binary on a real corpus model stays unpinned."""
import gzip
import os
import shutil

import pytest

import oracle_lib as O
from end2end_amd.engines import LanguageModel

ARPA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_3gram.arpa")
LABELS = ["_", "a", "b", " "]


def test_hand_computed_backoff_scores():
    lm = LanguageModel(ARPA, LABELS, case_sensitive=True)
    assert lm.order() == 3
    ix = {w: lm.word_index(w) for w in ["<unk>", "<s>", "</s>", "a", "ab", "b", "ba"]}
    assert ix["<unk>"] == 0 and len(set(ix.values())) == 7 and lm.word_index("zzz") == 0
    s, a, ab, bb, ba = ix["<s>"], ix["a"], ix["ab"], ix["b"], ix["ba"]
    approx = lambda x, y: abs(x - y) < 1e-6    # noqa: E731
    assert approx(lm.score([s], a), -0.4)                    # bigram "<s> a"
    assert approx(lm.score([a, s], bb), -0.2)                # trigram "<s> a b"
    assert approx(lm.score([a], ab), -0.9)                   # bigram "a ab"
    assert approx(lm.score([ab], bb), -0.2 - 1.1)            # bo(ab) + p(b)
    assert approx(lm.score([bb, a], ba), -0.2 - 0.4 - 1.3)   # bo(a b) + bo(b) + p(ba)
    assert approx(lm.score([a], 0), -0.3 - 1.0)              # bo(a) + p(<unk>)
    assert approx(lm.score([], a), -0.7)


def test_matches_oracle_lm_on_all_contexts_and_gzip(tmp_path):
    gz = tmp_path / "tiny.arpa.gz"
    with open(ARPA, "rb") as f, gzip.open(gz, "wb") as g:
        shutil.copyfileobj(f, g)
    lm = LanguageModel(str(gz), LABELS, case_sensitive=True)
    olm = O.OracleLM(ARPA)
    words = ["<unk>", "<s>", "a", "ab", "b", "ba", "nope"]
    for w in words:
        for c1 in words:
            for c2 in words:
                ctx_p = [lm.word_index(c1), lm.word_index(c2)]
                ctx_o = [olm.word_index(c1), olm.word_index(c2)]
                sp = lm.score(ctx_p, lm.word_index(w))
                so, _ = olm.base_score(ctx_o, olm.word_index(w))
                assert abs(sp - so) < 1e-6, (w, c1, c2)


def test_case_folding_and_errors(tmp_path):
    up = tmp_path / "upper.arpa"
    up.write_text(open(ARPA).read().replace("\tab", "\tAB").replace(" ab", " AB").replace("\tba", "\tBA"))
    lm = LanguageModel(str(up), LABELS, case_sensitive=False)
    assert lm.word_index("ab") == lm.word_index("AB") != 0        # query and vocabulary are both lower-cased
    lm_cs = LanguageModel(str(up), LABELS, case_sensitive=True)
    assert lm_cs.word_index("ab") == 0 and lm_cs.word_index("AB") != 0
    bad = tmp_path / "bad.arpa"
    bad.write_text("this is not an arpa file\n")
    from end2end_amd._runtime import E2EError
    with pytest.raises(E2EError, match="ARPA"):
        LanguageModel(str(bad), LABELS, True)
    with pytest.raises(E2EError, match="cannot open"):
        LanguageModel(str(tmp_path / "missing.arpa"), LABELS, True)


# ---------------------------------------------------------------------------------------------------------------------
# Independent pin: expected scores derived in pure Python from the ARPA definition (tests/golden/make_lm_golden.py,
# which shares no code with either C implementation).  Product (host tables of libe2e_ctc.so) AND oracle are held to it.
# ---------------------------------------------------------------------------------------------------------------------
import json  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "lm_golden.json")) as _f:
    LM_GOLDEN = json.load(_f)["models"]
GLABELS = ["_", "a", "b", "c", "d", "e", "'", " "]


def _tol(abs_terms):
    # float32 probabilities / back-offs (parsed from 4-decimal text) added in float32
    return 2.5e-7 * abs_terms + 1e-7


class _Product:
    def __init__(self, path):
        self.lm = LanguageModel(path, GLABELS, case_sensitive=True)

    def idx(self, w):
        return self.lm.word_index(w)

    def score(self, ctx_ids, wid):
        return self.lm.score(ctx_ids, wid)

    def order(self):
        return self.lm.order()


class _Oracle:
    def __init__(self, path):
        self.lm = O.OracleLM(path)

    def idx(self, w):
        return self.lm.word_index(w)

    def score(self, ctx_ids, wid):
        return self.lm.base_score(ctx_ids, wid)[0]

    def order(self):
        return self.lm.order()


@pytest.mark.parametrize("impl", [_Product, _Oracle], ids=["product", "oracle"])
@pytest.mark.parametrize("model", LM_GOLDEN, ids=lambda m: m["name"])
def test_scores_against_the_arpa_definition(model, impl):
    assert model["n_entries"] >= 200
    m = impl(os.path.join(GOLD, model["arpa"]))
    assert m.order() == model["order"]
    assert m.idx("<unk>") == 0 and m.idx("zz") == 0 and m.idx("Ab") == 0
    n_absent = 0
    for ctx, word, want, abs_terms, _state_len, suffix_absent in model["queries"]:
        ids = [m.idx(w) for w in ctx.split()] if ctx else []
        got = m.score(ids[: model["order"] - 1], m.idx(word))
        assert abs(got - want) <= _tol(abs_terms), (ctx, word, got, want)
        n_absent += suffix_absent
    assert n_absent >= 3          # listed trigrams whose suffix bigram is not listed are found all the same


@pytest.mark.parametrize("impl", [_Product, _Oracle], ids=["product", "oracle"])
@pytest.mark.parametrize("model", LM_GOLDEN, ids=lambda m: m["name"])
def test_state_kept_after_a_word_is_score_equivalent_to_kenlms(model, impl):
    """KenLM keeps `kenlm_state_len` words after scoring a word; the implementations here keep order-1 words.  Both must
    score every next word the same (checked from the minimised and from the full history)."""
    m = impl(os.path.join(GOLD, model["arpa"]))
    probe = [m.idx(w) for w in ("a", "b", "abc", "dead", "zz", "</s>")]
    k = 0
    for ctx, word, _want, _t, state_len, _ in model["queries"]:
        if state_len is None or not ctx:
            continue
        k += 1
        if k % 7:
            continue
        full = ([m.idx(word)] + [m.idx(w) for w in ctx.split()])[: model["order"] - 1]
        for w2 in probe:
            a, b = m.score(full, w2), m.score(full[:state_len], w2)
            assert abs(a - b) <= 1e-6, (ctx, word, state_len)


@pytest.mark.parametrize("impl", [_Product, _Oracle], ids=["product", "oracle"])
@pytest.mark.parametrize("model", LM_GOLDEN, ids=lambda m: m["name"])
def test_sentences_walked_from_begin_of_sentence(model, impl):
    m = impl(os.path.join(GOLD, model["arpa"]))
    for s in model["sentences"]:
        hist = [m.idx("<s>")]
        for w, (want, abs_terms) in zip(s["words"], s["scores"]):
            wid = m.idx(w)
            got = m.score(hist[: model["order"] - 1], wid)
            assert abs(got - want) <= _tol(abs_terms), (s["words"], w, got, want)
            hist = [wid] + hist

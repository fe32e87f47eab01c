"""CPU: the ARPA reader / back-off scorer of the product library (host tables) against hand-computed values and
against the oracle's independent C implementation.  LM parity with KenLM itself is UNPINNED (KenLM and its ARPA
fixture are absent from the reference tree); these tests pin the published ARPA back-off semantics."""
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
    from end2end_amd._lib import E2EError
    with pytest.raises(E2EError, match="ARPA"):
        LanguageModel(str(bad), LABELS, True)
    with pytest.raises(E2EError, match="cannot open"):
        LanguageModel(str(tmp_path / "missing.arpa"), LABELS, True)

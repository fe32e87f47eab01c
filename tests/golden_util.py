"""Loading helpers for tests/golden (data only)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


def beam_bruteforce_cases():
    """Exhaustive-enumeration decode answers (tests/golden/make_beam_golden.py; independent of oracle and product)."""
    with open(os.path.join(GOLDEN, "beam_bruteforce.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        labels = [chr(ord("a") + i) for i in range(c["V"])]
        labels[c["blank"]] = "_"
        if c["space_id"] >= 0:
            labels[c["space_id"]] = " "
        c["labels"] = labels
        # the empty sentence wins as the single id -1 (quirk Q6, src/decoders/ctc_decoder.cpp:232-245)
        c["want_ids"] = c["best"] if c["best"] else [-1]
    return cases


def beam_q7_cases():
    """Hand-built vectors whose answer quirk Q7 decides (tests/golden/make_beam_q7_golden.py: derivation on paper + an
    independent Python model of the reference's shared_ptr / weak_ptr ownership)."""
    with open(os.path.join(GOLDEN, "beam_q7.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        V = len(c["log_probs"][0])
        c["labels"] = [chr(ord("a") + i) for i in range(V)]
        c["labels"][c["blank"]] = "_"
    return cases


def encoder_cases():
    with open(os.path.join(GOLDEN, "encoder.json")) as f:
        return json.load(f)


_npz = {}


def npz(name):
    if name not in _npz:
        _npz[name] = np.load(os.path.join(GOLDEN, name))
    return _npz[name]


def engine_case(name):
    z = npz("loss_engine.npz")
    return {k: z[name + "/" + k] for k in ("lp", "targets", "x_len", "t_len", "losses", "grads")}


def module_case(name):
    z = npz("loss_module.npz")
    return {k: z[name + "/" + k] for k in ("input", "targets", "x_len", "t_len", "loss", "input_grad")}


def align_cases():
    """Forced alignments computed by the reference's own functions (tests/golden/make_align_golden.py)."""
    z = npz("align.npz")
    names = sorted({k.split("/")[0] for k in z.files})
    return [dict(name=n, **{k: z[n + "/" + k] for k in ("lp", "targets", "x_len", "t_len", "is_ctc", "out")}) for n in names]


def known_loss_inputs(case):
    """-> (lp float64 [B,T,V], targets, x_len, t_len, blank, cost) for a known-answer loss case."""
    p = np.array(case["probs"], dtype=np.float32)
    if case["input_kind"] == "probs_through_log_softmax":
        # the reference test feeds these numbers as LOGITS through log_softmax (tests/test_ctc.py:36-37)
        x = p.astype(np.float64)
        x = x - x.max(-1, keepdims=True)
        lp = x - np.log(np.exp(x).sum(-1, keepdims=True))
        lp = lp.astype(np.float32).astype(np.float64)
    else:
        lp = np.log(p).astype(np.float64)
    return lp, np.array(case["targets"]), np.array(case["x_len"]), np.array(case["t_len"]), case["blank"], case["cost"]

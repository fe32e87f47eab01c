#!/usr/bin/env python3
"""Generate the golden fixtures in this directory (run in the BUILD container only).

What is executed here is the REFERENCE itself:
  * its compiled loss engine, oracle/_ref/cpp_ctc_loss.so (built by
    oracle/Makefile from /root/reference/src/losses, unmodified), and
  * its Python loss module / autograd function / text encoder, imported from
    /root/reference (sys.dont_write_bytecode; the package __init__ is NOT run,
    because it eagerly imports the decoder extension, which cannot be built
    here -- KenLM headers are absent and no stand-ins are written).
The outputs (inputs + expected outputs, data only) are committed as
loss_engine.npz / loss_module.npz / encoder.json.

known_answers.json holds the known-answer vectors the reference's own tests
carry (tests/test_ctc.py:69-165, tests/test_ctc_decoder.py:44-59,86-166):
inputs and expected values only.

Nothing here is needed at test time; the fixtures are.
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

import torch  # noqa: E402
import cpp_ctc_loss  # noqa: E402  (the reference's own engine)


def import_reference_loss_module():
    # bare package objects so that submodules resolve without running
    # pytorch_end2end/__init__.py (which imports cpp_ctc_decoder)
    pkg = types.ModuleType("pytorch_end2end")
    pkg.__path__ = [os.path.join(REF, "pytorch_end2end")]
    sys.modules["pytorch_end2end"] = pkg
    from pytorch_end2end.modules.ctc_loss import CTCLoss
    from pytorch_end2end.encoders.text_encoders import CTCEncoder
    return CTCLoss, CTCEncoder


def rnd_case(seed, B, T, V, S, blank=0, dtype=torch.float32, ragged=True, min_t=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64).to(dtype)
    labels = [v for v in range(V) if v != blank]
    idx = torch.randint(0, len(labels), (B, max(S, 1)), generator=g)
    targets = torch.tensor(labels, dtype=torch.long)[idx]
    t_len = torch.randint(S // 2, S + 1, (B,), generator=g) if S > 0 else torch.zeros(B, dtype=torch.long)
    if ragged:
        lo = min_t if min_t is not None else max(1, T // 2)
        x_len = torch.randint(lo, T + 1, (B,), generator=g)
        x_len[0] = T
    else:
        x_len = torch.full((B,), T, dtype=torch.long)
    return x, targets, x_len, t_len


def gen_engine():
    """engine.compute() on log-probs (the pybind boundary, src/losses/ctc_loss_py.cpp:8-16)."""
    out = {}
    meta = []

    def add(name, lp, targets, x_len, t_len, blank):
        eng = cpp_ctc_loss.CTCLossEngine(blank)
        losses, grads = eng.compute(lp, targets, x_len, t_len)
        out[name + "/lp"] = lp.numpy()
        out[name + "/targets"] = targets.numpy()
        out[name + "/x_len"] = x_len.numpy()
        out[name + "/t_len"] = t_len.numpy()
        out[name + "/losses"] = losses.numpy()
        out[name + "/grads"] = grads.numpy()
        meta.append({"name": name, "blank": blank, "dtype": str(lp.dtype).replace("torch.", "")})

    # C1: README shape (README.md:54-71)
    x, tg, xl, tl = rnd_case(1, 4, 50, 28, 30, ragged=False)
    tl = torch.tensor([10, 17, 29, 22])
    add("c1_readme_f32", torch.log_softmax(x, -1), tg, xl, tl, 0)
    # ragged lengths -> padded frames (Q1), f32 and f64
    x, tg, xl, tl = rnd_case(2, 6, 40, 12, 12)
    add("ragged_f32", torch.log_softmax(x, -1), tg, xl, tl, 0)
    x, tg, xl, tl = rnd_case(3, 3, 33, 7, 9, dtype=torch.float64)
    add("ragged_f64", torch.log_softmax(x, -1), tg, xl, tl, 0)
    # blank != 0
    x, tg, xl, tl = rnd_case(4, 4, 30, 9, 8, blank=5)
    add("blank5_f32", torch.log_softmax(x, -1), tg, xl, tl, 5)
    x, tg, xl, tl = rnd_case(5, 3, 25, 6, 6, blank=5, dtype=torch.float64)
    add("blank_last_f64", torch.log_softmax(x, -1), tg, xl, tl, 5)
    # repeats: few symbols so that neighbours repeat often
    x, tg, xl, tl = rnd_case(6, 5, 48, 3, 16)
    add("repeats_f32", torch.log_softmax(x, -1), tg, xl, tl, 0)
    # longer lattice (several wavefronts wide), medium T
    x, tg, xl, tl = rnd_case(7, 3, 300, 29, 100, min_t=260)
    add("long_f32", torch.log_softmax(x, -1), tg, xl, tl, 0)
    # edge: empty targets (S=0), T=1, T=1&S=1, exactly feasible, infeasible (Q2)
    g = torch.Generator().manual_seed(8)
    x = torch.log_softmax(torch.randn(6, 8, 5, generator=g), -1)
    tg = torch.tensor([[1, 2, 3, 4], [1, 1, 2, 2], [3, 0, 0, 0], [2, 2, 2, 2], [1, 2, 1, 2], [4, 4, 0, 0]])
    xl = torch.tensor([8, 6, 1, 3, 4, 1])
    tl = torch.tensor([0, 4, 1, 4, 4, 0])   # row3: T=3 < S+repeats -> inf/NaN ; row1 exactly feasible
    add("edges_f32", x, tg, xl, tl, 0)
    # int32 targets/lengths are accepted (Q3) -- same numbers as ragged_f32
    # log-probs containing -inf (torch.log of exact zeros)
    p = torch.tensor([[[0.5, 0.5, 0.0], [0.0, 1.0, 0.0], [0.25, 0.25, 0.5], [1.0, 0.0, 0.0]]])
    add("neg_inf_f32", torch.log(p), torch.tensor([[1, 2]]), torch.tensor([4]), torch.tensor([2]), 0)
    # non-contiguous (time-major permuted view), as the module hands it over
    x, tg, xl, tl = rnd_case(9, 4, 20, 8, 6)
    lp_tm = torch.log_softmax(x, -1).permute(1, 0, 2).contiguous()   # (T,B,V)
    add("permuted_view_f32", lp_tm.permute(1, 0, 2), tg, xl, tl, 0)
    np.savez_compressed(os.path.join(HERE, "loss_engine.npz"), **out)
    return meta


def gen_module(CTCLoss):
    """Module-level loss and logits.grad (pytorch_end2end/modules/ctc_loss.py:25-57)."""
    out = {}
    meta = []
    combos = [
        dict(size_average=None, reduce=None, after_logsoftmax=False, time_major=False, blank_idx=0),
        dict(size_average=True, reduce=True, after_logsoftmax=False, time_major=False, blank_idx=0),
        dict(size_average=False, reduce=True, after_logsoftmax=False, time_major=True, blank_idx=0),
        dict(size_average=False, reduce=True, after_logsoftmax=True, time_major=True, blank_idx=3),
        dict(size_average=True, reduce=True, after_logsoftmax=True, time_major=False, blank_idx=0),
        dict(size_average=None, reduce=None, after_logsoftmax=True, time_major=False, blank_idx=0),
    ]
    for i, kw in enumerate(combos):
        for dt in (torch.float32, torch.float64):
            x, tg, xl, tl = rnd_case(100 + i, 4, 24, 7, 6, blank=kw["blank_idx"], dtype=dt)
            inp = torch.log_softmax(x, -1) if kw["after_logsoftmax"] else x
            if kw["time_major"]:
                inp = inp.permute(1, 0, 2).contiguous()
            inp = inp.detach().requires_grad_()
            loss = CTCLoss(**kw)(inp, tg, xl, tl)
            # weighted sum so that grad_output differs per utterance when not reduced
            w = torch.arange(1, loss.numel() + 1, dtype=loss.dtype).reshape(loss.shape) / 2.0
            (loss * w).sum().backward()
            name = "m%d_%s" % (i, "f32" if dt == torch.float32 else "f64")
            out[name + "/input"] = inp.detach().numpy()
            out[name + "/targets"] = tg.numpy()
            out[name + "/x_len"] = xl.numpy()
            out[name + "/t_len"] = tl.numpy()
            out[name + "/loss"] = loss.detach().numpy()
            out[name + "/input_grad"] = inp.grad.numpy()
            meta.append({"name": name, "kwargs": kw, "dtype": "float32" if dt == torch.float32 else "float64"})
    np.savez_compressed(os.path.join(HERE, "loss_module.npz"), **out)
    return meta


def gen_encoder(CTCEncoder):
    cases = []
    for chars, blank_id, texts in [
        ("ABCDEFGHIJKLMNOPQRSTUVWXYZ '", 0, ["hello world", "It's a TEST, ok?", ""]),
        ("abc ", 2, ["cab a", "xyz"]),
    ]:
        tf = str.upper if chars[0] == "A" else str.lower
        enc = CTCEncoder(chars, blank_id=blank_id, transform_fn=tf)
        for t in texts:
            ids = enc.encode(t).tolist()
            noisy = []
            for k in ids:
                noisy += [k, k, blank_id]
            cases.append({
                "characters": chars, "blank_id": blank_id, "transform": "upper" if tf is str.upper else "lower",
                "text": t, "clean": enc.clean(t), "encode": ids, "num_symbols": enc.num_symbols,
                "decode_in": noisy, "decode_out": enc.decode(noisy),
                "decode_pure_out": enc.decode_pure([k for k in noisy]),
            })
    with open(os.path.join(HERE, "encoder.json"), "w") as f:
        json.dump(cases, f, indent=1)


def main():
    CTCLoss, CTCEncoder = import_reference_loss_module()
    meta = {"engine": gen_engine(), "module": gen_module(CTCLoss)}
    gen_encoder(CTCEncoder)
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generator of the forced-alignment golden vectors (tests/golden/align.npz).

Runs the REFERENCE's own code -- pytorch_end2end/utils/alignment.py:_get_alignment_ctc_1d (:50-106),
_get_alignment_asg_1d (:10-47) and the batch driver get_alignment_3d (:109-138) -- imported from /root/reference in this
container.  The module decorates its functions with numba.jit, which is not installed here: a stand-in `numba` whose
`jit` returns the function unchanged is put on sys.modules for the import (generator only; nothing of this travels --
the outputs below are data).  Inputs come from seeded torch generators and are stored with the outputs.

    python tests/golden/make_align_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pytorch_end2end/utils/alignment.py"

fake = types.ModuleType("numba")
fake.jit = lambda *a, **k: (lambda f: f)
sys.modules["numba"] = fake
spec = importlib.util.spec_from_file_location("ref_alignment", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def batch_case(seed, B, T, V, S, is_ctc, sharp=1.0, dtype=torch.float32, ragged=True, repeats=False, tight=False):
    g = torch.Generator().manual_seed(seed)
    lp = torch.log_softmax(torch.randn(B, T, V, generator=g, dtype=torch.float64) * sharp, -1).to(dtype)
    tg = torch.randint(1, V, (B, max(S, 1)), generator=g)
    if repeats:
        tg[:, 1::2] = tg[:, 0::2][:, : tg[:, 1::2].shape[1]]          # doubled labels: the skip transition is forbidden
    t_len = torch.randint(max(S // 2, 1 if not is_ctc else 0), S + 1, (B,), generator=g) if S > 0 else torch.zeros(B, dtype=torch.long)
    x_len = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g) if ragged else torch.full((B,), T)
    x_len[0] = T
    if tight:                                                            # as few frames as the labelling allows
        for b in range(B):
            need = int(t_len[b]) + (int((tg[b, 1:t_len[b]] == tg[b, :t_len[b] - 1]).sum()) if is_ctc and t_len[b] > 1 else 0)
            x_len[b] = max(need, 1)
        T = int(x_len.max())
        lp = lp[:, :T].contiguous()
    if not is_ctc:
        x_len = torch.maximum(x_len, t_len)                              # ASG needs a frame per label
        t_len = torch.clamp(t_len, min=1)
    out = ref.get_alignment_3d(lp, tg, x_len, t_len, is_ctc=is_ctc)
    return dict(lp=lp.numpy(), targets=tg.numpy(), x_len=x_len.numpy(), t_len=t_len.numpy(),
                is_ctc=np.array(int(is_ctc)), out=out.numpy())


def main():
    cases = {
        "ctc_small": batch_case(1, 4, 12, 5, 4, True),
        "ctc_medium": batch_case(2, 6, 80, 12, 25, True, sharp=2.0),
        "ctc_repeats": batch_case(3, 5, 40, 6, 14, True, repeats=True),
        "ctc_tight": batch_case(4, 5, 40, 7, 12, True, repeats=True, tight=True),
        "ctc_long": batch_case(5, 3, 400, 29, 150, True, sharp=3.0),
        "ctc_t1": batch_case(6, 3, 1, 5, 1, True, ragged=False),
        "ctc_f64": batch_case(7, 3, 30, 6, 8, True, dtype=torch.float64),
        "ctc_flat_ties": batch_case(8, 3, 24, 5, 6, True, sharp=0.0),
        "asg_small": batch_case(11, 4, 12, 5, 4, False),
        "asg_medium": batch_case(12, 5, 90, 10, 30, False, sharp=2.0),
        "asg_tight": batch_case(13, 4, 30, 6, 12, False, tight=True),
    }
    # empty targets (only blanks), and an alignment with too few frames (the reference does not reject it)
    c = batch_case(9, 3, 10, 4, 3, True, ragged=False)
    c["t_len"][:] = [0, 3, 2]
    c["out"] = ref.get_alignment_3d(torch.from_numpy(c["lp"]), torch.from_numpy(c["targets"]), torch.from_numpy(c["x_len"]),
                                    torch.from_numpy(c["t_len"]), is_ctc=True).numpy()
    cases["ctc_empty_target"] = c
    c = batch_case(10, 2, 6, 4, 5, True, ragged=False)
    c["t_len"][:] = [5, 5]
    c["x_len"][:] = [3, 6]
    c["out"] = ref.get_alignment_3d(torch.from_numpy(c["lp"]), torch.from_numpy(c["targets"]), torch.from_numpy(c["x_len"]),
                                    torch.from_numpy(c["t_len"]), is_ctc=True).numpy()
    cases["ctc_too_few_frames"] = c
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[name + "/" + k] = v
        print(name, c["lp"].shape, "x_len", c["x_len"].tolist(), "t_len", c["t_len"].tolist())
    np.savez_compressed(os.path.join(HERE, "align.npz"), **flat)


if __name__ == "__main__":
    main()

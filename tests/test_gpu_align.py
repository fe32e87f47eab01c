"""-m gpu: Viterbi forced alignment (e2e_ctc_align) against the outputs of the reference's own functions
(tests/golden/align.npz, pytorch_end2end/utils/alignment.py run by make_align_golden.py) and the oracle."""
import numpy as np
import pytest
import torch

import golden_util as G
import oracle_lib as O
import gpu_util as U
from end2end_amd import _lib

pytestmark = pytest.mark.gpu


def c_abi_align(lp, targets, x_len, t_len, blank=0, is_ctc=True, pad=-100):
    L = _lib.load()
    d = U.dev()
    if not lp.is_cuda:
        base = lp
        lp = torch.empty_strided(base.shape, base.stride(), dtype=base.dtype, device=d)
        lp.copy_(base)
    B, T, V = lp.shape
    tg = torch.as_tensor(np.asarray(targets)).to(d, torch.long).reshape(B, -1).contiguous()
    if tg.shape[1] == 0:
        tg = torch.zeros((B, 1), dtype=torch.long, device=d)
    xl = torch.as_tensor(np.asarray(x_len)).to(d, torch.long)
    tl = torch.as_tensor(np.asarray(t_len)).to(d, torch.long)
    out = torch.full((B, T), 7, dtype=torch.long, device=d)
    n = L.e2e_ctc_align_workspace_bytes(B, T, V, tg.shape[1], int(is_ctc))
    ws = torch.empty(n, dtype=torch.uint8, device=d)
    sB, sT, sV = lp.stride()
    _lib.check(L.e2e_ctc_align(lp.data_ptr(), _lib.dtype_code(lp.dtype), sB, sT, sV, tg.data_ptr(), tg.stride(0),
                               xl.data_ptr(), tl.data_ptr(), B, T, V, tg.shape[1], blank, int(is_ctc), out.data_ptr(), pad,
                               ws.data_ptr(), ws.numel(), _lib.stream_ptr(d)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("case", G.align_cases(), ids=lambda c: c["name"])
def test_against_the_reference_functions(case):
    lp = torch.from_numpy(case["lp"])
    got = c_abi_align(lp, case["targets"], case["x_len"], case["t_len"], 0, bool(case["is_ctc"]))
    assert got.tolist() == case["out"].tolist()
    # a time-major permuted view (no copy) and the other precision give the same labelling
    tm = lp.permute(1, 0, 2).contiguous().permute(1, 0, 2)
    assert c_abi_align(tm, case["targets"], case["x_len"], case["t_len"], 0, bool(case["is_ctc"])).tolist() == case["out"].tolist()


def test_python_surface_matches_upstream_signature_and_result():
    from pytorch_end2end.utils.alignment import get_alignment_3d
    c = [x for x in G.align_cases() if x["name"] == "ctc_medium"][0]
    args = (torch.from_numpy(c["lp"]), torch.from_numpy(c["targets"]), torch.from_numpy(c["x_len"]), torch.from_numpy(c["t_len"]))
    for dev in ("cuda", "cpu"):
        out = get_alignment_3d(*(a.to(dev) for a in args))
        assert out.device.type == "cpu" and out.dtype == torch.long and out.tolist() == c["out"].tolist()


@pytest.mark.parametrize("is_ctc", [True, False])
def test_speech_shape_against_oracle_and_properties(is_ctc):
    # B=64, T=1000, V=29, S<=200 (the loss's headline shape): identical to the oracle; the labelling collapses to the
    # targets; its score is the best over a few perturbed labellings
    g = torch.Generator().manual_seed(31)
    B, T, V, S = 64, 1000, 29, 200
    lp = torch.log_softmax(torch.randn(B, T, V, generator=g) * 2.0, -1)
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    xl = torch.randint(T // 2, T + 1, (B,), generator=g); xl[0] = T
    got = c_abi_align(lp, tg, xl, tl, 0, is_ctc)
    want = O.ctc_align(lp.double().numpy(), tg.numpy(), xl.numpy(), tl.numpy(), 0, is_ctc, n_threads=16)
    assert np.array_equal(got, want)
    for b in (0, 17, 63):
        a = got[b, : xl[b]]
        assert (got[b, xl[b]:] == -100).all()
        keep = np.ones(len(a), bool); keep[1:] = a[1:] != a[:-1]
        collapsed = a[keep] if not is_ctc else a[keep][a[keep] != 0]
        if is_ctc:
            assert collapsed.tolist() == tg[b, : tl[b]].tolist()


def test_invalid_rows_are_padding_and_errors_are_reported():
    lp = torch.log_softmax(torch.randn(3, 6, 4), -1)
    got = c_abi_align(lp, [[1, 2], [1, 9], [1, 2]], [6, 6, 9], [2, 2, 2])
    assert (got[1] == -100).all() and (got[2] == -100).all() and (got[0] >= 0).all()
    L = _lib.load()
    assert L.e2e_ctc_align(None, 7, 1, 1, 1, None, 0, None, None, 1, 1, 1, 0, 0, 1, None, 0, None, 0, None) == -1


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_sixteen_bit_log_probabilities_are_aligned_as_their_f32_images(dtype):
    """get_alignment_3d reads f16 / bf16 log-probabilities as they are (no up-cast pass): the result is the f32 call's on the same
    values, bit for bit."""
    from end2end_amd.utils.alignment import get_alignment_3d
    g = torch.Generator().manual_seed(8)
    lp16 = torch.log_softmax(torch.randn(4, 120, 29, generator=g) * 2, -1).to(dtype)
    tg = torch.randint(1, 29, (4, 30), generator=g)
    xl, tl = torch.tensor([120, 100, 77, 120]), torch.tensor([30, 12, 25, 1])
    a16 = get_alignment_3d(lp16.cuda(), tg, xl, tl)
    a32 = get_alignment_3d(lp16.float().cuda(), tg, xl, tl)
    assert torch.equal(torch.as_tensor(a16), torch.as_tensor(a32))

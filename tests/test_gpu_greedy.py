"""-m gpu: HIP greedy decode through the C ABI -- bit-exact against the oracle and the known answers."""
import numpy as np
import pytest
import torch

import golden_util as G
import oracle_lib as O
import gpu_util as U

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", G.known_answers()["decode"], ids=lambda c: c["name"])
def test_known_answers(case):
    x = np.array(case["x"], dtype=np.float32)
    if case["input_kind"] == "log_of_probs":
        x = np.log(x)
    B, T, _ = x.shape
    xl = case.get("x_len") or [T] * B
    out, lens = U.c_abi_greedy(torch.from_numpy(x), xl, case["blank"])
    sent = ["".join(case["labels"][i] for i in out[b, : lens[b]]) for b in range(B)]
    assert sent == case["greedy"]
    if "greedy_targets" in case:
        assert out.tolist() == case["greedy_targets"] and lens.tolist() == case["greedy_lengths"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(7, 300, 29), (3, 1500, 29), (4, 70, 5), (3, 40, 200), (2, 33, 8000), (5, 1, 3),
                                   (3, 600, 32), (2, 257, 64), (3, 300, 30)])      # even alphabets: padded LDS rows
def test_random_bit_exact(shape, dtype):
    g = torch.Generator().manual_seed(sum(shape))
    B, T, V = shape
    x = (torch.randn(B, T, V, generator=g, dtype=torch.float64) * 3).to(dtype)
    x[:, ::7] = x[:, ::7].round()            # exact ties
    xl = torch.randint(1, T + 1, (B,), generator=g)
    xl[0] = T
    for blank in (0, V - 1):
        out, lens = U.c_abi_greedy(x, xl, blank)
        o_out, o_len = O.ctc_greedy(x.double().numpy(), xl.numpy(), blank)
        assert np.array_equal(lens, o_len)
        assert np.array_equal(out, o_out)        # includes the zero padding (Q5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(6, 500, 29), (3, 70, 5), (2, 40, 200), (2, 33, 8000), (3, 257, 64)])
def test_sixteen_bit_logits_are_compared_as_they_are(shape, dtype):
    """Native 16-bit input: argmax in the source dtype's ordering (what torch.argmax of that tensor gives), first maximum on
    ties -- and rounding to 16 bits makes ties frequent --, bit-exact against the oracle on the same rounded values; both
    the streaming kernel (contiguous small alphabets) and the general one (a strided view, wide alphabets)."""
    g = torch.Generator().manual_seed(sum(shape) + 1)
    B, T, V = shape
    x = (torch.randn(B, T, V, generator=g) * 3).to(dtype)
    xl = torch.randint(1, T + 1, (B,), generator=g)
    xl[0] = T
    for view in (x, x.permute(1, 0, 2).contiguous().permute(1, 0, 2)):
        out, lens = U.c_abi_greedy(view, xl, 0)
        o_out, o_len = O.ctc_greedy(x.double().numpy(), xl.numpy(), 0)
        assert np.array_equal(lens, o_len) and np.array_equal(out, o_out)


def test_time_major_view_and_full_c3_properties():
    g = torch.Generator().manual_seed(3)
    B, T, V = 1024, 1500, 29
    x = torch.randn(B, T, V, generator=g) * 3
    xl = torch.randint(T // 2, T + 1, (B,), generator=g)
    out, lens = U.c_abi_greedy(x, xl, 0)
    # idempotence: decoding the decoded one-hot sequence gives itself back
    am = x.argmax(-1).numpy()
    for b in (0, 17, 511, 1023):
        n = int(xl[b])
        want = [int(s) for k, s in enumerate(am[b, :n]) if s != 0 and (k == 0 or s != am[b, k - 1])]
        assert out[b, : lens[b]].tolist() == want and not out[b, lens[b]:].any()
    assert (lens <= xl.numpy()).all()
    # a time-major (T,B,V) tensor viewed batch-major
    xt = x[:8].permute(1, 0, 2).contiguous().permute(1, 0, 2)
    o2, l2 = U.c_abi_greedy(xt, xl[:8], 0)
    assert np.array_equal(o2, out[:8]) and np.array_equal(l2, lens[:8])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_streaming_kernel_corner_cases(dtype):
    """Contiguous small-alphabet input takes the streaming kernel (64-frame chunks per wave, 1024-frame super-tiles): several
    super-tiles with the collapse carried across them, NaN / inf rows (torch.argmax: first NaN wins, else first maximum),
    utterances that end inside a chunk -- and the same data through an unaligned view, which takes the tiled kernel."""
    g = torch.Generator().manual_seed(99)
    B, T, V = 3, 5000, 29
    x = (torch.randn(B, T, V, generator=g, dtype=torch.float64) * 2).to(dtype)
    x[0, 5::97, 7] = float("nan"); x[0, 5::97, 3] = float("nan")       # two NaNs in a row: the first one wins
    x[1, ::211] = float("-inf"); x[1, ::211, 11] = float("inf")
    x[2, 1000:1100] = 0.0                                                # a long run of exact ties (symbol 0)
    xl = torch.tensor([T, 4097, 1025])
    for blank in (0, 7):
        out, lens = U.c_abi_greedy(x, xl, blank)
        o_out, o_len = O.ctc_greedy(x.double().numpy(), xl.numpy(), blank)
        assert np.array_equal(lens, o_len) and np.array_equal(out, o_out)
        # unaligned: drop the first column of a wider tensor (row stride 30, element offset 1)
        wide = torch.zeros(B, T, V + 1, dtype=dtype)
        wide[:, :, 1:] = x
        o2, l2 = U.c_abi_greedy(wide[:, :, 1:], xl, blank)
        assert np.array_equal(l2, o_len) and np.array_equal(o2, o_out)

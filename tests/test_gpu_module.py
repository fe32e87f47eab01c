"""-m gpu: the Python surface (CTCLoss / CTCDecoder with the reference's kwargs) against fixtures captured from the
reference's own Python module + engine (tests/golden/loss_module.npz) and the reference tests' decode answers."""
import numpy as np
import pytest
import torch

import golden_util as G
import oracle_lib as O
import gpu_util as U
from end2end_amd import CTCDecoder, CTCLoss, DecoderResults, _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("m", G.meta()["module"], ids=lambda m: m["name"])
@pytest.mark.parametrize("device", ["cuda", "cpu"])
def test_module_loss_and_input_grad(m, fused, device):
    c = G.module_case(m["name"])
    kw = m["kwargs"]
    x = torch.from_numpy(c["input"]).to(device).requires_grad_()
    tg, xl, tl = (torch.from_numpy(c[k]).to(device) for k in ("targets", "x_len", "t_len"))
    loss = CTCLoss(fused=fused, **kw)(x, tg, xl, tl)
    assert loss.device.type == device and loss.dtype == x.dtype
    w = torch.arange(1, loss.numel() + 1, dtype=loss.dtype, device=loss.device).reshape(loss.shape) / 2.0
    (loss * w).sum().backward()
    rt, at = (1e-9, 1e-11) if m["dtype"] == "float64" else (1e-4, 2e-6)
    U.assert_same(loss.detach().cpu().numpy(), c["loss"], rt, at, "loss")
    U.assert_same(x.grad.cpu().numpy(), c["input_grad"], rt, at, "input grad")


def test_gradcheck_like_the_reference():
    # tests/test_ctc.py:168-191
    k = G.known_answers()["gradcheck"]
    np.random.seed(k["np_seed"])
    B, Tm, A, Sm = k["batch_size"], k["max_sequence_len"], k["alphabet_size"], k["max_targets_len"]
    tl = np.random.randint(low=1, high=Sm + 1, size=B)
    xl = tl + np.random.randint(low=0, high=(Tm - Sm + 1), size=B)
    logits = np.random.randn(B, Tm, A + 1)
    tg = (1 + np.random.rand(B, np.max(tl)) * A).astype(np.int64)
    d = U.dev()
    inp = (torch.tensor(logits, dtype=torch.float64, device=d).requires_grad_(), torch.tensor(tg, device=d),
           torch.tensor(xl, device=d), torch.tensor(tl, device=d))
    assert torch.autograd.gradcheck(CTCLoss(blank_idx=0, time_major=False, after_logsoftmax=False), inp,
                                    eps=k["eps"], atol=k["atol"])


def test_readme_example_runs_and_matches_oracle():
    # README.md:54-71
    torch.manual_seed(0)
    ctc_loss = CTCLoss(blank_idx=0, time_major=False, reduce=True, size_average=True, after_logsoftmax=False)
    B, V = 4, 28
    logits = torch.randn(B, 50, V).cuda().detach().requires_grad_()
    targets = torch.randint(1, V, (B, 30), dtype=torch.long)
    xl = torch.full((B,), 50, dtype=torch.long)
    tl = torch.randint(10, 30, (B,), dtype=torch.long)
    loss = ctc_loss(logits, targets, xl, tl)
    loss.backward()
    lp = torch.log_softmax(logits.detach().cpu().double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, targets.numpy(), xl.numpy(), tl.numpy(), 0)
    assert abs(loss.item() - l_o.mean()) < 1e-4 * l_o.mean()
    U.assert_same(logits.grad.cpu().numpy(), g_o / B, 1e-4, 2e-6, "grad")


@pytest.mark.parametrize("case", G.known_answers()["decode"], ids=lambda c: c["name"])
def test_decoder_known_answers(case):
    x = torch.tensor(case["x"], dtype=torch.float32)
    after = case["input_kind"] == "log_of_probs"
    if after:
        x = torch.log(x)
    xl = torch.tensor(case["x_len"]) if "x_len" in case else None
    dec = CTCDecoder(beam_width=case["beam_width"], after_logsoftmax=after, blank_idx=case["blank"], time_major=False,
                     labels=case["labels"], wip=case.get("wip", 0.0))
    for dev in ("cuda", "cpu"):
        xd = x.to(dev)
        res = dec.decode_greedy(xd, xl.to(dev) if xl is not None else None)
        assert isinstance(res, DecoderResults) and res.decoded_sentences == case["greedy"]
        # results are CPU tensors whatever the input's device, as upstream (ctc_decoder.cpp:157,449)
        assert res.decoded_targets.device.type == "cpu" and res.decoded_targets.dtype == torch.long
        if "greedy_targets" in case:
            assert res.decoded_targets.tolist() == case["greedy_targets"]
            assert res.decoded_targets_lengths.tolist() == case["greedy_lengths"]
        res = dec.decode(xd, xl.to(dev) if xl is not None else None)
        want = case["greedy"] if case["beam_width"] == 1 else case["beam"]      # beam_width == 1 routes to greedy
        assert res.decoded_sentences == want


def test_decoder_time_major_and_raw_logits():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 20, 6, generator=g) * 2
    labels = ["_", "a", "b", "c", " ", "d"]
    d_bm = CTCDecoder(beam_width=15, blank_idx=0, labels=labels, wip=0.5)
    d_tm = CTCDecoder(beam_width=15, blank_idx=0, labels=labels, wip=0.5, time_major=True)
    a = d_bm.decode(x.cuda())
    b = d_tm.decode(x.permute(1, 0, 2).contiguous().cuda())
    assert a.decoded_sentences == b.decoded_sentences
    _, _, want = O.ctc_beam(torch.log_softmax(x.double(), -1).numpy(), None, 0, 15, labels, wip=0.5)
    assert a.decoded_sentences == want


@pytest.mark.parametrize("m", G.meta()["module"], ids=lambda m: m["name"])
def test_reference_call_pattern_through_the_reference_module_names(m):
    """The reference's own module code path, statement by statement (pytorch_end2end/modules/ctc_loss.py:37-57,74-75 and
    functions/forward_backward.py:18-35), with the engine found the way upstream finds it:
    import_module("cpp_ctc_loss").CTCLossEngine(blank_idx).  Fixtures: the reference module's own outputs."""
    from importlib import import_module
    import torch.nn.functional as F
    engine = import_module("cpp_ctc_loss").CTCLossEngine(m["kwargs"].get("blank_idx", 0))

    class RefStyleFunction(torch.autograd.Function):
        @staticmethod
        def forward(ctx, engine, logits, targets, logits_lengths, targets_lengths):
            loss, grads = engine.compute(logits, targets, logits_lengths, targets_lengths)
            ctx.grads = grads
            return loss

        @staticmethod
        def backward(ctx, grad_output):
            loss_grads = ctx.grads
            if grad_output.is_cuda:
                loss_grads = loss_grads.cuda(grad_output.get_device())
            return None, loss_grads.contiguous() * grad_output.contiguous().view(-1, 1, 1), None, None, None

    c = G.module_case(m["name"])
    kw = m["kwargs"]
    x = torch.from_numpy(c["input"]).cuda().requires_grad_()
    tg, xl, tl = (torch.from_numpy(c[k]) for k in ("targets", "x_len", "t_len"))      # CPU ints, as users pass them
    lsm = x if kw.get("after_logsoftmax") else F.log_softmax(x, dim=2)
    if kw.get("time_major"):
        lsm = lsm.permute(1, 0, 2)
    loss = RefStyleFunction.apply(engine, lsm, tg, xl, tl)
    if kw.get("reduce"):
        loss = loss.mean() if kw.get("size_average") else loss.sum()
    w = torch.arange(1, loss.numel() + 1, dtype=loss.dtype, device=loss.device).reshape(loss.shape) / 2.0
    (loss * w).sum().backward()
    rt, at = (1e-9, 1e-11) if m["dtype"] == "float64" else (1e-4, 2e-6)
    U.assert_same(loss.detach().cpu().numpy(), c["loss"], rt, at, "loss")
    U.assert_same(x.grad.cpu().numpy(), c["input_grad"], rt, at, "input grad")


def test_reference_decoder_engine_name_keywords_and_cpu_results():
    """cpp_ctc_decoder.CTCDecoder called the way pytorch_end2end/decoders/ctc_decoder.py:61-64,108-110,145-147 calls it:
    positional constructor arguments, CPU tensors, the keywords logits_= / logits_lengths_=; results are CPU tensors."""
    import cpp_ctc_decoder
    for case in G.known_answers()["decode"]:
        x = torch.tensor(case["x"], dtype=torch.float32)
        lp = torch.log(x) if case["input_kind"] == "log_of_probs" else torch.log_softmax(x, -1)
        xl = torch.tensor(case["x_len"], dtype=torch.int32) if "x_len" in case else \
            torch.zeros(x.shape[0], dtype=torch.int).fill_(x.shape[1])
        dec = cpp_ctc_decoder.CTCDecoder(case["blank"], max(case["beam_width"], 2), case["labels"], "", 1.0,
                                         case.get("wip", 0.0), -10, True)
        ids, lens, sents = dec.decode_greedy(logits_=x, logits_lengths_=xl)
        assert sents == case["greedy"] and ids.device.type == "cpu" and lens.device.type == "cpu"
        assert ids.shape == x.shape[:2]                                   # (B, Tmax) zero padded (quirk Q5)
        if case["beam_width"] > 1:
            ids, lens, sents = dec.decode(logits_=lp, logits_lengths_=xl)
            assert sents == case["beam"] and ids.device.type == "cpu" and ids.dtype == torch.long
            assert ids.shape[1] == int(lens.max())                        # packed to the longest (ctc_decoder.cpp:192)


def test_backward_scales_in_place_and_a_retained_graph_can_be_walked_again():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 12, 5, generator=g).cuda().requires_grad_()
    tg, xl, tl = torch.tensor([[1, 2, 2], [3, 1, 0], [4, 4, 4]]), torch.tensor([12, 9, 12]), torch.tensor([3, 2, 3])
    loss = CTCLoss()(x, tg, xl, tl)
    w = torch.tensor([0.5, 0.0, -2.0], device="cuda")
    (loss * w).sum().backward(retain_graph=True)
    g1 = x.grad.clone()
    x.grad = None
    (loss * w).sum().backward()            # the first walk gave its buffer away: the engine is asked again
    assert torch.equal(g1, x.grad)
    lp = torch.log_softmax(x.detach().cpu().double(), -1).numpy()
    _, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    for b in range(3):
        g_o[b, xl[b]:] = 0
    U.assert_same(g1.cpu().numpy(), g_o * w.cpu().numpy()[:, None, None], 1e-4, 2e-6, "scaled grad")
    assert (g1[1] == 0).all()


@pytest.mark.parametrize("size_average", [True, False])
def test_reduced_module_path_matches_unreduced_and_scaled_backward(size_average):
    g = torch.Generator().manual_seed(8)
    x0 = torch.randn(5, 30, 9, generator=g)
    tg = torch.randint(1, 9, (5, 6), generator=g)
    xl, tl = torch.tensor([30, 22, 30, 17, 30]), torch.tensor([6, 3, 0, 6, 5])
    xa = x0.clone().cuda().requires_grad_()
    xb = x0.clone().cuda().requires_grad_()
    red = CTCLoss(reduce=True, size_average=size_average)(xa, tg, xl, tl)
    vec = CTCLoss()(xb, tg, xl, tl)
    want = vec.mean() if size_average else vec.sum()
    assert red.shape == want.shape and abs(red.item() - want.item()) <= 1e-6 * abs(want.item())
    (red * 3.0).backward()
    (want * 3.0).backward()
    U.assert_same(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), 1e-6, 1e-12, "input grad")


def test_decoder_rejects_an_alphabet_that_does_not_match_the_labels():
    dec = CTCDecoder(beam_width=4, labels=["_", "a", "b"])
    with pytest.raises(ValueError, match="labels"):
        dec.decode(torch.randn(1, 5, 4).cuda())


def test_engine_with_f32_chains_keeps_the_interface_and_the_stated_tolerance():
    """CTCLossEngine(blank, f32_chains=True): same call, same result types; long targets go through the packed-f32 chains."""
    from end2end_amd.engines import CTCLossEngine
    g = torch.Generator().manual_seed(12)
    B, T, V, S = 3, 400, 29, 180
    lp = torch.log_softmax(torch.randn(B, T, V, generator=g), -1)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl, tl = torch.tensor([T, T - 9, T]), torch.tensor([S, 150, 131])
    l0, g0 = CTCLossEngine(0).compute(lp, tg, xl, tl)
    l1, g1 = CTCLossEngine(0, f32_chains=True).compute(lp, tg, xl, tl)
    assert l1.device == lp.device and l1.dtype == lp.dtype and g1.shape == lp.shape
    U.assert_same(l1.numpy(), l0.numpy(), 2e-6, 1e-6, "losses")
    U.assert_same(g1.numpy(), g0.numpy(), 1e-4, 2e-5, "grads")
    assert not torch.equal(g0, g1)          # (a different kernel did run)


def test_module_with_f32_chains_trains_like_the_default():
    g = torch.Generator().manual_seed(21)
    B, T, V, S = 3, 320, 29, 150
    x0 = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl, tl = torch.tensor([T, T - 5, T]), torch.tensor([S, 140, 129])
    xa = x0.clone().cuda().requires_grad_()
    xb = x0.clone().cuda().requires_grad_()
    la = CTCLoss(reduce=True, size_average=True)(xa, tg, xl, tl)
    lb = CTCLoss(reduce=True, size_average=True, f32_chains=True)(xb, tg, xl, tl)
    la.backward(); lb.backward()
    assert abs(la.item() - lb.item()) <= 2e-6 * abs(la.item())
    U.assert_same(xb.grad.cpu().numpy() * B, xa.grad.cpu().numpy() * B, 1e-4, 2e-5, "input grad")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("reduce", [False, True])
def test_half_precision_logits_go_forward_and_backward(dtype, reduce):
    """Raw autocast outputs: the reference casts to double and back (src/losses/forward_backward.cpp:15,55-56); here the
    kernels read the 16-bit logits as they are, the lattice runs in f32 and loss / gradient come back in the source dtype."""
    g = torch.Generator().manual_seed(31)
    B, T, V, S = 3, 40, 11, 7
    x0 = torch.randn(B, T, V, generator=g).to(dtype)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl, tl = torch.tensor([T, T - 6, T]), torch.tensor([S, 4, 5])
    x = x0.clone().cuda().requires_grad_()
    loss = CTCLoss(reduce=reduce, size_average=True)(x, tg, xl, tl)
    assert loss.dtype == dtype
    w = torch.tensor([0.5, 1.0, -2.0], device="cuda", dtype=dtype)
    (loss * 2.0 if reduce else (loss * w).sum()).backward()
    assert x.grad.dtype == dtype and x.grad.shape == x.shape
    lp = torch.log_softmax(x0.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0
    want = g_o * (2.0 / B if reduce else w.float().cpu().numpy()[:, None, None])
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10            # the source dtype's rounding, twice
    U.assert_same(x.grad.float().cpu().numpy(), want, 4 * eps, 4 * eps, "input grad")
    got_l = loss.detach().float().cpu().numpy()
    U.assert_same(got_l, l_o.mean() if reduce else l_o, 4 * eps, 4 * eps, "loss")


@pytest.mark.parametrize("shape", [(8, 200, 48, 30), (4, 64, 8000, 20), (4, 200, 150, 120), (2, 200, 8000, 150), (2, 40, 9000, 20), (3, 50, 8001, 15)],
                         ids=["fast_path_V48", "wide_path_V8000", "fast_path_V150", "wide_path_V8000_151_columns",
                              "wide_two_pass_V9000", "wide_two_pass_V8001_unaligned"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_half_precision_logits_are_read_and_written_natively(dtype, shape):
    """No f32 copy of the (B,T,V) logits or of the gradient is made for 16-bit inputs: the allocator's peak over a
    forward + backward stays below what ONE f32 copy would add (the call allocates the 16-bit gradient, the B losses and
    nothing of the tensor's size besides; the workspace is cached by the warm-up call).  Values against the oracle on the
    rounded logits, at the source dtype's resolution."""
    B, T, V, S = shape
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(B, T, V, generator=g) * 1.5).to(dtype)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.full((B,), T); xl[1] = T - 9
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    crit = CTCLoss(reduce=True, size_average=True)

    def run():
        x = x0.clone().cuda().requires_grad_()
        loss = crit(x, tg, xl, tl)
        loss.backward()
        return x, loss
    run()                                                   # warm-up: workspace, module state
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    x, loss = run()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    n16 = B * T * V * 2
    # x (clone) + its gradient + small change; an f32 copy of either tensor would add 2 * n16 on top
    assert peak < 2 * n16 + n16 // 2 + (1 << 16), "peak %d bytes for a %d-byte tensor: something of f32 size was allocated" % (peak, n16)
    assert x.grad.dtype == dtype and loss.dtype == dtype
    lp = torch.log_softmax(x0.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    U.assert_same(x.grad.float().cpu().numpy(), g_o / B, 4 * eps, 4 * eps / B, "input grad")
    U.assert_same(loss.detach().float().cpu().numpy(), l_o.mean(), 4 * eps, 4 * eps, "loss")


@pytest.mark.parametrize("shape", [(4, 64, 8000, 20), (4, 100, 150, 60)], ids=["wide_path_V8000", "fast_path_V150"])
def test_f16_gradients_with_a_fused_loss_scale_keep_their_resolution(shape):
    """e2e_ctc_loss_opts.grad_scale = 1024 (a loss-scaling factor) on f16 logits.  The wide path keeps a row's
    exponentials packed in f16 between its passes; unscaled, values below 6e-5 sat in f16's subnormals and the factor
    multiplied their absolute error up.  Every written element must be within 1.5 ulp (f16) of the exact scaled gradient,
    small elements included."""
    import ctypes
    B, T, V, S = shape
    g = torch.Generator().manual_seed(9)
    x0 = (torch.randn(B, T, V, generator=g) * 2.0).to(torch.float16)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.full((B,), T); tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    scale = 1024.0
    losses, grads, _ = U.c_abi_loss(x0, tg, xl, tl, 0, False, _lib.ALGO_AUTO, opts=(scale, _lib.REDUCE_NONE))
    lp = torch.log_softmax(x0.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    want = g_o * scale
    got = grads.astype(np.float64)
    ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10          # an f16 ulp of the exact value (normal range)
    err = np.abs(got - want)
    # (the lattice itself is f32: 2e-6 * scale absolute on top of the output rounding)
    assert (err <= 1.5 * ulp + 2e-6 * scale + 1e-4 * np.abs(want)).all(), "worst %.3g ulp at |g| = %.3g" % (
        (err / ulp).max(), np.abs(want).flat[(err / ulp).argmax()])
    U.assert_same(losses, l_o, 1e-4, 2e-5, "losses")

"""-m gpu: runtime contract of the C ABI (asynchronous on the caller's stream, no allocation / sync inside -> graph
capturable) and an end-to-end training loop through the drop-in modules."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
import gpu_util as U
from end2end_amd import CTCDecoder, CTCLoss, _lib

pytestmark = pytest.mark.gpu


def _batch(seed, B=6, T=90, V=12, S=14):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.randint(T // 2 + S, T + 1, (B,), generator=g)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    return x, tg, xl, tl


def _raw_call(L, x, tg, xl, tl, losses, grads, ws, stream):
    _lib.check(L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), _lib.F32, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(),
                                      tl.data_ptr(), x.shape[0], x.shape[1], x.shape[2], tg.shape[1], 0,
                                      losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), _lib.ALGO_AUTO,
                                      stream))


def test_loss_call_is_graph_capturable_and_replays_on_new_data():
    import ctypes as C
    L = _lib.load()
    d = U.dev()
    x, tg, xl, tl = (t.to(d) for t in _batch(1))
    B, T, V = x.shape
    losses = torch.zeros(B, device=d)
    grads = torch.zeros(B, T, V, device=d)
    ws = torch.empty(L.e2e_ctc_loss_workspace_bytes(B, T, V, tg.shape[1], _lib.F32, _lib.ALGO_AUTO), dtype=torch.uint8, device=d)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _raw_call(L, x, tg, xl, tl, losses, grads, ws, C.c_void_p(s.cuda_stream))      # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _raw_call(L, x, tg, xl, tl, losses, grads, ws, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    x2, _, _, _ = _batch(2)
    x.copy_(x2.to(d))                    # new logits in the captured buffers
    losses.zero_(); grads.zero_()
    graph.replay()
    torch.cuda.synchronize()
    lp = torch.log_softmax(x2.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.cpu().numpy(), xl.cpu().numpy(), tl.cpu().numpy(), 0)
    for b in range(B):
        g_o[b, int(xl[b]):] = 0.0
    U.assert_same(losses.cpu().numpy(), l_o, 1e-4, 2e-6, "losses after replay")
    U.assert_same(grads.cpu().numpy(), g_o, 1e-4, 2e-6, "grads after replay")


def test_modules_follow_the_current_stream():
    d = U.dev()
    x, tg, xl, tl = _batch(3)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xd = x.to(d, non_blocking=True).requires_grad_()
        loss = CTCLoss(reduce=True, size_average=False)(xd, tg.to(d), xl.to(d), tl.to(d))
        loss.backward()
        dec = CTCDecoder(beam_width=8, blank_idx=0).decode(xd.detach(), xl.to(d))
    side.synchronize()
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, _ = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    assert abs(loss.item() - l_o.sum()) < 1e-4 * l_o.sum()
    o_ids, o_len, _ = O.ctc_beam(lp, xl.numpy(), 0, 8, None, wip=1.0)
    assert dec.decoded_targets_lengths.tolist() == o_len.tolist() and dec.decoded_targets.tolist() == o_ids.tolist()


def test_tiny_model_trains_with_the_drop_in_loss_and_decodes_its_targets():
    # a linear "acoustic model" overfits 8 utterances; greedy and beam decoding then return the transcripts
    torch.manual_seed(0)
    d = U.dev()
    B, T, F, V, S = 8, 40, 16, 9, 6
    feats = torch.randn(B, T, F, device=d)
    targets = torch.randint(1, V, (B, S), device=d)
    xl = torch.full((B,), T, dtype=torch.long, device=d)
    tl = torch.full((B,), S, dtype=torch.long, device=d)
    model = torch.nn.Sequential(torch.nn.Linear(F, 64), torch.nn.Tanh(), torch.nn.Linear(64, V)).to(d)
    opt = torch.optim.Adam(model.parameters(), lr=2e-2)
    ctc = CTCLoss(reduce=True, size_average=True)
    first = last = None
    for it in range(300):
        opt.zero_grad()
        loss = ctc(model(feats), targets, xl, tl)
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
        last = loss.item()
    assert np.isfinite(last) and last < 0.05 * first
    logits = model(feats).detach()
    labels = ["_"] + [chr(97 + i) for i in range(V - 1)]
    want = ["".join(labels[k] for k in row) for row in targets.tolist()]
    # collapse repeats in the references the way CTC does (a repeated label needs a blank; the model learned that)
    greedy = CTCDecoder(beam_width=1, blank_idx=0, labels=labels).decode(logits, xl)
    beam = CTCDecoder(beam_width=16, blank_idx=0, labels=labels, wip=0.0).decode(logits, xl)
    assert greedy.decoded_sentences == want and beam.decoded_sentences == want


def _events_ms(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]


def test_headline_call_is_not_pathologically_slow():
    """A performance guard, not a benchmark: the same ISA has run 6.5x slower when a kernel's code object was laid out
    differently (61 -> 401 us for the segment kernel), and nothing else in the suite would notice.  HIP events
    around the C-ABI calls at the sizes BASELINE.json names, limits ~3x the recorded numbers (boxes of the pool differ by
    12 %): configs[1] loss call 0.15 ms -> < 0.45; one GPU's share of configs[4] 1.75 ms -> < 5; configs[3] beam search
    13 ms -> < 40; the mislabelled-utterance regime of round 5 (0.45 ms) -> < 1.5."""
    import bench
    d = U.dev()
    # The limits are absolute times of an idle MI355X; a shared or partitioned GPU, a profiler or another SKU slows everything
    # down with no defect behind it.  The yardstick is timed in the same process: the library's streaming copy of
    # 256 MiB each way (5-6 TB/s on an idle MI355X); the limits stretch by what it falls short of 4 TB/s.  E2E_PERF_GUARD=0 skips.
    if os.environ.get("E2E_PERF_GUARD", "1") == "0":
        pytest.skip("E2E_PERF_GUARD=0")
    from end2end_amd import _lib
    L = _lib.load()
    n = 256 << 20
    src = torch.empty(n, dtype=torch.uint8, device=d).zero_()
    dst = torch.empty_like(src)
    cp = lambda: _lib.check(L.e2e_debug_stream_copy(dst.data_ptr(), src.data_ptr(), n, _lib.stream_ptr(d)))
    for _ in range(3):
        cp()
    tbs = 2 * n / (_events_ms(cp, 10) * 1e-3) / 1e12
    slack = max(1.0, 4.0 / tbs)
    del src, dst
    w = bench.WORKLOAD
    _, db = bench.make_batch(1000, w["B"], w["T"], w["V"], w["S"], d)
    hp = bench.HotPath(db)
    for _ in range(3):
        hp.call(hp.means[0, :1])
    ms = _events_ms(lambda: hp.call(hp.means[0, :1]), 20)
    assert ms < 0.45 * slack, "configs[1] loss call: %.3f ms" % ms
    del hp, db
    x, tg, xl, tl = bench.aligned_batch(10, w["B"], w["T"], w["V"], w["S"], 10.0)
    for k in range(8):
        tg[32 * k], tl[32 * k] = tg[32 * k + 1].clone(), tl[32 * k + 1].clone()
    hp = bench.HotPath(tuple(t.to(d) for t in (x, tg, xl, tl)))
    for _ in range(3):
        hp.call(hp.means[0, :1])
    ms = _events_ms(lambda: hp.call(hp.means[0, :1]), 10)
    assert ms < 1.5 * slack, "configs[1] shape with 8 mislabelled utterances: %.3f ms" % ms
    assert torch.isfinite(hp.losses).all()
    del hp
    ww = bench.WIDE
    _, wb = bench.make_batch(5000, ww["B"], ww["T"], ww["V"], ww["S"], d)
    hp = bench.HotPath(wb)
    for _ in range(2):
        hp.call()
    ms = _events_ms(hp.call, 5)
    assert ms < 5.0 * slack, "configs[4] share: %.3f ms" % ms
    del hp, wb
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(2)
    xb = torch.log_softmax(torch.randn(64, 1500, 29, generator=g) * 3, -1).to(d)
    xlb = torch.full((64,), 1500, dtype=torch.long, device=d)
    labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
    eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True, labels=labels, wip=1.0)._decoder
    eng.decode(xb, xlb)
    ms = _events_ms(lambda: eng.decode(xb, xlb), 3)
    assert ms < 40.0 * slack, "configs[3] beam search without LM: %.3f ms" % ms

"""Alphabets of 97..224 columns on the fast lattice kernels (ChainF64W + the segment kernel's wide-row form), directly and
behind the wide path's compaction: ALGO_FAST against the oracle, then timings of the word-piece shapes."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import oracle_lib as O, gpu_util as U
from end2end_amd import _lib

def check(B, T, V, S, logprobs=False, dtype=torch.float32, seed=3, ragged=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, V, generator=g)
    if logprobs: x = torch.log_softmax(x, -1)
    x = x.to(dtype)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.tensor([T] + [T - 9 * (b + 1) for b in range(B - 1)]) if ragged else torch.full((B,), T)
    tl = torch.tensor([S] + [max(1, S - 17 * (b + 1)) for b in range(B - 1)])
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, logprobs, _lib.ALGO_FAST)
    xd = x.double()
    lp = (xd if logprobs else torch.log_softmax(xd, -1)).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    if logprobs:
        pass
    for b in range(B):
        if not logprobs: g_o[b, xl[b]:] = 0
    nbad = int(np.isnan(lf).sum())
    lerr = np.nanmax(np.abs(lf - l_o) / np.maximum(1, np.abs(l_o)))
    ok = np.isfinite(gf).all(axis=(1, 2))
    gerr = np.abs(gf[ok].astype(np.float64) - g_o[ok]).max() if ok.any() else float("nan")
    print("B%d T%d V%d S%d lp=%d %s: flagged %d  loss rel err %.2e  grad abs err %.2e" % (B, T, V, S, logprobs, str(dtype)[6:], nbad, lerr, gerr), flush=True)

for shape in [(2, 300, 97, 120), (3, 200, 128, 100), (2, 256, 129, 150), (3, 400, 224, 223), (2, 300, 200, 40), (4, 64, 177, 30)]:
    check(*shape)
    check(*shape, logprobs=True)
for shape in [(2, 500, 225, 224), (2, 700, 300, 300), (3, 900, 448, 447), (2, 600, 448, 100), (2, 300, 97, 250), (2, 640, 150, 400)]:
    check(*shape)
    check(*shape, logprobs=True)
check(2, 700, 8000, 300, ragged=False)
check(1, 600, 32000, 447, ragged=False)
check(2, 256, 8000, 200, ragged=False)
check(2, 150, 32000, 120, ragged=False)

d = torch.device("cuda", 0)
for (B, T, V, S) in [(64, 256, 8000, 200), (16, 150, 32000, 120), (256, 1000, 200, 200), (32, 700, 8000, 400)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, V, generator=g).to(d); tg = torch.randint(1, V, (B, S), generator=g).to(d)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g).to(d); xl = torch.full((B,), T).to(d)
    L = _lib.load()
    losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
    n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 0); ws = torch.zeros(n, dtype=torch.uint8, device=d)
    def call():
        rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                    B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 0, None)
        _lib.check(rc)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    print("B%d T%d V%d S%d AUTO: %.3f ms per call, losses finite: %s" % (B, T, V, S, e0.elapsed_time(e1) / 10, bool(torch.isfinite(losses).all())), flush=True)

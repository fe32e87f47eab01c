import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import gpu_util as U, oracle_lib as O
from end2end_amd import _lib
g = torch.Generator().manual_seed(5)
for (B, T, V, S) in ((2, 400, 500, 150), (2, 700, 29, 300), (1, 650, 3000, 300), (1, 900, 9000, 400), (2, 300, 97, 120)):
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.full((B,), T); tl = torch.tensor([S] + [S // 2] * (B - 1))
    try:
        la, ga = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_AUTO)
    except Exception as e:
        print((B, T, V, S), "ERROR", str(e)[:200]); continue
    lp = torch.log_softmax(x.double(), -1).numpy()
    lo, go = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    print((B, T, V, S), "loss rel", float(np.max(np.abs(la - lo) / np.abs(lo))), "grad abs", float(np.abs(ga - go).max()))

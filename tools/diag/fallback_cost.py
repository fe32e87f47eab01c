"""Cost of the exact fallback: time E2E_ALGO_AUTO on a batch where a few utterances are flagged by the fast path
(partially trained regime, boost 2 over unit noise) and print which flags fired."""
import sys, os, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from end2end_amd import _lib
L = _lib.load(); d = torch.device("cuda", 0)
rng = np.random.default_rng(0)
B, T, V, S = 256, 1000, 29, 200
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for boost in [1.0, 2.0, 4.0, 8.0]:
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B)
    for b in range(B):
        Lb = int(tl[b])
        if variant == 0:      # labels on random even frames, blank boosted everywhere else
            slots = np.sort(rng.choice(np.arange(0, T, 2), size=Lb, replace=False))
            x[b, slots, tg[b, :Lb]] += boost
            rest = np.setdiff1d(np.arange(T), slots); x[b, rest, 0] += boost
        else:                 # only the labels' frames are informative, bunched into the first part of the utterance
            slots = np.sort(rng.choice(np.arange(0, int(T * 0.6)), size=Lb, replace=False))
            x[b, slots, tg[b, :Lb]] += boost
    xd = torch.from_numpy(x).to(d); tgd = torch.from_numpy(tg).to(d); tld = torch.from_numpy(tl).to(d); xld = torch.full((B,), T).to(d)
    losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
    res = {}
    for algo, name in [(2, "fast"), (0, "auto")]:
        n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, algo); ws = torch.zeros(n, dtype=torch.uint8, device=d)
        def call():
            rc = L.e2e_ctc_loss_fwd_bwd(xd.data_ptr(), 0, 0, *xd.stride(), tgd.data_ptr(), tgd.stride(0), xld.data_ptr(), tld.data_ptr(),
                                        B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), algo, None)
            assert rc == 0
        for _ in range(2): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 5 * 1e3
        if name == "fast":
            flags = (C.c_int * B)(); logz = (C.c_double * (2 * B))()
            L.e2e_debug_fast_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            assert L.e2e_debug_fast_state(ws.data_ptr(), B, T, V, S, flags, logz) == 0
            f = np.array(flags[:]); nflag = int((f != 0).sum()); kinds = sorted(set(f[f != 0].tolist()))
    print("boost %.1f: %3d of %d flagged (flag words %s): fast-only %.0f us, auto (with exact fallback) %.0f us" % (boost, nflag, B, kinds, res["fast"], res["auto"]))

# per-wave cycle accounts of the halo chain kernel (instrumented library: tools/diag/build_profile_lib.sh).  Default: the
# f64 form at this shape; E2E_F1_F32=1: the packed-f32 form.  The wave-count classes printed are S // 56 + 1 (a wave owns
# 112 pairs: classes 1-2 run one wave per direction, 3-4 two).
import sys, ctypes as C, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import end2end_amd._lib as _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.environ.get("E2E_PROF_LIB", "build/diag/prof_lib.so"))
L = _lib.load()
d = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(0)
B, T, V, S = 256, 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 29, int(sys.argv[1]) if len(sys.argv) > 1 else 200      # argv: S [V]
x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 2); ws = torch.zeros(n, dtype=torch.uint8, device=d)
buf = (C.c_ulonglong * (256 * 16 * 4))()
L.e2e_debug_fast_profile3.argtypes = [C.c_void_p, C.c_int]
for it in range(3):
    L.e2e_debug_fast_profile3(None, 1)
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(), B, T, V, S, 0,
                                losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 2, None)
    assert rc == 0
torch.cuda.synchronize()
assert L.e2e_debug_fast_profile3(buf, 0) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 2, 8, 4)
tlc = tl.cpu().numpy()
for W in range(1, 6):
    sel = (tlc // 56 + 1) == W
    if not sel.any(): continue
    print("utterances with %d waves per direction: %d" % (W, sel.sum()))
    for dname, dd in (("alpha", 0), ("beta", 1)):
        for w in range(W):
            r = a[sel, dd, w]
            print("  %-5s wave %d: total %.0f cycles/step | ring wait %.0f | neighbour wait %.0f | frame wait %.0f | rest %.0f" % (
                dname, w, r[:, 0].mean() / T, r[:, 1].mean() / T, r[:, 2].mean() / T, r[:, 3].mean() / T,
                (r[:, 0] - r[:, 1] - r[:, 2] - r[:, 3]).mean() / T))
p = (C.c_ulonglong * (256 * 16))()
L.e2e_debug_fast_profile.argtypes = [C.c_void_p, C.c_int]
assert L.e2e_debug_fast_profile(p, 256 * 16) == 0
q = np.array(p[:], dtype=np.float64).reshape(256, 4, 4)
for w, nm in ((2, "alpha prep (first of 2)"), (3, "beta prep (first of 2)")):
    print("%-24s total %.0f cycles/step, waiting for a free slot %.0f" % (nm, q[:, w, 0].mean() / T, q[:, w, 1].mean() / T))

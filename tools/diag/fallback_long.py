"""The fallback regime (logits x3 against unrelated targets) at shapes with eight pairs per segment-kernel lane -- long
transcripts, alphabets beyond 224 columns --, where range-flagged utterances have no f64 redo of single segments:
time per call and how many utterances the fast path handed over.  python tools/diag/fallback_long.py [B T V S [scale]]"""
import sys, os, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from end2end_amd import _lib
if os.environ.get("E2E_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["E2E_LIB"])
L = _lib.load(); d = torch.device("cuda", 0)
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
shapes = [tuple(int(a) for a in sys.argv[1:5])] if len(sys.argv) >= 5 else [(256, 1000, 29, 200), (256, 1000, 29, 300), (64, 600, 448, 100)]
for (B, T, V, S) in shapes:
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(B, T, V, generator=gen) * scale).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
    tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
    losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
    n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 0); ws = torch.zeros(n, dtype=torch.uint8, device=d)
    def call():
        _lib.check(L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                          B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 0, None))
    for _ in range(2): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): call()
    e1.record(); torch.cuda.synchronize()
    fw = (ctypes.c_int * B)(); lz = (ctypes.c_double * (2 * B))()
    L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    ok = L.e2e_debug_fast_state(ws.data_ptr(), B, T, V, S, fw, lz) == 0
    import collections
    if ok: print("   flag words (1 lengths, 2 blank label, 4 infeasible / inf, 8 range, 16 non-finite, 32 log Z mismatch, 64 tiny emissions, 128 hand-off, 512 redo failed):", dict(collections.Counter(int(f) for f in fw)))
    print("B=%d T=%d V=%d S<=%d: %.3f ms per call, %s utterances handed over, losses finite %s" % (
        B, T, V, S, e0.elapsed_time(e1) / 5, sum(1 for f in fw if f & 0x1ff) if ok else "?", bool(torch.isfinite(losses).all())), flush=True)

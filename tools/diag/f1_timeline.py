"""Where the lean halo chain kernel's time goes OUTSIDE its chain steps: stamps of the first chain wave of
either direction of every workgroup -- kernel entry, behind the entry barrier, first block of probabilities in the ring, last step
done, exit -- on two clocks: s_memtime (what tools/diag's "cycles per step" are counted in) and the 100 MHz wall clock.  Their
ratio is the s_memtime frequency; the workgroups' entry and exit times on the wall clock (one clock for the whole chip) show how
the launch ramps up and drains.  Instrumented library: tools/diag/build_profile_lib.sh.
  python tools/diag/f1_timeline.py [S] > profiles/r06_f1_timeline.txt"""
import sys, ctypes as C, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np, torch
import end2end_amd._lib as _lib
_lib.LIB_PATH = os.path.join(root, os.environ.get("E2E_PROF_LIB", "build/diag/prof_lib.so"))
L = _lib.load()
d = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(0)
B, T, V, S = 256, 1000, 29, int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 2); ws = torch.zeros(n, dtype=torch.uint8, device=d)
buf = (C.c_ulonglong * (256 * 2 * 12))()
L.e2e_debug_fast_timeline_h1.argtypes = [C.c_void_p]
def call():
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(), B, T, V, S, 0,
                                losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 2, None)
    assert rc == 0
for it in range(5): call()
torch.cuda.synchronize()
runs = []
for it in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    assert L.e2e_debug_fast_timeline_h1(buf) == 0
    runs.append((e0.elapsed_time(e1) * 1e3, np.array(buf[:], dtype=np.float64).reshape(256, 2, 6, 2)))
names = ["kernel entry", "behind the entry barrier (LDS cleared)", "first block of probabilities in the ring", "last chain step done", "wave exit (log Z written)"]
print("lean halo chain kernel, B=%d T=%d V=%d S<=%d; instrumented build (the stamps cost a few hundred cycles); five calls" % (B, T, V, S))
for ms, a in runs:
    mt, wc = a[..., 0], a[..., 1] * 10.0          # s_memtime ticks; wall clock in ns
    wc0 = wc[:, :, 0].min()
    f = (mt[:, :, 3] - mt[:, :, 2]) / (wc[:, :, 3] - wc[:, :, 2]) * 1e3      # MHz
    print("\ncall: %.1f us by HIP events (whole call: chains + segments + flagged launch).  s_memtime runs at %.1f MHz (min %.1f max %.1f over the workgroups)" % (
        ms, f.mean(), f.min(), f.max()))
    print("  chip-wide wall clock, us since the first workgroup's entry:")
    print("    workgroup entries: first 0.00, median %.2f, last %.2f" % (np.median(wc[:, 0, 0] - wc0) / 1e3, (wc[:, :, 0].max() - wc0) / 1e3))
    print("    wave exits (alpha side): first %.2f, median %.2f, last %.2f;  (beta side): first %.2f, median %.2f, last %.2f" % (
        (wc[:, 0, 4].min() - wc0) / 1e3, np.median(wc[:, 0, 4] - wc0) / 1e3, (wc[:, 0, 4].max() - wc0) / 1e3,
        (wc[:, 1, 4].min() - wc0) / 1e3, np.median(wc[:, 1, 4] - wc0) / 1e3, (wc[:, 1, 4].max() - wc0) / 1e3))
    for dname, dd in (("alpha", 0), ("beta", 1)):
        print("  %s chain, wave 0 -- per workgroup, mean (min .. max) in us on the wall clock:" % dname)
        for k in range(1, 5):
            seg = (wc[:, dd, k] - wc[:, dd, k - 1]) / 1e3
            print("    %-44s -> %-44s %7.2f (%6.2f .. %6.2f)" % (names[k - 1], names[k], seg.mean(), seg.min(), seg.max()))
        steps = (wc[:, dd, 3] - wc[:, dd, 2]) / 1e3
        print("    steps only: %.2f us = %.1f ns per step = %.0f s_memtime ticks per step" % (steps.mean(), steps.mean() * 1e3 / T, (mt[:, dd, 3] - mt[:, dd, 2]).mean() / T))
    slow = np.argsort(-(wc[:, 0, 4] - wc[:, 0, 0]))[:5]
    print("  slowest workgroups (entry -> exit, us):", ", ".join("%d: %.1f (S=%d)" % (b_, (wc[b_, 0, 4] - wc[b_, 0, 0]) / 1e3, int(tl[b_])) for b_ in slow))

"""Diagnostic: the exact kernel's scaled form against its log-domain form (E2E_EXACT_LOGDOMAIN=1 in a child process) on
the shapes that reach it under AUTO: targets beyond the fast kernels, word-piece alphabets with many distinct labels,
a C2 batch with a few utterances the fast path hands over (a blank-valued label).  Prints time per call and the largest
relative difference of losses / absolute difference of gradients between the two forms."""
import sys, os, subprocess
sys.path.insert(0, os.getcwd())
import torch

CASES = [("S=500 beyond the fast kernels", 32, 1200, 29, 500, 450, False),
         ("word pieces V=8000 S=200", 32, 1000, 8000, 200, 150, False),
         ("word pieces V=32000 S=120", 32, 500, 32000, 120, 100, False),
         ("C2 with 8 blank-valued labels", 256, 1000, 29, 200, 100, True)]

def child():
    from end2end_amd import _lib
    L = _lib.load(); d = torch.device("cuda", 0)
    out = {}
    for name, B, T, V, S, lo, with_blank in CASES:
        gen = torch.Generator().manual_seed(1)
        x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen)
        if with_blank: tg[:8, 5] = 0
        tg = tg.to(d)
        tl = torch.randint(lo, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
        losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
        n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 0); ws = torch.zeros(n, dtype=torch.uint8, device=d)
        def call():
            rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                        B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 0, None)
            assert rc == 0
        for _ in range(2): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        out[name] = (e0.elapsed_time(e1) / 5, losses.cpu(), grads[:8].cpu())
    torch.save(out, sys.argv[2])

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(); sys.exit(0)
res = {}
for tag, env in (("scaled", {}), ("logdomain", {"E2E_EXACT_LOGDOMAIN": "1"})):
    f = "/tmp/scaled_%s.pt" % tag
    subprocess.run([sys.executable, __file__, "child", f], check=True, env={**os.environ, **env})
    res[tag] = torch.load(f)
for name in res["scaled"]:
    ts, ls, gs = res["scaled"][name]; tl_, ll, gl = res["logdomain"][name]
    print("%-36s scaled %8.3f ms  log-domain %8.3f ms  max rel dloss %.2e  max abs dgrad %.2e  (nan %d/%d)" % (
        name, ts, tl_, float(((ls - ll).abs() / ll.abs()).nan_to_num(0).max()), float((gs - gl).abs().nan_to_num(0).max()),
        int(torch.isnan(ls).sum()), int(torch.isnan(ll).sum())))

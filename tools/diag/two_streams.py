"""How much of the segment kernel's work fits beside the chain kernel?  Two INDEPENDENT headline batches are run (a) one after
the other on one stream and (b) at the same time on two streams: in (b) the segment kernel of one batch meets the chain kernel
of the other on the chip, and whatever the pair gains over (a) is the room a fused / overlapped call could use.  With the
regular library the chain kernel's 12 waves x 168 registers fill every SIMD's register file and nothing of the segment kernel
(244 registers) fits beside it; build/diag/ab_coresident.so (tools/diag/build_variant.sh coresident "-DE2E_HX_ABL=2
-DE2E_F2_HALF=1" ctc_loss_fast.hip ctc_loss_fast_h1.hip: chain kernel without its producer waves -- wrong results, 8 waves --,
segment kernel at 168 registers) shows what co-residency would buy.   python tools/diag/two_streams.py lib.so [lib2.so ...]"""
import ctypes as C, os, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from end2end_amd import _lib
d = torch.device("cuda", 0)
B, T, V, S = 256, 1000, 29, 200
def batch(seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
    tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
    return x, tg, xl, tl, torch.empty(B, device=d), torch.empty(B, T, V, device=d)
for path in sys.argv[1:]:
    L = C.CDLL(os.path.join(root, path))
    L.e2e_ctc_loss_workspace_bytes.restype = C.c_size_t; L.e2e_ctc_loss_workspace_bytes.argtypes = [C.c_int] * 6
    L.e2e_ctc_loss_fwd_bwd.restype = C.c_int
    L.e2e_ctc_loss_fwd_bwd.argtypes = _lib.load().e2e_ctc_loss_fwd_bwd.argtypes
    n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 2)
    sets = [batch(s) + (torch.zeros(n, dtype=torch.uint8, device=d),) for s in (1, 2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def call(k, stream):
        x, tg, xl, tl, lo, gr, ws = sets[k]
        rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                    B, T, V, S, 0, lo.data_ptr(), gr.data_ptr(), ws.data_ptr(), ws.numel(), 2, C.c_void_p(stream.cuda_stream))
        assert rc == 0
    def timed(concurrent, reps=20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        streams[1].wait_event(e0)
        for _ in range(reps):
            call(0, streams[0]); call(1, streams[1] if concurrent else streams[0])
        if concurrent:
            ev = torch.cuda.Event(); ev.record(streams[1]); streams[0].wait_event(ev)
        e1.record(streams[0]); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for _ in range(3): timed(False, 3); timed(True, 3)
    ser = [timed(False) for _ in range(6)]; con = [timed(True) for _ in range(6)]
    print("%-32s two batches one after the other %.1f us, on two streams %.1f us (%.1f us per batch)" % (
        os.path.basename(path), statistics.median(ser), statistics.median(con), statistics.median(con) / 2))

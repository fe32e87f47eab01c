"""A/B two builds of libe2e_ctc.so in ONE process on the headline shape (interleaved rounds, median); AB_B / AB_T / AB_V / AB_S: another shape; AB_SCALE: logits scaled (3 = the fallback regime); AB_DTYPE=bf16 / f16: 16-bit logits; AB_ALGO (default 2 = FAST).
(Separate processes are no substitute: the same binary differs by 2-4 % from process to process on the wide-alphabet shape -- where its buffers land --, which is more than most of the effects one wants to see.)"""
import ctypes as C, os, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from end2end_amd import _lib
def bind(path):
    L = C.CDLL(path)
    L.e2e_ctc_loss_workspace_bytes.restype = C.c_size_t; L.e2e_ctc_loss_workspace_bytes.argtypes = [C.c_int] * 6
    L.e2e_ctc_loss_fwd_bwd.restype = C.c_int
    L.e2e_ctc_loss_fwd_bwd.argtypes = _lib.load().e2e_ctc_loss_fwd_bwd.argtypes
    L.e2e_ctc_loss_fwd_bwd_opt.restype = C.c_int
    L.e2e_ctc_loss_fwd_bwd_opt.argtypes = _lib.load().e2e_ctc_loss_fwd_bwd_opt.argtypes
    return L
libs = {os.path.basename(p): bind(os.path.join(root, p)) for p in sys.argv[1:]}
d = torch.device("cuda", 0)
B, T, V, S = (int(os.environ.get(k, v)) for k, v in (("AB_B", "256"), ("AB_T", "1000"), ("AB_V", "29"), ("AB_S", "200")))
gen = torch.Generator().manual_seed(0)
DT = {"bf16": torch.bfloat16, "f16": torch.float16}.get(os.environ.get("AB_DTYPE", ""), torch.float32); CODE = _lib.dtype_code(DT); ALGO = int(os.environ.get("AB_ALGO", "2"))
x = (torch.randn(B, T, V, generator=gen) * float(os.environ.get("AB_SCALE", "1"))).to(DT).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d, dtype=DT)
n = max(L.e2e_ctc_loss_workspace_bytes(B, T, V, S, CODE, ALGO) for L in libs.values()); ws = torch.zeros(n, dtype=torch.uint8, device=d)
MEAN = os.environ.get("AB_MEAN") == "1"          # the call bench.py times: e2e_ctc_loss_fwd_bwd_opt with the batch mean written by the call's tail
mean_out = torch.zeros(1, device=d, dtype=torch.float32)
def call_mean(L):
    o = _lib.LossOpts(1.0 / B, mean_out.data_ptr(), _lib.REDUCE_MEAN, 0)
    rc = L.e2e_ctc_loss_fwd_bwd_opt(x.data_ptr(), CODE, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                    B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), ALGO, None, C.byref(o))
    assert rc == 0
def call_plain(L):
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), CODE, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), ALGO, None)
    assert rc == 0
call = call_mean if MEAN else call_plain
res = {k: [] for k in libs}
for rnd in range(12):
    for k, L in libs.items():
        for _ in range(3): call(L)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call(L)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
for k, v in res.items():
    print("%-28s median %.1f us  min %.1f us" % (k, statistics.median(v), min(v)))

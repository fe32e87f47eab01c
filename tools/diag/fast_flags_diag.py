import sys, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import golden_util as G, oracle_lib as O
from end2end_amd import _lib
L = _lib.load()
d = torch.device("cuda", 0)
def run(x, tg, xl, tl, logprobs, name):
    x = x.to(d); B,T,V = x.shape
    tg = torch.as_tensor(tg).to(d, torch.long).contiguous(); Smax = tg.shape[1]
    xl_d = torch.as_tensor(xl).to(d, torch.long); tl_d = torch.as_tensor(tl).to(d, torch.long)
    losses = torch.empty(B, device=d); grads = torch.empty(B,T,V, device=d)
    n = L.e2e_ctc_loss_workspace_bytes(B,T,V,Smax,0,2)
    ws = torch.zeros(n, dtype=torch.uint8, device=d)
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(),0,int(logprobs),*x.stride(),tg.data_ptr(),tg.stride(0),xl_d.data_ptr(),tl_d.data_ptr(),B,T,V,Smax,0,losses.data_ptr(),grads.data_ptr(),ws.data_ptr(),ws.numel(),2,None)
    assert rc == 0, L.e2e_last_error()
    fl = (C.c_int*B)(); lz = (C.c_double*(2*B))()
    L.e2e_debug_fast_state.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_int,C.c_int,C.c_void_p,C.c_void_p]
    L.e2e_debug_fast_state(ws.data_ptr(),B,T,V,Smax,fl,lz)
    fl = np.array(fl[:]); lz = np.array(lz[:]).reshape(B,2)
    print(name, "flags", fl[:16], "nflag", (fl!=0).sum(), "of", B)
    print("  logz a/b", lz[:4].tolist())
    return losses.cpu().numpy(), grads.cpu().numpy(), fl, lz
c = G.engine_case("long_f32")
l,g,fl,lz = run(torch.from_numpy(c["lp"]), c["targets"], c["x_len"], c["t_len"], True, "long_f32")
print("  want", -c["losses"])
gen = torch.Generator().manual_seed(0)
B,T,V,S = 16,1000,29,200
x = torch.randn(B,T,V,generator=gen); tg = torch.randint(1,V,(B,S),generator=gen); tl = torch.randint(S//2,S+1,(B,),generator=gen); xl=torch.full((B,),T)
l,g,fl,lz = run(x,tg,xl,tl,False,"c2x16")
lo,go = O.ctc_loss(torch.log_softmax(x[:2].double(),-1).numpy(), tg[:2].numpy(), xl[:2].numpy(), tl[:2].numpy(),0)
print("  oracle", -lo, "fast", -l[:2])
ok = fl==0
if ok[:2].all(): print("  grad maxabs diff", np.abs(g[:2]-go).max())

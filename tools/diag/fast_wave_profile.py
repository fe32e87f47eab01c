import sys, ctypes as C, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import end2end_amd._lib as _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.environ.get("E2E_PROF_LIB", "build/diag/prof_lib.so"))
L = _lib.load()
d = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(0)
B,T,V,S = 256,1000,29,int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = torch.randn(B,T,V,generator=gen).to(d); tg = torch.randint(1,V,(B,S),generator=gen).to(d); tl = torch.randint(S//2,S+1,(B,),generator=gen).to(d); xl=torch.full((B,),T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B,T,V, device=d)
n = L.e2e_ctc_loss_workspace_bytes(B,T,V,S,0,2); ws = torch.zeros(n, dtype=torch.uint8, device=d)
for it in range(3):
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(),0,0,*x.stride(),tg.data_ptr(),tg.stride(0),xl.data_ptr(),tl.data_ptr(),B,T,V,S,0,losses.data_ptr(),grads.data_ptr(),ws.data_ptr(),ws.numel(),2,None)
    assert rc == 0
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256*16))()
L.e2e_debug_fast_profile.argtypes = [C.c_void_p, C.c_int]
assert L.e2e_debug_fast_profile(buf, 256*16) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256,4,4)
names = ["alpha chain","beta chain","alpha prep","beta prep"]
for w in range(4):
    print("%-12s total cycles: mean %.0f max %.0f | spin cycles: mean %.0f (%.0f%%)" % (names[w], a[:,w,0].mean(), a[:,w,0].max(), a[:,w,1].mean(), 100*a[:,w,1].mean()/a[:,w,0].mean()))
print("per step (T=1000): alpha %.0f beta %.0f cycles" % (a[:,0,0].mean()/1000, a[:,1,0].mean()/1000))
for w in range(2):
    print("%-12s per step: spin %.0f | ring loads %.0f | 8-step compute %.0f | rest %.0f" % (names[w], a[:,w,1].mean()/1000, a[:,w,2].mean()/1000, a[:,w,3].mean()/1000, (a[:,w,0]-a[:,w,1]-a[:,w,2]-a[:,w,3]).mean()/1000))

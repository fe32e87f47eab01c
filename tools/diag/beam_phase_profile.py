"""Phase breakdown of the fast beam kernel (workgroup 0, s_memtime between the phases): needs the instrumented library,
`bash tools/diag/build_profile_lib.sh` -> build/diag/prof_lib.so.  The kernel is called through the C ABI of THAT library
(the engine would go through the pybind module, which is linked against the regular one).
  python tools/diag/beam_phase_profile.py [W] [lm]"""
import sys, os, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import end2end_amd._lib as _lib
_lib.LIB_PATH = os.path.join(root, os.environ.get("E2E_PROF_LIB", "build/diag/prof_lib.so"))
import gpu_util as U
d = torch.device("cuda", 0)
labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
g = torch.Generator().manual_seed(2)
B, T, W = 64, 300, int(sys.argv[1]) if len(sys.argv) > 1 else 100
x = torch.log_softmax((torch.randn(B, T, 29, generator=g) * 3), -1).to(d)
lm = None
if len(sys.argv) > 2 and sys.argv[2] == "lm":          # the synthetic 3-gram model of bench.py
    import tempfile, bench
    from end2end_amd.engines import LanguageModel
    td = tempfile.mkdtemp(); path = os.path.join(td, "synthetic_3gram.arpa"); bench.synthetic_arpa(path, labels)
    lm = LanguageModel(path, labels, False)
kw = dict(lmwt=1.0, wip=1.0, oov_penalty=-10.0) if lm is not None else dict(wip=1.0)
for _ in range(2): U.c_abi_beam(x, None, 0, W, labels, lm, **kw)
buf = (C.c_ulonglong * 16)()
L = _lib.load(); L.e2e_debug_beam_profile.argtypes = [C.c_void_p]
assert L.e2e_debug_beam_profile(buf) == 0
names = ["(unused)", "pairs", "members", "select: rank + place", "(unused)", "guards+tables / LM followers", "(unused)", "select: radix passes", "select: gather", "(unused)", "(passes)", "guards (LM kernel)", "LM state signatures", "LM leaders", "LM rows"]
tot = sum(buf[k] for k in range(15) if k != 10)
for k, nm in enumerate(names):
    if k != 10 and buf[k]: print("%-30s %8.0f ticks/step (%4.1f%%)" % (nm, buf[k] / T, 100.0 * buf[k] / tot))
print("total %.0f s_memtime ticks/step; radix passes per step %.2f" % (tot / T, buf[10] / T))
if os.environ.get("BEAM_PHASES_JSON"):
    # one record per configuration, merged into the file: {"no_lm": {...}, "lm": {...}}
    import json
    path = os.environ["BEAM_PHASES_JSON"]
    try:
        rec = json.load(open(path))
    except Exception:
        rec = {}
    rec["lm" if lm is not None else "no_lm"] = {
        "workload": "B=64 T=%d V=29 beam=%d%s, workgroup 0" % (T, W, " + synthetic 3-gram ARPA" if lm is not None else ""),
        "cycles_per_step": tot / T, "radix_passes_per_step": buf[10] / T,
        "phase_cycles": {nm: buf[k] / T for k, nm in enumerate(names) if k != 10 and buf[k]},
        "unit": "s_memtime ticks (shader cycles) per frame"}
    json.dump(rec, open(path, "w"), indent=1)
if lm is not None and hasattr(L, "e2e_debug_beam_sigs"):
    # how often does an LM state that has to ask come back later in the same utterance?
    L.e2e_debug_beam_sigs.argtypes = [C.c_void_p, C.c_int]
    cap = 1 << 17
    sb = (C.c_ulonglong * cap)()
    L.e2e_debug_beam_sigs(sb, cap)                      # (drop what the warm-up calls left)
    U.c_abi_beam(x, None, 0, W, labels, lm, **kw)
    n = L.e2e_debug_beam_sigs(sb, cap)
    sigs = [sb[i] for i in range(n)]
    print("utterance 0: %d asking states over %d steps (%.1f per step), %d distinct (%.0f%% would hit a per-utterance cache)" % (
        n, T, n / T, len(set(sigs)), 100.0 * (1 - len(set(sigs)) / max(n, 1))))

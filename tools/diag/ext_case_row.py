"""One case of fuzz_ext_vs_oracle.py in detail: the row with the largest gradient error, column by column (FUZZ_ONLY / seed as there)."""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(root, "tools", "diag"))
import numpy as np
g = runpy.run_path(os.path.join(root, "tools", "diag", "fuzz_ext_vs_oracle.py"))
ga, g_o, la, l_o, xl = g["ga"], g["g_o"], g["la"], g["l_o"], g["xl"]
d = np.abs(ga.astype(np.float64) - g_o)
b, t, v = np.unravel_index(d.argmax(), d.shape)
print("worst element: utterance %d frame %d column %d: got %.9g want %.9g" % (b, t, v, ga[b, t, v], g_o[b, t, v]))
lp = g["lp32"][b, t].numpy() if "lp32" in g else g["lp"][b, t].numpy()
y = np.exp(lp)
post_o = y - g_o[b, t]; post_g = y - ga[b, t].astype(np.float64)
idx = np.argsort(-np.abs(post_o))[:8]
print("largest posteriors of the row (column, oracle, ours, ratio):")
for k in idx: print("  %4d  %.9e  %.9e  %.6f" % (k, post_o[k], post_g[k], post_g[k] / post_o[k] if post_o[k] else float("nan")))
print("row sums of the posterior: oracle %.9f ours %.9f" % (post_o.sum(), post_g.sum()))
for dt in (-2, -1, 1, 2):
    if 0 <= t + dt < xl[b]:
        yy = np.exp((g["lp32"] if "lp32" in g else g["lp"])[b, t + dt].numpy())
        print("frame %d: posterior sums oracle %.9f ours %.9f; max err %.3e" % (t + dt, (yy - g_o[b, t + dt]).sum(), (yy - ga[b, t + dt]).sum(), d[b, t + dt].max()))

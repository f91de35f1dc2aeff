"""Iterations the lmax estimate needs from different start vectors.  python tools/lanczos_start.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from meld_amd import filter as mf
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, _ = synthetic_cells(N, 50, seed=0)
G = meld_amd.MELD(knn=15, verbose=0).fit(torch.from_numpy(X).cuda()).graph
ops = mf._ops_of(G)
idx = torch.arange(G.n_pad, dtype=torch.float64, device="cuda")
rnd = torch.frac(torch.sin(idx * 12.9898 + 1.0) * 43758.5453) - 0.5
dw = G.dw_dev[: G.n_pad].clone()
cands = {"pseudo-random (current)": rnd, "dw * random": dw * rnd, "dw": dw.clone(), "dw - mean": dw - dw.mean(), "dw^2 * random": dw * dw * rnd,
         "dw^4 * random": dw ** 4 * rnd, "dw^8 * random": (dw / dw.max()) ** 8 * rnd, "dw^16 * random": (dw / dw.max()) ** 16 * rnd,
         "dw^4": (dw / dw.max()) ** 4, "dw^16": (dw / dw.max()) ** 16}
ref = None
for name, u in cands.items():
    u = u.clone(); u[G.N:] = 0
    for tol in (1e-3,):
        th, info = mf._lanczos_lmax_device(G, ops, u, tol, 300, 5)
        print("%-28s tol %.0e: theta %.8f after %3d iterations (residual %.2e)" % (name, tol, th, info["iterations"], info["residual"]))
th, info = mf._lanczos_lmax_device(G, ops, rnd, 1e-6, 300, 5)
print("reference (tol 1e-6): %.8f after %d iterations" % (th, info["iterations"]))

"""lmax estimate vs Lanczos tolerance on the benchmark graph: python tools/lmax_tol.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch

import meld_amd
from meld_amd.filter import lanczos_lmax
from oracle import meld_oracle as mo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, labels = mo.synthetic_cells(N, n_dims=50, seed=0)
op = meld_amd.MELD(knn=15, verbose=0).fit(torch.from_numpy(X).cuda())
G = op.graph
ref, info = lanczos_lmax(G, tol=1e-9, max_iter=400)
print("reference (tol 1e-9): %.15g  iterations %d" % (ref, info["iterations"]))
for tol in (1e-2, 5e-3, 3e-3, 2e-3, 1e-3, 3e-4, 1e-4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th, inf = lanczos_lmax(G, tol=tol)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("tol %.0e: iterations %3d  rel.err %.2e  residual %.2e  %.1f ms" % (tol, inf["iterations"], abs(th - ref) / ref, inf["residual"], 1e3 * dt))

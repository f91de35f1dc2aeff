"""Build a graph once and time the Chebyshev recurrence only: python tools/cheby_only.py N [reorder]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from meld_amd import graph as mg
from meld_amd.filter import chebyshev_apply, chebyshev_coefficients, spectral_kernel
from oracle import meld_oracle as mo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reorder = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
X, labels = mo.synthetic_cells(n, n_dims=50, seed=0)
G = meld_amd.build_knn_graph(torch.from_numpy(X).cuda(), knn=15, reorder=reorder)
lmax = 0.12
c = chebyshev_coefficients(spectral_kernel("heat", 60, 0, 1, lmax), lmax, 30)
p = 2
s = torch.rand(n, p, dtype=torch.float64, device="cuda")
for r in range(3):
    mg.record_events(True)
    out = chebyshev_apply(G, s, c, lmax)
    torch.cuda.synchronize()
    ms = mg.event_times_ms()["cheby_steps"][0]
byts = 12 * G.nnz + 4 * (n + 1) + 8 * n + 40 * n * p
print("N=%d reorder=%s nnz=%d  %.1f us/step  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (n, reorder, G.nnz, 1e3 * ms / 29, byts / (ms / 29 * 1e-3) / 1e9, 100 * byts / (ms / 29 * 1e-3) / 8e12))
